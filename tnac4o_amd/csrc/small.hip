// Small dense building blocks shared by the QR (K3) and Jacobi SVD (K4/K5) drivers:
//   gram_partial        partial Gram matrices of <= 64 long vectors (split over the long dimension)
//   eig_small           sum partials + parallel-order two-sided Jacobi on the <= 64 x 64 Gram matrix (one workgroup)
//   rows_times_small    X[r, :b] <- X[r, :b] . S      (tall panel times small matrix, in place)
//   small_t_times_vecs  vecs <- S^T . vecs            (small matrix times a bundle of long vectors, in place)
// All are HBM/LDS-streaming VALU kernels; the flops that matter live in gemm_f64.hip.
#include <stdlib.h>
#include <algorithm>
#include <mutex>
#include <type_traits>

#include "common.h"

namespace tn {

// ------------------------------------------------------------------------------------------ addressing
__device__ __forceinline__ int64_t vec_offset(int v, int w, int blk0, int blk1, int64_t vs) {
    return v < w ? ((int64_t)blk0 * w + v) * vs : ((int64_t)blk1 * w + (v - w)) * vs;
}

// ------------------------------------------------------------------------------------------ gram_partial
template <int NV>
__global__ __launch_bounds__(256) void gram_partial_kernel(const double* __restrict__ X, int64_t vs, int64_t es,
                                                           int64_t L, int nvec, int w, const int* __restrict__ pairs,
                                                           int nchunk, double* __restrict__ part) {
    constexpr int KC = 64, NVP = NV + 1;
    constexpr int TPR = 256 / NV;        // threads per Gram row
    constexpr int JW = NV / TPR;         // Gram columns per thread
    __shared__ double Xs[KC * NVP];
    const int tid = threadIdx.x, chunk = blockIdx.x, grp = blockIdx.y;
    const int blk0 = pairs ? pairs[2 * grp] : 0, blk1 = pairs ? pairs[2 * grp + 1] : 1;
    const int64_t lc = ((L + nchunk - 1) / nchunk + KC - 1) / KC * KC;
    const int64_t c_lo = (int64_t)chunk * lc, c_hi = (c_lo + lc < L) ? c_lo + lc : L;
    const int i = tid / TPR, j0 = (tid % TPR) * JW;
    double acc[JW];
#pragma unroll
    for (int j = 0; j < JW; ++j) acc[j] = 0.0;
    const bool efast = (es == 1);
    for (int64_t c0 = c_lo; c0 < c_hi; c0 += KC) {
        for (int idx = tid; idx < NV * KC; idx += 256) {
            const int v = efast ? idx / KC : idx % NV;
            const int c = efast ? idx % KC : idx / NV;
            double x = 0.0;
            if (v < nvec && c0 + c < c_hi) x = X[vec_offset(v, w, blk0, blk1, vs) + (c0 + c) * es];
            Xs[c * NVP + v] = x;
        }
        __syncthreads();
#pragma unroll 4
        for (int c = 0; c < KC; ++c) {
            const double xi = Xs[c * NVP + i];
#pragma unroll
            for (int j = 0; j < JW; ++j) acc[j] += xi * Xs[c * NVP + j0 + j];
        }
        __syncthreads();
    }
    double* out = part + ((int64_t)grp * nchunk + chunk) * nvec * nvec;
    if (i < nvec)
#pragma unroll
        for (int j = 0; j < JW; ++j)
            if (j0 + j < nvec) out[i * nvec + j0 + j] = acc[j];
}

// chunks of 64 elements (what a workgroup of the LDS-resident SVD kernel holds of every vector: the partial sums of all forms
// then add up in the same order), fewer and longer ones beyond 4096
int gram_nchunk(int64_t L) {
    int64_t n = (L + 63) / 64;
    if (n < 1) n = 1;
    if (n > 64) n = 64;
    return (int)n;
}

int gram_partial(hipStream_t st, const double* X, int64_t vs, int64_t es, int64_t L, int nvec, int w, const int* pairs,
                 int ngroups, int nchunk, double* part) {
    TN_CHECK_ARG(nvec >= 1 && nvec <= NBMAX, "nvec out of range");
    if (ngroups <= 0) return 0;
    dim3 grid(nchunk, ngroups);
    prof_begin(st, PROF_GRAM);
    if (nvec <= 32)
        hipLaunchKernelGGL((gram_partial_kernel<32>), grid, dim3(256), 0, st, X, vs, es, L, nvec, w, pairs, nchunk, part);
    else
        hipLaunchKernelGGL((gram_partial_kernel<64>), grid, dim3(256), 0, st, X, vs, es, L, nvec, w, pairs, nchunk, part);
    TN_CHECK_LAUNCH("gram_partial_kernel");
    prof_end(st, PROF_GRAM, 2.0 * nvec * nvec * (double)L * ngroups, 8.0 * ngroups * ((double)nvec * L + (double)nchunk * nvec * nvec));
    return 0;
}

// ------------------------------------------------------------------------------------------ eig_small
// Round-robin ("circle") pairing of n (even) indices: step s in [0, n-1), slot a in [0, n/2).
__device__ __forceinline__ void rr_pair(int n, int s, int a, int& p, int& q) {
    if (a == 0) {
        p = n - 1;
        q = s;
    } else {
        p = (s + a) % (n - 1);
        q = (s - a + (n - 1)) % (n - 1);
    }
    if (p > q) {
        const int t = p;
        p = q;
        q = t;
    }
}

template <int NB>
__global__ __launch_bounds__(256) void eig_small_kernel(const double* __restrict__ part, int nchunk, int nvec, int mode,
                                                        int max_sweeps, double dead_thresh,
                                                        double* __restrict__ out, int* __restrict__ dead,
                                                        int* __restrict__ nrot_out, double* __restrict__ maxoff_out,
                                                        double relevant2) {
    constexpr int P = NB + 1;
    __shared__ double G[NB * P];
    __shared__ double J[NB * P];
    __shared__ double dsc[NB];
    __shared__ int cnt, total;
    __shared__ int stepflag[2];             // "some rotation in this step", double-buffered over consecutive steps
    __shared__ double red[256];
    const int tid = threadIdx.x, grp = blockIdx.x;
    const int n = (nvec + 1) & ~1;          // even working size (a padding index never rotates)
    const double* pg = part + (int64_t)grp * nchunk * nvec * nvec;

    {   // Gram = sum of the split-K partials.  All of a thread's loads of one chunk are issued together (a loop that
        // loads, adds and moves on would wait a full memory round trip per element: 128 dependent trips here).
        constexpr int EPT = NB * NB / 256;
        double acc[EPT];
#pragma unroll
        for (int t = 0; t < EPT; ++t) acc[t] = 0.0;
        for (int c = 0; c < nchunk; c += 2) {
            double v0[EPT], v1[EPT];
            const bool two = (c + 1 < nchunk);
#pragma unroll
            for (int t = 0; t < EPT; ++t) {
                const int e = tid + 256 * t, i = e / NB, j = e % NB;
                const bool in = (i < nvec && j < nvec);
                const int64_t o = (int64_t)c * nvec * nvec + i * nvec + j;
                v0[t] = in ? pg[o] : 0.0;
                v1[t] = (in && two) ? pg[o + (int64_t)nvec * nvec] : 0.0;
            }
#pragma unroll
            for (int t = 0; t < EPT; ++t) acc[t] = (acc[t] + v0[t]) + v1[t];
        }
#pragma unroll
        for (int t = 0; t < EPT; ++t) {
            const int e = tid + 256 * t, i = e / NB, j = e % NB;
            const bool in = (i < nvec && j < nvec);
            G[i * P + j] = in ? acc[t] : (i == j ? 1.0 : 0.0);
            J[i * P + j] = (i == j) ? 1.0 : 0.0;
        }
    }
    if (tid == 0) total = 0;
    __syncthreads();
    if (mode != 2) {
        if (tid < NB) {
            const double gii = G[tid * P + tid];
            const bool ok = (gii > dead_thresh) && (gii < 1.7e308);
            dsc[tid] = ok ? sqrt(gii) : 1.0;
            if (tid < nvec && dead) dead[grp * nvec + tid] = ok ? 0 : 1;
            if (!ok) dsc[tid] = 0.0;        // marks a dead column
        }
        __syncthreads();
        for (int e = tid; e < NB * NB; e += 256) {
            const int i = e / NB, j = e % NB;
            const double di = dsc[i], dj = dsc[j];
            double g;
            if (di == 0.0 || dj == 0.0) g = (i == j) ? 1.0 : 0.0;
            else g = (i == j) ? 1.0 : G[i * P + j] * fast_rcp(di * dj);
            G[i * P + j] = g;
        }
        __syncthreads();
    }
    // largest relative off-diagonal before rotating (diagnostic / convergence measure): max of g_ij^2 / (g_ii g_jj) with the
    // reciprocal diagonal prepared once, one square root at the end, wave-level reduction
    {
        __shared__ double rdg[NB];
        if (tid < NB) {
            const double gii = fabs(G[tid * P + tid]);
            // vectors whose squared norm is <= relevant2 do not enter the convergence measure (SVD: they lie a factor 4 below
            // the truncation threshold and are discarded whatever their mutual angles are); a group in which only such
            // vectors are non-orthogonal is not rotated at all
            rdg[tid] = (gii > relevant2 && tid < nvec) ? fast_rcp(gii) : 0.0;
        }
        __syncthreads();
        double m = 0.0;
#pragma unroll
        for (int t = 0; t < NB * NB / 256; ++t) {
            const int e = tid + 256 * t, i = e / NB, j = e % NB;
            const double g = G[i * P + j];
            const double r2 = (i < j) ? g * g * rdg[i] * rdg[j] : 0.0;
            m = r2 > m ? r2 : m;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
        if ((tid & 63) == 0) red[tid >> 6] = m;
        __syncthreads();
        if (tid == 0) {
            const double mm = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
            red[0] = sqrt(mm);
            if (maxoff_out) maxoff_out[grp] = red[0];
        }
        __syncthreads();
    }

    // Parallel-order Jacobi on the full NB x NB arrays (indices >= nvec are isolated: unit diagonal, zero coupling,
    // so they never rotate).  Trip counts are compile-time so that each phase issues all of its LDS loads at once.
    const bool need = (mode != 1) && (nvec >= 2) && (red[0] > 8.881784197001252e-16);
    if (need) {
        // Each step of the round-robin tournament applies NB/2 disjoint plane rotations R:  G <- R^T G R  and  J <- J R.
        //  * G lives in LDS and is updated as independent 2 x 2 blocks (rotation a x rotation b) by all 256 threads.  A thread
        //    keeps the same rotation slots for the whole sweep and derives their index pairs arithmetically; the HP lanes that
        //    share a row read HP distinct columns of it (no bank conflicts).
        //  * J never touches LDS during the sweeps (it was half of the traffic of an LDS-bound step): lane i of wave 3 holds
        //    ROW i of J in registers, so a rotation of columns (p, q) mixes two registers of the same lane.  Register indices
        //    must be compile-time constants, so the row is kept in a frame that turns with the tournament: in step s position f
        //    holds column (f + s) mod (NB-1) (position NB-1 holds column NB-1), which makes the pairs of every step the fixed
        //    positions (a, NB-1-a) and (0, NB-1); after the rotations the frame advances by one position (NB-2 register moves).
        //    A sweep has NB-1 steps, so the frame is back in place at its end.  J only follows the rotations, nothing of a
        //    later step depends on it: wave 3 applies step s-1 while wave 0 decides the rotations of step s.
        constexpr int HP = NB / 2, GB = HP * HP / 256;
        __shared__ __attribute__((aligned(16))) double cs2[2][HP * 2];      // (c, s) of slot a, double-buffered over the steps
        __shared__ __attribute__((aligned(16))) double csj[2][HP * 2];      // (c, +-s): the same rotation as seen from J's frame
        const double tol = 8.881784197001252e-16;      // 2^-50
        const int tb = tid % HP, t0 = tid / HP;
        constexpr int TS = 256 / HP;
        constexpr int nsteps = NB - 1;
        auto slot_pair = [](int s, int a, int& p, int& q) {
            if (a == 0) { p = NB - 1; q = s; }
            else { p = s + a; p -= (p >= NB - 1) ? NB - 1 : 0; q = s - a; q += (q < 0) ? NB - 1 : 0; }
            if (p > q) { const int t = p; p = q; q = t; }
        };
        const bool jwave = (tid >= 192);                   // wave 3
        double jr[NB];                                     // (wave 3) row `tid - 192` of J in the turning frame
#pragma unroll
        for (int k = 0; k < NB; ++k) jr[k] = (k == tid - 192) ? 1.0 : 0.0;
        // The frame is advanced once per group of FU steps (FU divides NB-1), so that the register moves are amortised: within
        // a group, step s = s0 + u finds column (s + a) at position (u + a) mod (NB-1) and column (s - a) at (u - a) mod (NB-1),
        // column s itself (slot 0) at position u -- all compile-time constants once the loop over u is unrolled.
        constexpr int FU = (NB == 64) ? 7 : ((NB == 32) ? 1 : 1);      // 63 = 9 x 7; 31 is prime (one step per group)
        static_assert((NB - 1) % FU == 0, "group length must divide the sweep");
        auto apply_j = [&](int t, auto uc) {               // rotations of step t (position offset u) on the register row
            constexpr int u = decltype(uc)::value;
            if (stepflag[t & 1]) {
                double2 r[HP];
#pragma unroll
                for (int a = 0; a < HP; ++a) r[a] = *reinterpret_cast<const double2*>(&csj[t & 1][2 * a]);      // all reads in flight
                {   // slot 0: (column s at position u, column NB-1 at position NB-1)
                    const double x = jr[u], y = jr[NB - 1];
                    jr[u] = r[0].x * x - r[0].y * y;
                    jr[NB - 1] = r[0].y * x + r[0].x * y;
                }
#pragma unroll
                for (int a = 1; a < HP; ++a) {
                    constexpr int M = NB - 1;
                    const int px = (u + a) % M, py = (u - a + M) % M;
                    const double x = jr[px], y = jr[py];
                    jr[px] = r[a].x * x - r[a].y * y;
                    jr[py] = r[a].y * x + r[a].x * y;
                }
            }
        };
        auto advance_frame = [&]() {                       // position f <- position (f + FU) mod (NB-1)
            double tmp[FU];
#pragma unroll
            for (int f = 0; f < FU; ++f) tmp[f] = jr[f];
#pragma unroll
            for (int f = 0; f < NB - 1 - FU; ++f) jr[f] = jr[f + FU];
#pragma unroll
            for (int f = 0; f < FU; ++f) jr[NB - 1 - FU + f] = tmp[f];
        };
        auto step = [&](int s, auto uc, int& mine) -> void {     // one step of the tournament; u = s mod FU
            constexpr int u = decltype(uc)::value;
            if (tid < 64) {                                // wave 0 decides the rotations of this step
                bool rot = false;
                if (tid < HP) {
                    int p, q;
                    slot_pair(s, tid, p, q);
                    const double gpq = G[p * P + q], gpp = G[p * P + p], gqq = G[q * P + q];
                    double c = 1.0, sn = 0.0;
                    // rotate iff |g_pq| > tol sqrt(g_pp g_qq), tested on the squares (no square root on the critical path)
                    const double g2 = gpq * gpq;
                    if (g2 > tol * tol * fabs(gpp * gqq)) {
                        // smaller-angle rotation from the double angle: cos 2t = |d| / hyp, d = g_qq - g_pp,
                        // hyp^2 = d^2 + 4 g_pq^2;  c^2 = (1 + cos 2t) / 2,  s = g_pq / (hyp c) with the sign of d g_pq
                        const double d = gqq - gpp;
                        const double rh = fast_rsqrt(d * d + 4.0 * g2);
                        const double c2 = 0.5 + 0.5 * fabs(d) * rh;
                        const double rcv = fast_rsqrt(c2);
                        const double sabs = fabs(gpq) * rh * rcv;
                        if (sabs <= 1.0 && c2 <= 1.0000000000000002) {     // (fails for non-finite intermediates: no rotation)
                            c = c2 * rcv;
                            sn = ((d >= 0.0) == (gpq >= 0.0)) ? sabs : -sabs;
                            ++mine;
                            rot = true;
                        }
                    }
                    *reinterpret_cast<double2*>(&cs2[s & 1][2 * tid]) = make_double2(c, sn);
                    // J's frame holds the pair as (column (s + a) mod (NB-1), column (s - a) mod (NB-1)), slot 0 as (s, NB-1);
                    // the rotation is defined on (smaller, larger) column: flip the sine where the frame has them the other way
                    int up = s + tid; up -= (up >= NB - 1) ? NB - 1 : 0;
                    const bool flipped = (tid > 0) && (up != p);
                    *reinterpret_cast<double2*>(&csj[s & 1][2 * tid]) = make_double2(c, flipped ? -sn : sn);
                }
                const unsigned long long any = __ballot(rot);
                if (tid == 0) stepflag[s & 1] = (any != 0ull) ? 1 : 0;
            } else if (jwave && s > 0) {                   // J follows one step behind (reads the other halves of csj / stepflag)
                if constexpr (u == 0) {
                    apply_j(s - 1, std::integral_constant<int, FU - 1>{});
                    advance_frame();
                } else {
                    apply_j(s - 1, std::integral_constant<int, u - 1>{});
                }
            }
            __syncthreads();
            if (stepflag[s & 1] == 0) return;              // uniform: nothing to rotate in this step
            {
                int pb, qb;
                slot_pair(s, tb, pb, qb);
                const double2 rb = *reinterpret_cast<const double2*>(&cs2[s & 1][2 * tb]);
                const double cb = rb.x, sb = rb.y;
                double g00[GB], g01[GB], g10[GB], g11[GB];
                int pa[GB], qa[GB];
                double2 ra[GB];
#pragma unroll
                for (int k = 0; k < GB; ++k) {
                    slot_pair(s, t0 + TS * k, pa[k], qa[k]);
                    ra[k] = *reinterpret_cast<const double2*>(&cs2[s & 1][2 * (t0 + TS * k)]);
                    g00[k] = G[pa[k] * P + pb]; g01[k] = G[pa[k] * P + qb];
                    g10[k] = G[qa[k] * P + pb]; g11[k] = G[qa[k] * P + qb];
                }
#pragma unroll
                for (int k = 0; k < GB; ++k) {
                    const double ca = ra[k].x, sa = ra[k].y;
                    const double t00 = ca * g00[k] - sa * g10[k], t01 = ca * g01[k] - sa * g11[k];
                    const double t10 = sa * g00[k] + ca * g10[k], t11 = sa * g01[k] + ca * g11[k];
                    G[pa[k] * P + pb] = cb * t00 - sb * t01; G[pa[k] * P + qb] = sb * t00 + cb * t01;
                    G[qa[k] * P + pb] = cb * t10 - sb * t11; G[qa[k] * P + qb] = sb * t10 + cb * t11;
                }
            }
            __syncthreads();
        };
        for (int sweep = 0; sweep < max_sweeps; ++sweep) {
            if (tid == 0) cnt = 0;
            int mine = 0;                                  // rotations decided by this thread in this sweep
            __syncthreads();
            for (int s0 = 0; s0 < nsteps; s0 += FU) {
                if constexpr (FU == 7) {
                    step(s0 + 0, std::integral_constant<int, 0>{}, mine);
                    step(s0 + 1, std::integral_constant<int, 1>{}, mine);
                    step(s0 + 2, std::integral_constant<int, 2>{}, mine);
                    step(s0 + 3, std::integral_constant<int, 3>{}, mine);
                    step(s0 + 4, std::integral_constant<int, 4>{}, mine);
                    step(s0 + 5, std::integral_constant<int, 5>{}, mine);
                    step(s0 + 6, std::integral_constant<int, 6>{}, mine);
                } else {
                    step(s0, std::integral_constant<int, 0>{}, mine);
                }
            }
            if (jwave) {                                   // the last step of the sweep; the frame is back in place afterwards
                apply_j(nsteps - 1, std::integral_constant<int, FU - 1>{});
                advance_frame();
            }
            if (mine) atomicAdd(&cnt, mine);               // once per sweep, off the per-step critical path
            __syncthreads();
            if (tid == 0) total += cnt;
            const int done = (cnt == 0);
            __syncthreads();
            if (done) break;
        }
        if (jwave && tid - 192 < NB) {
#pragma unroll
            for (int k = 0; k < NB; ++k) J[(tid - 192) * P + k] = jr[k];
        }
        __syncthreads();
    }
    if (tid == 0 && nrot_out) nrot_out[grp] = total;
    if (mode != 1 && total > 0) {
        // One Newton-Schulz step J <- J (3I - J^T J)/2: the accumulated product of ~n*sweeps plane rotations drifts
        // from orthogonality by ~sqrt(n*sweeps) eps; this squares the defect, so repeated application of J over many
        // Jacobi rounds does not inflate vector norms (singular values) beyond rounding.
        // Both products run on the matrix cores, each wave owning a strip of 16 x 16 output tiles.
        __shared__ double eigdiag[NB];
        typedef double d4e __attribute__((ext_vector_type(4)));
        constexpr int NT = NB / 16, TPW = NT * NT / 4;          // tiles per wave: 4 (NB = 64) or 1 (NB = 32)
        const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
        const int ti = (NB == 64) ? wave : (wave >> 1), tj0 = (NB == 64) ? 0 : (wave & 1);
        if (tid < NB) eigdiag[tid] = G[tid * P + tid];          // rotated diagonal = eigenvalues (needed below)
        __syncthreads();
        d4e acc[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = d4e{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < NB / 4; ++ks) {                   // S = J^T J
            const int k = ks * 4 + lk;
            const double fa = J[k * P + ti * 16 + li];
#pragma unroll
            for (int t = 0; t < TPW; ++t)
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, J[k * P + (tj0 + t) * 16 + li], acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) G[(ti * 16 + lk + 4 * r) * P + (tj0 + t) * 16 + li] = acc[t][r];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = d4e{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < NB / 4; ++ks) {                   // N = J S
            const int k = ks * 4 + lk;
            const double fa = J[(ti * 16 + li) * P + k];
#pragma unroll
            for (int t = 0; t < TPW; ++t)
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, G[k * P + (tj0 + t) * 16 + li], acc[t], 0, 0, 0);
        }
        double nv[TPW][4];
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                nv[t][r] = 1.5 * J[(ti * 16 + lk + 4 * r) * P + (tj0 + t) * 16 + li] - 0.5 * acc[t][r];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) J[(ti * 16 + lk + 4 * r) * P + (tj0 + t) * 16 + li] = nv[t][r];
        if (tid < NB) G[tid * P + tid] = eigdiag[tid];
        __syncthreads();
    }
    double* o = out + (int64_t)grp * nvec * nvec;
    for (int e = tid; e < nvec * nvec; e += 256) {
        const int i = e / nvec, j = e % nvec;
        double v = J[i * P + j];
        if (mode != 2) {
            const double di = dsc[i];
            v = (di == 0.0) ? 0.0 : v / di;
            if (dsc[j] == 0.0) v = 0.0;
        }
        o[e] = v;
    }
}

// ------------------------------------------------------------------------------------------ eig_small, pipelined form
// Same mathematics as eig_small_kernel (parallel-order two-sided Jacobi on the <= 64 x 64 Gram matrix, J accumulated in the
// registers of one wave), reorganised so that a step costs ONE barrier and nothing serial sits between barriers:
//   * G is double-buffered: the update of step s reads G[cur] and writes G[nxt] (waves 1 and 2, HP*HP/128 independent 2 x 2
//     blocks per thread);
//   * wave 0 decides the rotations of step s+1 DURING step s: the three entries a rotation depends on (g_pp, g_qq, g_pq of its
//     pair in step s+1) are single outputs of three 2 x 2 blocks of step s, which wave 0 evaluates itself from G[cur] and the
//     rotations of step s with the very operations of the update (rot_p / rot_q below: explicit mul + fma, so the values are
//     bit-identical to what the update stores) -- the chain of reciprocal square roots that used to sit between two barriers of
//     every step now runs beside the update;
//   * wave 3 applies step s to its register rows of J in the same phase (no lag any more).
// A step without any rotation (flag decided together with the parameters) skips the update and keeps the buffers.
__device__ __forceinline__ double rot_p(double c, double s, double x, double y) { return __fma_rn(-s, y, __dmul_rn(c, x)); }   // c x - s y
__device__ __forceinline__ double rot_q(double c, double s, double x, double y) { return __fma_rn(s, x, __dmul_rn(c, y)); }    // s x + c y
__device__ __forceinline__ double rsqrt2n(double x) {      // hardware seed (~2^-26) + two Newton steps: full double accuracy
    double r = __builtin_amdgcn_rsq(x);
    r = r * (1.5 - 0.5 * x * r * r);
    r = r * (1.5 - 0.5 * x * r * r);
    return r;
}

template <int NB>
__global__ __launch_bounds__(256) void eig_small2_kernel(const double* __restrict__ part, int nchunk, int nvec, int mode,
                                                         int max_sweeps, double dead_thresh,
                                                         double* __restrict__ out, int* __restrict__ dead,
                                                         int* __restrict__ nrot_out, double* __restrict__ maxoff_out,
                                                         double relevant2) {
    constexpr int P = NB + 1;
    __shared__ double Gb[2][NB * P];
    __shared__ double J[NB * P];
    __shared__ double dsc[NB];
    __shared__ int cnt, total;
    __shared__ int stepflag[2];
    __shared__ double red[256];
    const int tid = threadIdx.x, grp = blockIdx.x;
    const double* pg = part + (int64_t)grp * nchunk * nvec * nvec;
    double* G = Gb[0];
    {
        constexpr int EPT = NB * NB / 256;
        double acc[EPT];
#pragma unroll
        for (int t = 0; t < EPT; ++t) acc[t] = 0.0;
        for (int c = 0; c < nchunk; c += 2) {
            double v0[EPT], v1[EPT];
            const bool two = (c + 1 < nchunk);
#pragma unroll
            for (int t = 0; t < EPT; ++t) {
                const int e = tid + 256 * t, i = e / NB, j = e % NB;
                const bool in = (i < nvec && j < nvec);
                const int64_t o = (int64_t)c * nvec * nvec + i * nvec + j;
                v0[t] = in ? pg[o] : 0.0;
                v1[t] = (in && two) ? pg[o + (int64_t)nvec * nvec] : 0.0;
            }
#pragma unroll
            for (int t = 0; t < EPT; ++t) acc[t] = (acc[t] + v0[t]) + v1[t];
        }
#pragma unroll
        for (int t = 0; t < EPT; ++t) {
            const int e = tid + 256 * t, i = e / NB, j = e % NB;
            const bool in = (i < nvec && j < nvec);
            G[i * P + j] = in ? acc[t] : (i == j ? 1.0 : 0.0);
            J[i * P + j] = (i == j) ? 1.0 : 0.0;
        }
    }
    if (tid == 0) total = 0;
    __syncthreads();
    if (mode != 2) {
        if (tid < NB) {
            const double gii = G[tid * P + tid];
            const bool ok = (gii > dead_thresh) && (gii < 1.7e308);
            dsc[tid] = ok ? sqrt(gii) : 1.0;
            if (tid < nvec && dead) dead[grp * nvec + tid] = ok ? 0 : 1;
            if (!ok) dsc[tid] = 0.0;
        }
        __syncthreads();
        for (int e = tid; e < NB * NB; e += 256) {
            const int i = e / NB, j = e % NB;
            const double di = dsc[i], dj = dsc[j];
            double g;
            if (di == 0.0 || dj == 0.0) g = (i == j) ? 1.0 : 0.0;
            else g = (i == j) ? 1.0 : G[i * P + j] * fast_rcp(di * dj);
            G[i * P + j] = g;
        }
        __syncthreads();
    }
    {
        __shared__ double rdg[NB];
        if (tid < NB) {
            const double gii = fabs(G[tid * P + tid]);
            rdg[tid] = (gii > relevant2 && tid < nvec) ? fast_rcp(gii) : 0.0;
        }
        __syncthreads();
        double m = 0.0;
#pragma unroll
        for (int t = 0; t < NB * NB / 256; ++t) {
            const int e = tid + 256 * t, i = e / NB, j = e % NB;
            const double g = G[i * P + j];
            const double r2 = (i < j) ? g * g * rdg[i] * rdg[j] : 0.0;
            m = r2 > m ? r2 : m;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
        if ((tid & 63) == 0) red[tid >> 6] = m;
        __syncthreads();
        if (tid == 0) {
            const double mm = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
            red[0] = sqrt(mm);
            if (maxoff_out) maxoff_out[grp] = red[0];
        }
        __syncthreads();
    }
    const bool need = (mode != 1) && (nvec >= 2) && (red[0] > 8.881784197001252e-16);
    int cur = 0;
    if (need) {
        constexpr int HP = NB / 2;
        constexpr int M = NB - 1;                          // steps per sweep
        constexpr int UB = HP * HP / 128;                  // 2 x 2 blocks per thread of the two updating waves
        __shared__ __attribute__((aligned(16))) double cs2[2][HP * 2];
        __shared__ __attribute__((aligned(16))) double csj[2][HP * 2];
        const double tol = 8.881784197001252e-16;          // 2^-50
        auto slot_pair = [](int s, int a, int& p, int& q) {
            if (a == 0) { p = NB - 1; q = s; }
            else { p = s + a; p -= (p >= M) ? M : 0; q = s - a; q += (q < 0) ? M : 0; }
            if (p > q) { const int t = p; p = q; q = t; }
        };
        // slot of index i in step s, and whether i is the larger member of its pair
        auto slot_of = [&](int s, int i, int& a, int& hi) {
            if (i == NB - 1 || i == s) a = 0;
            else { int d = i - s; d += (d < 0) ? M : 0; a = (d <= HP - 1) ? d : M - d; }
            int p, q;
            slot_pair(s, a, p, q);
            hi = (i == q) ? 1 : 0;
        };
        // rotation of pair (p < q) from g_pp, g_qq, g_pq (the smaller-angle rotation, as in eig_small_kernel)
        auto decide = [&](double gpp, double gqq, double gpq, double& c, double& sn) -> bool {
            c = 1.0; sn = 0.0;
            const double g2 = gpq * gpq;
            if (g2 > tol * tol * fabs(gpp * gqq)) {
                const double d = gqq - gpp;
                const double rh = rsqrt2n(d * d + 4.0 * g2);
                const double c2 = 0.5 + 0.5 * fabs(d) * rh;
                const double rcv = rsqrt2n(c2);
                const double sabs = fabs(gpq) * rh * rcv;
                if (sabs <= 1.0 && c2 <= 1.0000000000000002) {
                    c = c2 * rcv;
                    sn = ((d >= 0.0) == (gpq >= 0.0)) ? sabs : -sabs;
                    return true;
                }
            }
            return false;
        };
        // wave 0, lane a < HP: parameters of step s into buffer `pb` (the buffers alternate from step to step ACROSS sweeps: a sweep
        // has an odd number of steps, so the parity of s itself would collide at the sweep boundary)
        auto publish = [&](int pb, int s, int a, int p, double c, double sn, bool rot) {
            *reinterpret_cast<double2*>(&cs2[pb][2 * a]) = make_double2(c, sn);
            int up = s + a; up -= (up >= M) ? M : 0;
            const bool flipped = (a > 0) && (up != p);
            *reinterpret_cast<double2*>(&csj[pb][2 * a]) = make_double2(c, flipped ? -sn : sn);
            const unsigned long long any = __ballot(rot);
            if (a == 0) stepflag[pb] = (any != 0ull) ? 1 : 0;
        };
        const bool jwave = (tid >= 192);
        double jr[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) jr[k] = (k == tid - 192) ? 1.0 : 0.0;
        constexpr int FU = (NB == 64) ? 7 : 1;
        static_assert(M % FU == 0, "group length must divide the sweep");
        auto apply_j = [&](int pb, auto uc) {
            constexpr int u = decltype(uc)::value;
            double2 r[HP];
#pragma unroll
            for (int a = 0; a < HP; ++a) r[a] = *reinterpret_cast<const double2*>(&csj[pb][2 * a]);
            {
                const double x = jr[u], y = jr[NB - 1];
                jr[u] = r[0].x * x - r[0].y * y;
                jr[NB - 1] = r[0].y * x + r[0].x * y;
            }
#pragma unroll
            for (int a = 1; a < HP; ++a) {
                const int px = (u + a) % M, py = (u - a + M) % M;
                const double x = jr[px], y = jr[py];
                jr[px] = r[a].x * x - r[a].y * y;
                jr[py] = r[a].y * x + r[a].x * y;
            }
        };
        auto advance_frame = [&]() {
            double tmp[FU];
#pragma unroll
            for (int f = 0; f < FU; ++f) tmp[f] = jr[f];
#pragma unroll
            for (int f = 0; f < NB - 1 - FU; ++f) jr[f] = jr[f + FU];
#pragma unroll
            for (int f = 0; f < FU; ++f) jr[NB - 1 - FU + f] = tmp[f];
        };
        int mine = 0;                                      // (wave 0) rotations decided for the sweep in progress
        int mine_next = 0;                                 // ... for the next sweep (the look-ahead of a sweep's last step)
        // parameters of step 0 straight from G
        if (tid < 64) {
            bool rot = false;
            double c = 1.0, sn = 0.0;
            int p = 0, q = 0;
            if (tid < HP) {
                slot_pair(0, tid, p, q);
                rot = decide(G[p * P + p], G[q * P + q], G[p * P + q], c, sn);
                if (rot) ++mine;
            }
            if (tid < HP) publish(0, 0, tid, p, c, sn, rot);
            else (void)__ballot(false);
        }
        __syncthreads();
        int par = 0;                                       // parameter buffer of the step in progress
        // one step of the tournament: phase of step s (u = s mod FU compile-time for the register frame of J)
        auto step = [&](int s, auto uc) -> void {
            constexpr int u = decltype(uc)::value;
            const int flag = stepflag[par];                 // uniform
            const double* Gc = Gb[cur];
            double* Gn = Gb[cur ^ 1];
            if (tid < 64) {
                // ---- wave 0: rotations of step s+1 from G[cur] and the rotations of step s
                const int s1 = (s + 1 == M) ? 0 : s + 1;
                bool rot = false;
                double c = 1.0, sn = 0.0;
                int p = 0, q = 0;
                if (tid < HP) {
                    slot_pair(s1, tid, p, q);
                    double gpp, gqq, gpq;
                    if (flag) {
                        int ap, hp_, aq, hq_;
                        slot_of(s, p, ap, hp_);
                        slot_of(s, q, aq, hq_);
                        int pp, qp, pq, qq;
                        slot_pair(s, ap, pp, qp);           // pair holding p in step s
                        slot_pair(s, aq, pq, qq);           // pair holding q in step s
                        const double2 rp = *reinterpret_cast<const double2*>(&cs2[par][2 * ap]);
                        const double2 rq = *reinterpret_cast<const double2*>(&cs2[par][2 * aq]);
                        // block (ap, ap) -> g_pp;  block (aq, aq) -> g_qq;  block (ap, aq) -> g_pq
                        const double a00 = Gc[pp * P + pp], a01 = Gc[pp * P + qp], a10 = Gc[qp * P + pp], a11 = Gc[qp * P + qp];
                        const double b00 = Gc[pq * P + pq], b01 = Gc[pq * P + qq], b10 = Gc[qq * P + pq], b11 = Gc[qq * P + qq];
                        const double x00 = Gc[pp * P + pq], x01 = Gc[pp * P + qq], x10 = Gc[qp * P + pq], x11 = Gc[qp * P + qq];
                        {
                            const double t0 = hp_ ? rot_q(rp.x, rp.y, a00, a10) : rot_p(rp.x, rp.y, a00, a10);
                            const double t1 = hp_ ? rot_q(rp.x, rp.y, a01, a11) : rot_p(rp.x, rp.y, a01, a11);
                            gpp = hp_ ? rot_q(rp.x, rp.y, t0, t1) : rot_p(rp.x, rp.y, t0, t1);
                        }
                        {
                            const double t0 = hq_ ? rot_q(rq.x, rq.y, b00, b10) : rot_p(rq.x, rq.y, b00, b10);
                            const double t1 = hq_ ? rot_q(rq.x, rq.y, b01, b11) : rot_p(rq.x, rq.y, b01, b11);
                            gqq = hq_ ? rot_q(rq.x, rq.y, t0, t1) : rot_p(rq.x, rq.y, t0, t1);
                        }
                        {
                            const double t0 = hp_ ? rot_q(rp.x, rp.y, x00, x10) : rot_p(rp.x, rp.y, x00, x10);
                            const double t1 = hp_ ? rot_q(rp.x, rp.y, x01, x11) : rot_p(rp.x, rp.y, x01, x11);
                            gpq = hq_ ? rot_q(rq.x, rq.y, t0, t1) : rot_p(rq.x, rq.y, t0, t1);
                        }
                    } else {
                        gpp = Gc[p * P + p]; gqq = Gc[q * P + q]; gpq = Gc[p * P + q];
                    }
                    rot = decide(gpp, gqq, gpq, c, sn);
                    if (rot) { if (s + 1 == M) ++mine_next; else ++mine; }
                }
                if (tid < HP) publish(par ^ 1, s1, tid, p, c, sn, rot);
                else (void)__ballot(false);
            } else if (jwave) {
                if (flag) apply_j(par, uc);
                if constexpr (u == FU - 1) advance_frame();
            } else if (flag) {
                // ---- waves 1 and 2: G[nxt] <- R^T G[cur] R, block (a, b) = rows of slot a x columns of slot b
                const int t128 = tid - 64;
                const int tb = t128 % HP, t0 = t128 / HP;
                constexpr int TS = 128 / HP;
                int pb, qb;
                slot_pair(s, tb, pb, qb);
                const double2 rb = *reinterpret_cast<const double2*>(&cs2[par][2 * tb]);
                double g00[UB], g01[UB], g10[UB], g11[UB];
                int pa[UB], qa[UB];
                double2 ra[UB];
#pragma unroll
                for (int k = 0; k < UB; ++k) {
                    slot_pair(s, t0 + TS * k, pa[k], qa[k]);
                    ra[k] = *reinterpret_cast<const double2*>(&cs2[par][2 * (t0 + TS * k)]);
                    g00[k] = Gc[pa[k] * P + pb]; g01[k] = Gc[pa[k] * P + qb];
                    g10[k] = Gc[qa[k] * P + pb]; g11[k] = Gc[qa[k] * P + qb];
                }
#pragma unroll
                for (int k = 0; k < UB; ++k) {
                    const double ca = ra[k].x, sa = ra[k].y;
                    const double t00 = rot_p(ca, sa, g00[k], g10[k]), t01 = rot_p(ca, sa, g01[k], g11[k]);
                    const double t10 = rot_q(ca, sa, g00[k], g10[k]), t11 = rot_q(ca, sa, g01[k], g11[k]);
                    Gn[pa[k] * P + pb] = rot_p(rb.x, rb.y, t00, t01); Gn[pa[k] * P + qb] = rot_q(rb.x, rb.y, t00, t01);
                    Gn[qa[k] * P + pb] = rot_p(rb.x, rb.y, t10, t11); Gn[qa[k] * P + qb] = rot_q(rb.x, rb.y, t10, t11);
                }
            }
            __syncthreads();
            cur ^= flag;
            par ^= 1;
        };
        for (int sweep = 0; sweep < max_sweeps; ++sweep) {
            if (tid == 0) cnt = 0;
            for (int s0 = 0; s0 < M; s0 += FU) {
                if constexpr (FU == 7) {
                    step(s0 + 0, std::integral_constant<int, 0>{});
                    step(s0 + 1, std::integral_constant<int, 1>{});
                    step(s0 + 2, std::integral_constant<int, 2>{});
                    step(s0 + 3, std::integral_constant<int, 3>{});
                    step(s0 + 4, std::integral_constant<int, 4>{});
                    step(s0 + 5, std::integral_constant<int, 5>{});
                    step(s0 + 6, std::integral_constant<int, 6>{});
                } else {
                    step(s0, std::integral_constant<int, 0>{});
                }
            }
            if (mine) atomicAdd(&cnt, mine);
            __syncthreads();
            if (tid == 0) total += cnt;
            const int done = (cnt == 0);
            __syncthreads();
            mine = mine_next;
            mine_next = 0;
            if (done) break;
        }
        if (jwave && tid - 192 < NB) {
#pragma unroll
            for (int k = 0; k < NB; ++k) J[(tid - 192) * P + k] = jr[k];
        }
        __syncthreads();
    }
    double* Gf = Gb[cur];                                  // the rotated Gram matrix
    double* Gs = Gb[cur ^ 1];                              // scratch for the Newton-Schulz step
    if (tid == 0 && nrot_out) nrot_out[grp] = total;
    if (mode != 1 && total > 0) {
        typedef double d4e __attribute__((ext_vector_type(4)));
        constexpr int NT = NB / 16, TPW = NT * NT / 4;
        const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
        const int ti = (NB == 64) ? wave : (wave >> 1), tj0 = (NB == 64) ? 0 : (wave & 1);
        d4e acc[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = d4e{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < NB / 4; ++ks) {                   // S = J^T J
            const int k = ks * 4 + lk;
            const double fa = J[k * P + ti * 16 + li];
#pragma unroll
            for (int t = 0; t < TPW; ++t)
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, J[k * P + (tj0 + t) * 16 + li], acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) Gs[(ti * 16 + lk + 4 * r) * P + (tj0 + t) * 16 + li] = acc[t][r];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = d4e{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < NB / 4; ++ks) {                   // N = J S
            const int k = ks * 4 + lk;
            const double fa = J[(ti * 16 + li) * P + k];
#pragma unroll
            for (int t = 0; t < TPW; ++t)
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, Gs[k * P + (tj0 + t) * 16 + li], acc[t], 0, 0, 0);
        }
        double nv[TPW][4];
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                nv[t][r] = 1.5 * J[(ti * 16 + lk + 4 * r) * P + (tj0 + t) * 16 + li] - 0.5 * acc[t][r];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) J[(ti * 16 + lk + 4 * r) * P + (tj0 + t) * 16 + li] = nv[t][r];
        __syncthreads();
    }
    (void)Gf;
    double* o = out + (int64_t)grp * nvec * nvec;
    for (int e = tid; e < nvec * nvec; e += 256) {
        const int i = e / nvec, j = e % nvec;
        double v = J[i * P + j];
        if (mode != 2) {
            const double di = dsc[i];
            v = (di == 0.0) ? 0.0 : v / di;
            if (dsc[j] == 0.0) v = 0.0;
        }
        o[e] = v;
    }
}

// Jacobi rotation of a pair (p < q) from g_pp, g_qq, g_pq: the smaller-angle rotation, (c, sn) with G' = R^T G R, R = [[c, sn], [-sn, c]];
// false (c = 1, sn = 0) when |g_pq| <= tol sqrt(|g_pp g_qq|)  (the rule of eig_small3_kernel's `decide`)
__device__ __forceinline__ bool eig_decide(double gpp, double gqq, double gpq, double tol, double& c, double& sn) {
    c = 1.0; sn = 0.0;
    const double g2 = gpq * gpq;
    if (g2 > tol * tol * fabs(gpp * gqq)) {
        const double d = gqq - gpp;
        const double rh = rsqrt2n(d * d + 4.0 * g2);
        const double c2 = 0.5 + 0.5 * fabs(d) * rh;
        const double rcv = rsqrt2n(c2);
        const double sabs = fabs(gpq) * rh * rcv;
        if (sabs <= 1.0 && c2 <= 1.0000000000000002) {
            c = c2 * rcv;
            sn = ((d >= 0.0) == (gpq >= 0.0)) ? sabs : -sabs;
            return true;
        }
    }
    return false;
}

// ------------------------------------------------------------------------------------------ eig_small, third form
// The pipelined kernel above is bound by the instruction streams of its single waves (a wave issues one fp64 instruction per 4
// cycles: 128 for the update of a quarter of G, 128 for J, ~70 dependent ones for the rotation parameters), not by its barrier.
// Here the workgroup has 8 waves: wave 0 decides the next step's rotations, wave 3 turns its register rows of J, and FIVE waves
// (1, 2, 4, 5, 6) update G over the 2 x 2 blocks of the upper triangle only (528 instead of 1024 for a 64 x 64 matrix), writing
// every block and its mirror image, so that G stays exactly symmetric and a thread has two blocks per step instead of eight.
// -DTN_CLOCKS: thread 0 of group 0 of the persistent SVD kernel accumulates the 100 MHz wall clock per phase of the eigenproblem:
// [0] partial sums -> G  [1] measure + fast path  [2] cyclic sweeps  [3] Newton-Schulz  [4] store  [5] calls  [6] calls with cyclic sweeps
#ifdef TN_CLOCKS
__device__ long long eig_clk[8];
}  // namespace tn
extern "C" int tn_debug_eig_clocks(long long* host, int reset) {
    int rc = (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(tn::eig_clk), sizeof(long long) * 8);
    if (reset) { long long z[8] = {}; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(tn::eig_clk), z, sizeof(z)); }
    return rc;
}
namespace tn {
#define EIG_CLK(k) do { if (COH && threadIdx.x == 0 && grp == 0) { const long long now_ = wall_clock64(); eig_clk[k] += now_ - eclk_last_; eclk_last_ = now_; } } while (0)
#define EIG_CNT(k) do { if (COH && threadIdx.x == 0 && grp == 0) eig_clk[k] += 1; } while (0)
#define EIG_CLK_INIT long long eclk_last_ = wall_clock64()
#else
#define EIG_CLK(k) do {} while (0)
#define EIG_CNT(k) do {} while (0)
#define EIG_CLK_INIT do {} while (0)
#endif
// (The body is a device function: the standalone kernel below runs it once per launch, the persistent SVD kernels once per Jacobi
// round.  pool: LDS, 4 x NB x (NB + 1) doubles (the two copies of G, J, and the fast path's R), supplied by the caller, who may use
// it for something else between calls.  COH: the partial sums / the
// results are exchanged with other workgroups of the SAME launch -- agent-scope accesses.)
template <int NB, bool COH = false>
__device__ __forceinline__ void eig_small3_body(double* pool, const int grp, const double* part, int nchunk, int nvec, int mode,
                                                int max_sweeps, double dead_thresh, double* out, int* dead, int* nrot_out, double* maxoff_out,
                                                double relevant2, int dbg, double fast_thr) {
    constexpr int P = NB + 1;
    auto ldc = [](const double* q) -> double {
        if constexpr (COH) return __hip_atomic_load((const __attribute__((address_space(1))) double*)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else return *q;
    };
    auto stc = [](double* q, double v) {
        if constexpr (COH) __hip_atomic_store((__attribute__((address_space(1))) double*)q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *q = v;
    };
    double (*Gb)[NB * P] = reinterpret_cast<double (*)[NB * P]>(pool);
    double* J = pool + 2 * NB * P;
    __shared__ double dsc[NB];
    __shared__ int cnt, total;
    __shared__ int stepflag[2];
    __shared__ double red[512];
    constexpr int NT_ = 512;
    const int tid = threadIdx.x;
    EIG_CLK_INIT;
    EIG_CNT(5);
    const double* pg = part + (int64_t)grp * nchunk * nvec * nvec;
    double* G = Gb[0];
    {
        constexpr int EPT = NB * NB / NT_;
        double acc[EPT];
#pragma unroll
        for (int t = 0; t < EPT; ++t) acc[t] = 0.0;
        // four partial sums per memory round trip, added in chunk order (an absent one adds +0.0, which changes nothing: acc is never -0.0).
        // Full pairs of the SVD (64 vectors): only the ten 16 x 16 tiles on and above the diagonal are read (dealt to the threads tile by
        // tile, 16 consecutive elements of a row per 16 lanes) and an element below them is the mirror image of its transpose -- the
        // same products summed in the same order, bit for bit -- so the persistent SVD kernel need not publish the lower tiles at all
        // (a third of what its workgroups exchange).
        const bool upper_only = (NB == 64) && nvec == 64 && mode == 2;
        const int64_t nn = (int64_t)nvec * nvec;
        if (upper_only) {
            constexpr int UPT = 5;                           // 10 tiles x 256 elements over 512 threads
            double au[UPT];
            int ui[UPT], uj[UPT];
#pragma unroll
            for (int t = 0; t < UPT; ++t) {
                const int u = tid + NT_ * t, q = u >> 8, r = (u >> 4) & 15, c2 = u & 15;
                // tile q of the row-by-row list (0,0) (0,1) (0,2) (0,3) (1,1) (1,2) (1,3) (2,2) (2,3) (3,3)
                const int ti = q < 4 ? 0 : q < 7 ? 1 : q < 9 ? 2 : 3, tj = q < 4 ? q : q < 7 ? q - 3 : q < 9 ? q - 5 : 3;
                ui[t] = ti * 16 + r; uj[t] = tj * 16 + c2;
                au[t] = 0.0;
            }
            for (int c = 0; c < nchunk; c += 4) {
                double v0[UPT], v1[UPT], v2[UPT], v3[UPT];
                const bool two = (c + 1 < nchunk), three = (c + 2 < nchunk), four = (c + 3 < nchunk);
#pragma unroll
                for (int t = 0; t < UPT; ++t) {
                    const int64_t o = (int64_t)c * nn + ui[t] * nvec + uj[t];
                    v0[t] = ldc(pg + o);
                    v1[t] = two ? ldc(pg + o + nn) : 0.0;
                    v2[t] = three ? ldc(pg + o + 2 * nn) : 0.0;
                    v3[t] = four ? ldc(pg + o + 3 * nn) : 0.0;
                }
#pragma unroll
                for (int t = 0; t < UPT; ++t) au[t] = (((au[t] + v0[t]) + v1[t]) + v2[t]) + v3[t];
            }
#pragma unroll
            for (int t = 0; t < UPT; ++t) {
                G[ui[t] * P + uj[t]] = au[t];
                if ((ui[t] >> 4) != (uj[t] >> 4)) G[uj[t] * P + ui[t]] = au[t];
            }
#pragma unroll
            for (int t = 0; t < EPT; ++t) {
                const int e = tid + NT_ * t, i = e / NB, j = e % NB;
                J[i * P + j] = (i == j) ? 1.0 : 0.0;
            }
        } else {
        for (int c = 0; c < nchunk; c += 4) {
            double v0[EPT], v1[EPT], v2[EPT], v3[EPT];
            const bool two = (c + 1 < nchunk), three = (c + 2 < nchunk), four = (c + 3 < nchunk);
#pragma unroll
            for (int t = 0; t < EPT; ++t) {
                const int e = tid + NT_ * t, i = e / NB, j = e % NB;
                const bool in = (i < nvec && j < nvec);
                const int64_t o = (int64_t)c * nn + i * nvec + j;
                v0[t] = in ? ldc(pg + o) : 0.0;
                v1[t] = (in && two) ? ldc(pg + o + nn) : 0.0;
                v2[t] = (in && three) ? ldc(pg + o + 2 * nn) : 0.0;
                v3[t] = (in && four) ? ldc(pg + o + 3 * nn) : 0.0;
            }
#pragma unroll
            for (int t = 0; t < EPT; ++t) acc[t] = (((acc[t] + v0[t]) + v1[t]) + v2[t]) + v3[t];
        }
#pragma unroll
        for (int t = 0; t < EPT; ++t) {
            const int e = tid + NT_ * t, i = e / NB, j = e % NB;
            const bool in = (i < nvec && j < nvec);
            G[i * P + j] = in ? acc[t] : (i == j ? 1.0 : 0.0);
            J[i * P + j] = (i == j) ? 1.0 : 0.0;
        }
        }
    }
    if (tid == 0) total = 0;
    __syncthreads();
    if (mode != 2) {
        if (tid < NB) {
            const double gii = G[tid * P + tid];
            const bool ok = (gii > dead_thresh) && (gii < 1.7e308);
            dsc[tid] = ok ? sqrt(gii) : 1.0;
            if (tid < nvec && dead) dead[grp * nvec + tid] = ok ? 0 : 1;
            if (!ok) dsc[tid] = 0.0;
        }
        __syncthreads();
        for (int e = tid; e < NB * NB; e += NT_) {
            const int i = e / NB, j = e % NB;
            const double di = dsc[i], dj = dsc[j];
            double g;
            if (di == 0.0 || dj == 0.0) g = (i == j) ? 1.0 : 0.0;
            else g = (i == j) ? 1.0 : G[i * P + j] * fast_rcp(di * dj);
            G[i * P + j] = g;
        }
        __syncthreads();
    }
    __shared__ double rdg[NB];
    // largest relative off-diagonal |g_ij| / sqrt(g_ii g_jj) among the relevant vectors -> red[0] (ends with a barrier)
    auto measure = [&](const double* Gm) {
        if (tid < NB) {
            const double gii = fabs(Gm[tid * P + tid]);
            rdg[tid] = (gii > relevant2 && tid < nvec) ? fast_rcp(gii) : 0.0;
        }
        __syncthreads();
        double m = 0.0;
#pragma unroll
        for (int t = 0; t < NB * NB / NT_; ++t) {
            const int e = tid + NT_ * t, i = e / NB, j = e % NB;
            const double g = Gm[i * P + j];
            const double r2 = (i < j) ? g * g * rdg[i] * rdg[j] : 0.0;
            m = r2 > m ? r2 : m;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
        if ((tid & 63) == 0) red[tid >> 6] = m;
        __syncthreads();
        if (tid == 0) {
            const double mm = fmax(fmax(fmax(red[0], red[1]), fmax(red[2], red[3])), fmax(fmax(red[4], red[5]), fmax(red[6], red[7])));
            red[0] = sqrt(mm);
        }
        __syncthreads();
    };
    EIG_CLK(0);
    measure(G);
    if (tid == 0 && maxoff_out) stc(maxoff_out + grp, red[0]);
    // ---- near-diagonal fast path (SVD pair step, 64 x 64): in the quadratic regime -- every relative off-diagonal below fast_thr, which
    // is what a pair meets from the second outer sweep on -- the 2 x 63 dependent steps of the cyclic sweeps are replaced by Newton-like
    // steps on the whole matrix: K = the antisymmetric matrix of ALL Jacobi angles of the current G (K_pq = sin theta_pq), R = I + K made
    // orthogonal by one Newton-Schulz step (= exp(K) to second order), G <- R^T G R and J <- J R on the matrix cores: five 64^3 products
    // per step instead of 63 barriers.  Off-diagonals fall quadratically (1e-4 -> 1e-8 -> 1e-16).  The step is only taken while
    // ||K||_F stays small (nearly equal diagonal entries can ask for large angles: those belong to the cyclic sweep) and while it pays
    // (the largest off-diagonal must shrink fourfold); otherwise the cyclic sweeps below take over from the current G and J.  G only
    // steers: J stays orthogonal to rounding (Newton-Schulz here and at the end), and the caller measures convergence on the vectors.
    int fast_rot = 0;
    bool fast_done = false;
    // (Gc: the current matrix, red[0]: its measure; Wk: scratch of the same size.  Leaves red[0] = the measure of what it leaves in Gc.)
    auto fast_try = [&](double* Gc, double* Wk) {
        if constexpr (NB == 64) {
            double* Rb = pool + 3 * NB * P;
            typedef double d4f __attribute__((ext_vector_type(4)));
            const int w8 = tid >> 6, ln = tid & 63, li = ln & 15, lk = ln >> 4;
            const int ti = w8 >> 1, tj0 = (w8 & 1) * 2;
            int rot_here = 0;
            // acc[t] = op(A) B for this wave's two 16 x 16 tiles (row block ti, column blocks tj0, tj0 + 1)
            auto mm64 = [&](const double* Am, bool at, const double* Bm, d4f (&acc)[2]) {
                acc[0] = d4f{0.0, 0.0, 0.0, 0.0};
                acc[1] = acc[0];
#pragma unroll
                for (int ks = 0; ks < NB / 4; ++ks) {
                    const int k = ks * 4 + lk;
                    const double fa = at ? Am[k * P + ti * 16 + li] : Am[(ti * 16 + li) * P + k];
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, Bm[k * P + (tj0 + t) * 16 + li], acc[t], 0, 0, 0);
                }
            };
            auto put = [&](double* Dm, const d4f (&acc)[2]) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Dm[(ti * 16 + lk + 4 * r) * P + (tj0 + t) * 16 + li] = acc[t][r];
            };
            const double tolf = 8.881784197001252e-16;
            double m_prev = 0.0;
            for (int it = 0; it < 5 && !fast_done; ++it) {
                // relevance from the CURRENT diagonal: a pair takes part when at least one of its vectors is relevant.  Two vectors
                // below the threshold are both discarded by the caller whatever they look like, and their mutual angles are noise
                // (large: they alone would push ||K|| over the limit); but a vector that has just dropped below the threshold must still be
                // cleaned of its components along the kept ones, or it is not small at all.  Done = no such pair asks for a rotation,
                // the criterion of the cyclic sweeps.
                if (tid < NB) {
                    const double gii = fabs(Gc[tid * P + tid]);
                    rdg[tid] = (gii > relevant2 && tid < nvec) ? 1.0 : 0.0;
                }
                __syncthreads();
                double k2 = 0.0, mw = 0.0;
                int nrl = 0;
                for (int e = tid; e < NB * NB; e += NT_) {
                    const int i = e / NB, j = e % NB;
                    double v = (i == j) ? 1.0 : 0.0;
                    if (i != j) {
                        const int p = i < j ? i : j, q = i < j ? j : i;
                        double c, sn;
                        v = 0.0;
                        if (rdg[p] + rdg[q] > 0.0) {
                            const double gpp = Gc[p * P + p], gqq = Gc[q * P + q], gpq = Gc[p * P + q];
                            if (eig_decide(gpp, gqq, gpq, tolf, c, sn)) {
                                v = (i < j) ? sn : -sn; k2 += sn * sn; ++nrl;
                                const double r2 = gpq * gpq / fabs(gpp * gqq);
                                mw = r2 > mw ? r2 : mw;
                            }
                        }
                    }
                    Rb[i * P + j] = v;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { k2 += __shfl_xor(k2, o, 64); nrl += __shfl_xor(nrl, o, 64); mw = fmax(mw, __shfl_xor(mw, o, 64)); }
                __syncthreads();
                if (ln == 0) { red[8 + w8] = k2; red[16 + w8] = (double)nrl; red[24 + w8] = mw; }
                __syncthreads();
                double ksum = 0.0, nsum = 0.0, mwide = 0.0;
#pragma unroll
                for (int w = 0; w < 8; ++w) { ksum += red[8 + w]; nsum += red[16 + w]; mwide = fmax(mwide, red[24 + w]); }
                mwide = sqrt(mwide);
                if (tid == 0) red[0] = mwide;                      // what the launch leaves (read after the loop)
                if (nsum == 0.0) { fast_done = true; break; }      // nothing left to rotate
                if (!(ksum <= 0.02)) break;                        // ||K||_F^2 > 0.02 (or NaN): the cyclic sweeps take over
                if (it > 0 && !(mwide <= 0.25 * m_prev)) break;    // the steps do not pay: cyclic sweeps from here
                m_prev = mwide;
                d4f acc[2];
                // R = I + K is orthogonal to ||K||^2; a Newton-Schulz step takes an error d to 3/4 d^2.  J must leave this launch
                // orthogonal to ~1e-8 (its own closing Newton-Schulz step squares that), so: as many steps as ||K|| asks for.
                const int nns = ksum <= 3e-8 ? 1 : ksum <= 1e-4 ? 2 : 3;
                for (int ns = 0; ns < nns; ++ns) {
                    mm64(Rb, true, Rb, acc);                       // S = R^T R
                    put(Wk, acc);
                    __syncthreads();
                    mm64(Rb, false, Wk, acc);                      // R S
                    d4f rn[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) rn[t][r] = 1.5 * Rb[(ti * 16 + lk + 4 * r) * P + (tj0 + t) * 16 + li] - 0.5 * acc[t][r];
                    __syncthreads();
                    put(Rb, rn);                                   // R <- 1.5 R - 0.5 R R^T R
                    __syncthreads();
                }
                mm64(Gc, false, Rb, acc);                          // W = G R
                put(Wk, acc);
                __syncthreads();
                mm64(Rb, true, Wk, acc);                           // R^T W
                d4f jn[2];
                mm64(J, false, Rb, jn);                            // J R
                __syncthreads();
                put(Gc, acc);
                put(J, jn);
                __syncthreads();
                for (int e = tid; e < NB * NB; e += NT_) {         // exactly symmetric again (the cyclic sweeps rely on it): the mean of both halves
                    const int i = e / NB, j = e % NB;
                    if (i < j) { const double v = 0.5 * (Gc[i * P + j] + Gc[j * P + i]); Gc[i * P + j] = v; Gc[j * P + i] = v; }
                }
                __syncthreads();
                rot_here += (int)(0.5 * nsum);
            }
            __syncthreads();
            fast_rot += rot_here;
            if (tid == 0) total += rot_here;
            __syncthreads();
        }
    };
    const bool fast_on = (NB == 64) && mode == 2 && nvec >= 2 && fast_thr > 0.0;
    if (fast_on && red[0] > 8.881784197001252e-16 && red[0] <= fast_thr) fast_try(Gb[0], Gb[1]);
    const bool need = (mode != 1) && (nvec >= 2) && !fast_done && (red[0] > 8.881784197001252e-16);
    EIG_CLK(1);
    if (need) EIG_CNT(6);
    int cur = 0;
    if (need) {
        constexpr int HP = NB / 2;
        constexpr int M = NB - 1;                          // steps per sweep
        constexpr int NBLK = HP * (HP + 1) / 2;            // 2 x 2 blocks (a <= b): G stays exactly symmetric, half the arithmetic
        constexpr int GT = 320;                            // threads of the five updating waves (1, 2, 4, 5, 6)
        constexpr int UB = (NBLK + GT - 1) / GT;           // blocks per updating thread
        __shared__ __attribute__((aligned(16))) double cs2[2][HP * 2];
        __shared__ __attribute__((aligned(16))) double csj[2][HP * 2];
        const double tol = 8.881784197001252e-16;          // 2^-50
        auto slot_pair = [](int s, int a, int& p, int& q) {
            if (a == 0) { p = NB - 1; q = s; }
            else { p = s + a; p -= (p >= M) ? M : 0; q = s - a; q += (q < 0) ? M : 0; }
            if (p > q) { const int t = p; p = q; q = t; }
        };
        // slot of index i in step s, and whether i is the larger member of its pair
        auto slot_of = [&](int s, int i, int& a, int& hi) {
            if (i == NB - 1 || i == s) a = 0;
            else { int d = i - s; d += (d < 0) ? M : 0; a = (d <= HP - 1) ? d : M - d; }
            int p, q;
            slot_pair(s, a, p, q);
            hi = (i == q) ? 1 : 0;
        };
        // rotation of pair (p < q) from g_pp, g_qq, g_pq (the smaller-angle rotation, as in eig_small_kernel)
        auto decide = [&](double gpp, double gqq, double gpq, double& c, double& sn) -> bool {
            c = 1.0; sn = 0.0;
            const double g2 = gpq * gpq;
            if (g2 > tol * tol * fabs(gpp * gqq)) {
                const double d = gqq - gpp;
                const double rh = rsqrt2n(d * d + 4.0 * g2);
                const double c2 = 0.5 + 0.5 * fabs(d) * rh;
                const double rcv = rsqrt2n(c2);
                const double sabs = fabs(gpq) * rh * rcv;
                if (sabs <= 1.0 && c2 <= 1.0000000000000002) {
                    c = c2 * rcv;
                    sn = ((d >= 0.0) == (gpq >= 0.0)) ? sabs : -sabs;
                    return true;
                }
            }
            return false;
        };
        __shared__ unsigned char blk_a[NBLK], blk_b[NBLK];
        // index tables (the modular arithmetic of the tournament once per launch instead of in every thread in every step; the kernel is
        // bound by instruction issue): pq_tab[s][a] = the pair of slot a in step s (p < q, packed p | q << 8); so_tab[s][i] = slot of
        // index i in step s, bit 7 set when i is the larger member of its pair.  (Fetching them one step ahead as well was tried: the
        // extra reads cost more than the shorter dependency chain gains.)
        __shared__ unsigned short pq_tab[M * HP];
        __shared__ unsigned char so_tab[M * NB];
        for (int e = tid; e < M * HP; e += NT_) {
            int p, q;
            slot_pair(e / HP, e % HP, p, q);
            pq_tab[e] = (unsigned short)(p | (q << 8));
        }
        for (int e = tid; e < M * NB; e += NT_) {
            int a, hi;
            slot_of(e / NB, e % NB, a, hi);
            so_tab[e] = (unsigned char)(a | (hi << 7));
        }
        for (int e = tid; e < NBLK; e += NT_) {            // block list: e -> (a, b), a <= b, row by row
            int a = 0, rem = e;
            while (rem >= HP - a) { rem -= HP - a; ++a; }
            blk_a[e] = (unsigned char)a;
            blk_b[e] = (unsigned char)(a + rem);
        }
        auto pair_of = [&](int s, int a, int& p, int& q) { const unsigned v = pq_tab[s * HP + a]; p = (int)(v & 255u); q = (int)(v >> 8); };
        // wave 0, lane a < HP: parameters of step s into buffer `pb` (the buffers alternate from step to step ACROSS sweeps: a sweep
        // has an odd number of steps, so the parity of s itself would collide at the sweep boundary)
        auto publish = [&](int pb, int s, int a, int p, double c, double sn, bool rot) {
            *reinterpret_cast<double2*>(&cs2[pb][2 * a]) = make_double2(c, sn);
            int up = s + a; up -= (up >= M) ? M : 0;
            const bool flipped = (a > 0) && (up != p);
            *reinterpret_cast<double2*>(&csj[pb][2 * a]) = make_double2(c, flipped ? -sn : sn);
            const unsigned long long any = __ballot(rot);
            if (a == 0) stepflag[pb] = (any != 0ull) ? 1 : 0;
        };
        const int wave_ = tid >> 6, lane_ = tid & 63;
        const int wave_g = (wave_ == 1) ? 0 : (wave_ == 2) ? 1 : (wave_ == 4) ? 2 : (wave_ == 5) ? 3 : (wave_ == 6) ? 4 : -1;
        const bool jwave = (wave_ == 3);
        double jr[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) jr[k] = (k == tid - 192) ? 1.0 : 0.0;
        if (fast_rot > 0 && jwave && tid - 192 < NB) {     // the fast path has turned J already: go on from there
#pragma unroll
            for (int k = 0; k < NB; ++k) jr[k] = J[(tid - 192) * P + k];
        }
        constexpr int FU = (NB == 64) ? 7 : 1;
        static_assert(M % FU == 0, "group length must divide the sweep");
        auto apply_j = [&](int pb, auto uc) {
            constexpr int u = decltype(uc)::value;
            constexpr int CH = (HP >= 16) ? 16 : HP;       // parameters fetched 16 pairs at a time (two waves per SIMD: 256 registers each)
#pragma unroll
            for (int a0 = 0; a0 < HP; a0 += CH) {
                double2 r[CH];
#pragma unroll
                for (int a = 0; a < CH; ++a) r[a] = *reinterpret_cast<const double2*>(&csj[pb][2 * (a0 + a)]);
#pragma unroll
                for (int a = 0; a < CH; ++a) {
                    const int aa = a0 + a;
                    const int px = (aa == 0) ? u : (u + aa) % M, py = (aa == 0) ? NB - 1 : (u - aa + M) % M;
                    const double x = jr[px], y = jr[py];
                    jr[px] = r[a].x * x - r[a].y * y;
                    jr[py] = r[a].y * x + r[a].x * y;
                }
            }
        };
        auto advance_frame = [&]() {
            double tmp[FU];
#pragma unroll
            for (int f = 0; f < FU; ++f) tmp[f] = jr[f];
#pragma unroll
            for (int f = 0; f < NB - 1 - FU; ++f) jr[f] = jr[f + FU];
#pragma unroll
            for (int f = 0; f < FU; ++f) jr[NB - 1 - FU + f] = tmp[f];
        };
        int mine = 0;                                      // (wave 0) rotations decided for the sweep in progress
        int mine_next = 0;                                 // ... for the next sweep (the look-ahead of a sweep's last step)
        // parameters of step 0 straight from G
        if (tid < 64) {
            bool rot = false;
            double c = 1.0, sn = 0.0;
            int p = 0, q = 0;
            if (tid < HP) {
                slot_pair(0, tid, p, q);
                rot = decide(G[p * P + p], G[q * P + q], G[p * P + q], c, sn);
                if (rot) ++mine;
            }
            if (tid < HP) publish(0, 0, tid, p, c, sn, rot);
            else (void)__ballot(false);
        }
        __syncthreads();
        int par = 0;                                       // parameter buffer of the step in progress
        // one step of the tournament: phase of step s (u = s mod FU compile-time for the register frame of J)
        auto step = [&](int s, auto uc) -> void {
            constexpr int u = decltype(uc)::value;
            const int flag = stepflag[par];                 // uniform
            const double* Gc = Gb[cur];
            double* Gn = Gb[cur ^ 1];
            if (tid < 64) {
                // ---- wave 0: rotations of step s+1 from G[cur] and the rotations of step s
                const int s1 = (s + 1 == M) ? 0 : s + 1;
                bool rot = false;
                double c = 1.0, sn = 0.0;
                int p = 0, q = 0;
                if (tid < HP) {
                    pair_of(s1, tid, p, q);
                    double gpp, gqq, gpq;
                    if (dbg & 4) { gpp = 1.0; gqq = 1.0; gpq = 0.0; }
                    else if (flag) {
                        const unsigned sp = so_tab[s * NB + p], sq = so_tab[s * NB + q];
                        const int ap = (int)(sp & 127u), hp_ = (int)(sp >> 7), aq = (int)(sq & 127u), hq_ = (int)(sq >> 7);
                        int pp, qp, pq, qq;
                        pair_of(s, ap, pp, qp);             // pair holding p in step s
                        pair_of(s, aq, pq, qq);             // pair holding q in step s
                        const double2 rp = *reinterpret_cast<const double2*>(&cs2[par][2 * ap]);
                        const double2 rq = *reinterpret_cast<const double2*>(&cs2[par][2 * aq]);
                        // block (ap, ap) -> g_pp;  block (aq, aq) -> g_qq;  block (ap, aq) -> g_pq
                        const double a00 = Gc[pp * P + pp], a01 = Gc[pp * P + qp], a10 = Gc[qp * P + pp], a11 = Gc[qp * P + qp];
                        const double b00 = Gc[pq * P + pq], b01 = Gc[pq * P + qq], b10 = Gc[qq * P + pq], b11 = Gc[qq * P + qq];
                        const double x00 = Gc[pp * P + pq], x01 = Gc[pp * P + qq], x10 = Gc[qp * P + pq], x11 = Gc[qp * P + qq];
                        {
                            const double t0 = hp_ ? rot_q(rp.x, rp.y, a00, a10) : rot_p(rp.x, rp.y, a00, a10);
                            const double t1 = hp_ ? rot_q(rp.x, rp.y, a01, a11) : rot_p(rp.x, rp.y, a01, a11);
                            gpp = hp_ ? rot_q(rp.x, rp.y, t0, t1) : rot_p(rp.x, rp.y, t0, t1);
                        }
                        {
                            const double t0 = hq_ ? rot_q(rq.x, rq.y, b00, b10) : rot_p(rq.x, rq.y, b00, b10);
                            const double t1 = hq_ ? rot_q(rq.x, rq.y, b01, b11) : rot_p(rq.x, rq.y, b01, b11);
                            gqq = hq_ ? rot_q(rq.x, rq.y, t0, t1) : rot_p(rq.x, rq.y, t0, t1);
                        }
                        if (ap <= aq) {     // block (ap, aq): rows of p's pair, columns of q's pair, output (hp_, hq_)
                            const double t0 = hp_ ? rot_q(rp.x, rp.y, x00, x10) : rot_p(rp.x, rp.y, x00, x10);
                            const double t1 = hp_ ? rot_q(rp.x, rp.y, x01, x11) : rot_p(rp.x, rp.y, x01, x11);
                            gpq = hq_ ? rot_q(rq.x, rq.y, t0, t1) : rot_p(rq.x, rq.y, t0, t1);
                        } else {            // the update computes block (aq, ap) and mirrors it: rows of q's pair, columns of p's pair
                            // (its entries G[pq][pp], G[pq][qp], G[qq][pp], G[qq][qp] equal x00, x10, x01, x11: G is exactly symmetric)
                            const double t0 = hq_ ? rot_q(rq.x, rq.y, x00, x01) : rot_p(rq.x, rq.y, x00, x01);
                            const double t1 = hq_ ? rot_q(rq.x, rq.y, x10, x11) : rot_p(rq.x, rq.y, x10, x11);
                            gpq = hp_ ? rot_q(rp.x, rp.y, t0, t1) : rot_p(rp.x, rp.y, t0, t1);
                        }
                    } else {
                        gpp = Gc[p * P + p]; gqq = Gc[q * P + q]; gpq = Gc[p * P + q];
                    }
                    rot = decide(gpp, gqq, gpq, c, sn);
                    if (dbg) rot = true;                   // diagnostics: keep every step "rotating"
                    if (rot) { if (s + 1 == M) ++mine_next; else ++mine; }
                }
                if (tid < HP) publish(par ^ 1, s1, tid, p, c, sn, rot);
                else (void)__ballot(false);
            } else if (jwave) {
                if (flag && !(dbg & 1)) apply_j(par, uc);
                if constexpr (u == FU - 1) advance_frame();
            } else if (flag && wave_g >= 0 && !(dbg & 2)) {
                // ---- waves 1, 2, 4, 5, 6: G[nxt] <- R^T G[cur] R over the blocks (a <= b), each written to both triangles
                int pa[UB], qa[UB], pb[UB], qb[UB];
                double2 ra[UB], rb[UB];
                double g00[UB], g01[UB], g10[UB], g11[UB];
                bool on[UB];
#pragma unroll
                for (int k = 0; k < UB; ++k) {
                    const int e = wave_g * 64 + lane_ + GT * k;
                    on[k] = e < NBLK;
                    const int a = on[k] ? blk_a[e] : 0, b = on[k] ? blk_b[e] : 0;
                    pair_of(s, a, pa[k], qa[k]);
                    pair_of(s, b, pb[k], qb[k]);
                    ra[k] = *reinterpret_cast<const double2*>(&cs2[par][2 * a]);
                    rb[k] = *reinterpret_cast<const double2*>(&cs2[par][2 * b]);
                    g00[k] = Gc[pa[k] * P + pb[k]]; g01[k] = Gc[pa[k] * P + qb[k]];
                    g10[k] = Gc[qa[k] * P + pb[k]]; g11[k] = Gc[qa[k] * P + qb[k]];
                }
#pragma unroll
                for (int k = 0; k < UB; ++k) {
                    if (!on[k]) continue;
                    const double t00 = rot_p(ra[k].x, ra[k].y, g00[k], g10[k]), t01 = rot_p(ra[k].x, ra[k].y, g01[k], g11[k]);
                    const double t10 = rot_q(ra[k].x, ra[k].y, g00[k], g10[k]), t11 = rot_q(ra[k].x, ra[k].y, g01[k], g11[k]);
                    const double o00 = rot_p(rb[k].x, rb[k].y, t00, t01), o01 = rot_q(rb[k].x, rb[k].y, t00, t01);
                    const double o10 = rot_p(rb[k].x, rb[k].y, t10, t11), o11 = rot_q(rb[k].x, rb[k].y, t10, t11);
                    if (pa[k] == pb[k]) {                  // diagonal block: (p, q) and (q, p) both take o01
                        Gn[pa[k] * P + pa[k]] = o00; Gn[pa[k] * P + qa[k]] = o01;
                        Gn[qa[k] * P + pa[k]] = o01; Gn[qa[k] * P + qa[k]] = o11;
                    } else {
                        Gn[pa[k] * P + pb[k]] = o00; Gn[pb[k] * P + pa[k]] = o00;
                        Gn[pa[k] * P + qb[k]] = o01; Gn[qb[k] * P + pa[k]] = o01;
                        Gn[qa[k] * P + pb[k]] = o10; Gn[pb[k] * P + qa[k]] = o10;
                        Gn[qa[k] * P + qb[k]] = o11; Gn[qb[k] * P + qa[k]] = o11;
                    }
                }
            }
            __syncthreads();
            cur ^= flag;
            par ^= 1;
        };
        for (int sweep = 0; sweep < max_sweeps; ++sweep) {
            if (tid == 0) cnt = 0;
            for (int s0 = 0; s0 < M; s0 += FU) {
                if constexpr (FU == 7) {
                    step(s0 + 0, std::integral_constant<int, 0>{});
                    step(s0 + 1, std::integral_constant<int, 1>{});
                    step(s0 + 2, std::integral_constant<int, 2>{});
                    step(s0 + 3, std::integral_constant<int, 3>{});
                    step(s0 + 4, std::integral_constant<int, 4>{});
                    step(s0 + 5, std::integral_constant<int, 5>{});
                    step(s0 + 6, std::integral_constant<int, 6>{});
                } else {
                    step(s0, std::integral_constant<int, 0>{});
                }
            }
            if (mine) atomicAdd(&cnt, mine);
            __syncthreads();
            if (tid == 0) total += cnt;
            const int done = (cnt == 0);
            __syncthreads();
            mine = mine_next;
            mine_next = 0;
            if (done) break;
        }
        if (jwave && tid - 192 < NB) {
#pragma unroll
            for (int k = 0; k < NB; ++k) J[(tid - 192) * P + k] = jr[k];
        }
        __syncthreads();
    }
    EIG_CLK(2);
    double* Gf = Gb[cur];                                  // the rotated Gram matrix
    double* Gs = Gb[cur ^ 1];                              // scratch for the Newton-Schulz step
    if (tid == 0 && nrot_out) {
        if constexpr (COH) __hip_atomic_store((__attribute__((address_space(1))) int*)(nrot_out + grp), (int)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else nrot_out[grp] = total;
    }
    if (mode != 1 && total > 0) {
        typedef double d4e __attribute__((ext_vector_type(4)));
        constexpr int NT = NB / 16, TPW = NT * NT / 4;
        const bool mm = tid < 256;                            // the two products run on waves 0-3
        const int lane = tid & 63, wave = (tid >> 6) & 3, li = lane & 15, lk = lane >> 4;
        const int ti = (NB == 64) ? wave : (wave >> 1), tj0 = (NB == 64) ? 0 : (wave & 1);
        d4e acc[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = d4e{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < NB / 4; ++ks) {                   // S = J^T J
            const int k = ks * 4 + lk;
            const double fa = J[k * P + ti * 16 + li];
#pragma unroll
            for (int t = 0; t < TPW; ++t)
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, J[k * P + (tj0 + t) * 16 + li], acc[t], 0, 0, 0);
        }
        if (mm) {
#pragma unroll
            for (int t = 0; t < TPW; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) Gs[(ti * 16 + lk + 4 * r) * P + (tj0 + t) * 16 + li] = acc[t][r];
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = d4e{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < NB / 4; ++ks) {                   // N = J S
            const int k = ks * 4 + lk;
            const double fa = J[(ti * 16 + li) * P + k];
#pragma unroll
            for (int t = 0; t < TPW; ++t)
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, Gs[k * P + (tj0 + t) * 16 + li], acc[t], 0, 0, 0);
        }
        double nv[TPW][4];
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                nv[t][r] = 1.5 * J[(ti * 16 + lk + 4 * r) * P + (tj0 + t) * 16 + li] - 0.5 * acc[t][r];
        __syncthreads();
        if (mm) {
#pragma unroll
            for (int t = 0; t < TPW; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) J[(ti * 16 + lk + 4 * r) * P + (tj0 + t) * 16 + li] = nv[t][r];
        }
        __syncthreads();
    }
    (void)Gf;
    EIG_CLK(3);
    double* o = out + (int64_t)grp * nvec * nvec;
    for (int e = tid; e < nvec * nvec; e += NT_) {
        const int i = e / nvec, j = e % nvec;
        double v = J[i * P + j];
        if (mode != 2) {
            const double di = dsc[i];
            v = (di == 0.0) ? 0.0 : v / di;
            if (dsc[j] == 0.0) v = 0.0;
        }
        stc(o + e, v);
    }
    EIG_CLK(4);
}

template <int NB>
__global__ __launch_bounds__(512) void eig_small3_kernel(const double* __restrict__ part, int nchunk, int nvec, int mode,
                                                         int max_sweeps, double dead_thresh,
                                                         double* __restrict__ out, int* __restrict__ dead,
                                                         int* __restrict__ nrot_out, double* __restrict__ maxoff_out,
                                                         double relevant2, int dbg, double fast_thr) {
    __shared__ double pool[4 * NB * (NB + 1)];
    eig_small3_body<NB>(pool, (int)blockIdx.x, part, nchunk, nvec, mode, max_sweeps, dead_thresh, out, dead, nrot_out, maxoff_out, relevant2, dbg, fast_thr);
}

int eig_small(hipStream_t st, const double* part, int nchunk, int nvec, int ngroups, int mode, int max_sweeps,
              double dead_thresh, double* out, int* dead, int* nrot, double* maxoff, double relevant2, int allow_fast) {
    TN_CHECK_ARG(nvec >= 1 && nvec <= NBMAX, "nvec out of range");
    if (ngroups <= 0) return 0;
    prof_begin(st, PROF_EIG);
    // TN_EIG_PIPELINED=0 selects the first-generation kernel (two barriers per Jacobi step) for A/B measurements
    static const int gen = [] { const char* e = getenv("TN_EIG_PIPELINED"); return e ? atoi(e) : 2; }();      // 0, 1 (256 threads), 2 (512 threads)
    const bool pipelined = gen == 1;
    static const int dbg = [] { const char* e = getenv("TN_EIG_DBG"); return e ? atoi(e) : 0; }();      // timing diagnostics of the third form
    // TN_EIG_FAST: largest relative off-diagonal up to which a pair takes the near-diagonal fast path (0 = never; see eig_small3_kernel)
    static const double fast_thr = [] { const char* e = getenv("TN_EIG_FAST"); return e ? atof(e) : 1e-2; }();
    if (gen >= 2) {
        if (nvec <= 32)
            hipLaunchKernelGGL((eig_small3_kernel<32>), dim3(ngroups), dim3(512), 0, st, part, nchunk, nvec, mode, max_sweeps,
                               dead_thresh, out, dead, nrot, maxoff, relevant2, dbg, 0.0);
        else
            hipLaunchKernelGGL((eig_small3_kernel<64>), dim3(ngroups), dim3(512), 0, st, part, nchunk, nvec, mode, max_sweeps,
                               dead_thresh, out, dead, nrot, maxoff, relevant2, dbg, allow_fast ? fast_thr : 0.0);
    } else if (pipelined) {
        if (nvec <= 32)
            hipLaunchKernelGGL((eig_small2_kernel<32>), dim3(ngroups), dim3(256), 0, st, part, nchunk, nvec, mode, max_sweeps,
                               dead_thresh, out, dead, nrot, maxoff, relevant2);
        else
            hipLaunchKernelGGL((eig_small2_kernel<64>), dim3(ngroups), dim3(256), 0, st, part, nchunk, nvec, mode, max_sweeps,
                               dead_thresh, out, dead, nrot, maxoff, relevant2);
    } else if (nvec <= 32)
        hipLaunchKernelGGL((eig_small_kernel<32>), dim3(ngroups), dim3(256), 0, st, part, nchunk, nvec, mode, max_sweeps,
                           dead_thresh, out, dead, nrot, maxoff, relevant2);
    else
        hipLaunchKernelGGL((eig_small_kernel<64>), dim3(ngroups), dim3(256), 0, st, part, nchunk, nvec, mode, max_sweeps,
                           dead_thresh, out, dead, nrot, maxoff, relevant2);
    TN_CHECK_LAUNCH("eig_small_kernel");
    prof_end(st, PROF_EIG, 0.0, 8.0 * ngroups * ((double)nchunk + 1.0) * nvec * nvec);
    return 0;
}

// ------------------------------------------------------------------------------------------ all Jacobi rounds of an SVD in ONE launch
// jacobi_core (svd.hip) spends three launches per round -- pair Gram matrices, eig_small, rotation of the vectors -- and one host
// synchronisation per sweep: ~60 launches and 5-6 round trips per truncated SVD of the headline workload, 11 k launches per sweep of a
// chain.  svdl_kernel runs the SAME three steps (every sum in the same order: results identical bit for bit) as phases of one
// persistent launch, separated by grid barriers; the convergence test of a sweep (largest relative off-diagonal met) is taken by
// every workgroup from the same numbers; the squared norms of the rotated vectors -- what the host sorts the singular values from --
// close the launch.  Every vector is cut into chunks of 64 elements and workgroup c keeps chunk c of ALL vectors (X and the accumulator
// P: nvp x 64 doubles, up to 96 KB) in LDS for the whole launch; ng further workgroups solve the eigenproblems.  Per round: the chunk
// workgroups form their 64-element share of every pair's Gram matrix straight from LDS (MFMA, the sequence of the GEMM kernel over
// K = 64) and publish it; barrier; eigenproblems (the body of eig_small3_kernel; its prologue adds the shares up in chunk order,
// gram_nchunk cuts the separate launches at the same 64); barrier; the chunk workgroups fetch J and rotate their rows in LDS.  What
// crosses workgroups (agent-scope accesses) is the partial Gram matrices and J -- the vectors travel once in and once out.
// A form that kept the vectors in memory and ran the GEMM tile body between the barriers was measured first: every K step of those
// small products is a memory round trip, ~40 us per round around the eigenproblems against ~23 us for the two GEMM launches.
// The workgroups spin on barriers, so they must be co-resident: the grid is kept within the budget of cholqr.hip (fused_forms_allowed),
// the spins are bounded, and a launch in which a barrier gave up says so in its status word (the caller redoes the rounds with the
// three-launch form and takes the stream off the single-launch forms).  The barrier state cleans itself: all workgroups leave sooner
// or later, the last one resets the counters.
// Needs nvp <= 256 (LDS: 133 KB of vectors; 192 while the J of the pair being rotated was staged in the pool) and ceil(pitch / 64) + ng
// workgroups within the budget; otherwise the rounds stay separate launches.
// -DTN_CLOCKS: thread 0 of workgroup 0 (a chunk workgroup) accumulates the 100 MHz wall clock per phase over all launches:
// [0] load  [1] Gram shares  [2] barrier  [3] wait for the eigenproblems  [4] rotation  [5] store + norms  [6] rounds  [7] launches
#ifdef TN_CLOCKS
__device__ long long svdl_clk[8];
}  // namespace tn
extern "C" int tn_debug_svdl_clocks(long long* host, int reset) {
    int rc = (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(tn::svdl_clk), sizeof(long long) * 8);
    if (reset) { long long z[8] = {}; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(tn::svdl_clk), z, sizeof(z)); }
    return rc;
}
namespace tn {
#define SVL_CLK(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) { const long long now_ = wall_clock64(); svdl_clk[k] += now_ - clk_last_; clk_last_ = now_; } } while (0)
#define SVL_CNT(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) svdl_clk[k] += 1; } while (0)
#define SVL_CLK_INIT long long clk_last_ = wall_clock64()
#else
#define SVL_CLK(k) do {} while (0)
#define SVL_CNT(k) do {} while (0)
#define SVL_CLK_INIT do {} while (0)
#endif
struct SvdjState { int counter; int exits; int gaveup; int pad; };
__device__ SvdjState svdj_state_pool[CHOLQR_SLOTS];
constexpr unsigned SVDJ_MAGIC = 0x53564a31u;

// naps: the pause between two looks at the counter, in units of ~0.4 us.  The wait for the eigenproblems lasts 15-100 us and up to 30
// workgroups per chain sit in it.
__device__ __forceinline__ bool svdj_barrier(int* counter, int target, int* s_flag, int tid, unsigned spin_limit, int naps = 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) {
        atomicAdd(counter, 1);
        int ok = 1;
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(15);
            if (++spins > spin_limit) { ok = 0; break; }
        }
        *s_flag = ok;
    }
    __syncthreads();
    return *s_flag != 0;
}

struct SvdlArgs {
    double* X;
    int64_t pitch, L;
    int nvp, ncw, ng, nr, nchunk;
    const int* pairs;
    double* part;
    double* Js;
    int* nrot;
    double* maxoff;
    double relevant2, fast_thr, last_tol;
    int inner_first, inner_later, dbg;
    double* norms;                     // nvp squared norms, then status: sweeps, converged, workgroups that gave up
    SvdjState* stt;
    unsigned spin_limit;
    unsigned magic;
    int eig_naps;
};
typedef const __attribute__((address_space(4))) SvdlArgs* SvdlArgsK;

__device__ __noinline__ void svdl_eig(SvdlArgsK a_in, double* pool, int grp, int r, int outer) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long v = (unsigned long long)a_in;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    SvdlArgsK a = (SvdlArgsK)(((unsigned long long)hi << 32) | lo);
    const int inner = (a->ng == 1) ? 12 : (outer == 0 ? a->inner_first : a->inner_later);
    eig_small3_body<64, true>(pool, grp, a->part, a->nchunk, 64, 2, inner, 0.0, a->Js, nullptr, a->nrot, a->maxoff + (int64_t)r * a->ng,
                              a->relevant2, a->dbg, outer < 8 ? a->fast_thr : 0.0);
#endif
}

__global__ __launch_bounds__(512) void svdl_kernel(SvdlArgs a) {
    constexpr int NB = 64, P = NB + 1, XP = 65, W = 32;
    typedef double d4l __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) double* gcd;
    typedef __attribute__((address_space(1))) double* gd;
    __shared__ double pool[4 * NB * P];
    __shared__ double redn[256];
    __shared__ int s_flag;
    __shared__ int s_pairs[64];
    __shared__ int s_rot[32];
    const int tid = threadIdx.x, blk = blockIdx.x, nwg = gridDim.x;
    SvdlArgsK ak = (SvdlArgsK)__builtin_amdgcn_kernarg_segment_ptr();
    const bool chunk_wg = blk < a.ncw;
    const int c0 = blk * 64, grp = blk - a.ncw;
    double* Xc = pool;                       // chunk workgroups: [nvp][XP], nvp <= 256
    int nbar = 0, sweeps = 0;
    bool alive = ak != nullptr && ak->magic == SVDJ_MAGIC && ak->nvp == a.nvp && ak->pairs == a.pairs && ak->norms == a.norms && ak->Js == a.Js;
    bool converged = false;
    auto bar = [&](int naps = 0) -> bool { ++nbar; return svdj_barrier(&a.stt->counter, nbar * nwg, &s_flag, tid, a.spin_limit, naps); };
    const int wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int ti = wave >> 1, tj0 = (wave & 1) * 2;          // this wave's two 16 x 16 tiles of a 64 x 64 result
    SVL_CLK_INIT;
    SVL_CNT(7);
    if (chunk_wg && alive) {
        for (int e = tid; e < a.nvp * 64; e += 512) {
            const int r = e >> 6, c = e & 63;
            Xc[r * XP + c] = (c0 + c < a.pitch) ? a.X[(int64_t)r * a.pitch + c0 + c] : 0.0;
        }
    }
    __syncthreads();
    SVL_CLK(0);
    for (int outer = 0; outer < 40 && !converged && alive; ++outer) {
        for (int r = 0; r < a.nr; ++r) {
            SVL_CNT(6);
            if (tid < 2 * a.ng && tid < 64) s_pairs[tid] = a.pairs[(int64_t)r * a.ng * 2 + tid];
            __syncthreads();
            if (chunk_wg && blk < a.nchunk) {
                // this chunk's share of every pair's Gram matrix (the MFMA sequence of gemm_kernel over K = 64, elements past L as zeros)
                for (int z = 0; z < a.ng; ++z) {
                    const int b0 = s_pairs[2 * z], b1 = s_pairs[2 * z + 1];
                    auto prow = [&](int v) -> int { return v < W ? b0 * W + v : b1 * W + (v - W); };
                    const int ra = prow(ti * 16 + li), rb0 = prow(tj0 * 16 + li), rb1 = prow((tj0 + 1) * 16 + li);
                    // (the eigenproblem reads the tiles on and above the diagonal only: the others are neither formed nor published)
                    const bool on0 = ti <= tj0, on1 = ti <= tj0 + 1;          // (wave-uniform)
                    d4l acc0 = d4l{0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
                    if (on1) {
#pragma unroll
                        for (int k4 = 0; k4 < 16; ++k4) {
                            const int k = k4 * 4 + lk;
                            const bool in = c0 + k < a.L;
                            const double fa = in ? Xc[ra * XP + k] : 0.0;
                            const double f0 = in ? Xc[rb0 * XP + k] : 0.0, f1 = in ? Xc[rb1 * XP + k] : 0.0;
                            if (on0) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, f0, acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, f1, acc1, 0, 0, 0);
                        }
                    }
                    gd o = (gd)(a.part + ((int64_t)z * a.nchunk + blk) * (NB * NB));
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = ti * 16 + lk + 4 * q;
                        if (on0) __hip_atomic_store(o + row * NB + tj0 * 16 + li, acc0[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (on1) __hip_atomic_store(o + row * NB + (tj0 + 1) * 16 + li, acc1[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            SVL_CLK(1);
            if (!(alive = bar())) break;
            SVL_CLK(2);
            if (!chunk_wg) {
                svdl_eig(ak, pool, grp, r, outer);
                __syncthreads();
            }
            if (!(alive = bar(chunk_wg ? a.eig_naps : 0))) break;
            SVL_CLK(3);
            if (chunk_wg) {
                if (tid < a.ng && tid < 32) s_rot[tid] = __hip_atomic_load((const __attribute__((address_space(1))) int*)(a.nrot + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
                for (int z = 0; z < a.ng; ++z) {
                    if (s_rot[z] == 0) continue;                    // (uniform) no rotation in this pair
                    const int b0 = s_pairs[2 * z], b1 = s_pairs[2 * z + 1];
                    auto prow = [&](int v) -> int { return v < W ? b0 * W + v : b1 * W + (v - W); };
                    // this lane's operands of J (64 x 64) come straight from memory (L2): 16 independent loads in flight, no staging in LDS --
                    // the pool belongs to the vectors alone, up to 256 of them
                    gcd jsrc = (gcd)(a.Js + (int64_t)z * (NB * NB));
                    double ja[16];
#pragma unroll
                    for (int k4 = 0; k4 < 16; ++k4) ja[k4] = __hip_atomic_load(jsrc + (k4 * 4 + lk) * NB + ti * 16 + li, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // rows of the pair <- J^T rows (gemm_kernel's sequence over K = 64 vectors)
                    d4l acc0 = d4l{0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
#pragma unroll
                    for (int k4 = 0; k4 < 16; ++k4) {
                        const int k = k4 * 4 + lk;
                        const int rk = prow(k);
                        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ja[k4], Xc[rk * XP + tj0 * 16 + li], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ja[k4], Xc[rk * XP + (tj0 + 1) * 16 + li], acc1, 0, 0, 0);
                    }
                    __syncthreads();                                // every wave has read the old rows
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int ri = prow(ti * 16 + lk + 4 * q);
                        Xc[ri * XP + tj0 * 16 + li] = acc0[q];
                        Xc[ri * XP + (tj0 + 1) * 16 + li] = acc1[q];
                    }
                    __syncthreads();
                }
            }
        }
        if (!alive) break;
        ++sweeps;
        SVL_CLK(4);
        double m = 0.0;
        for (int e = tid; e < a.nr * a.ng; e += 512) m = fmax(m, __hip_atomic_load((gcd)(a.maxoff + e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
        if ((tid & 63) == 0) redn[tid >> 6] = m;
        __syncthreads();
        double worst = 0.0;
#pragma unroll
        for (int w = 0; w < 8; ++w) worst = fmax(worst, redn[w]);
        __syncthreads();
        converged = worst < 4.0e-15 || worst <= a.last_tol;
    }
    if (alive) {
        if (chunk_wg) {
            for (int e = tid; e < a.nvp * 64; e += 512) {
                const int r = e >> 6, c = e & 63;
                if (c0 + c < a.pitch) __hip_atomic_store((gd)(a.X + (int64_t)r * a.pitch + c0 + c), Xc[r * XP + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        alive = bar();
    }
    if (alive) {
        // squared norms of the vectors, summed as vec_norm2_kernel sums them
        for (int row = blk; row < a.nvp; row += nwg) {
            gcd x = (gcd)(a.X + (int64_t)row * a.pitch);
            double sacc = 0.0;
            if (tid < 256)
                for (int64_t c = tid; c < a.L; c += 256) { const double t = __hip_atomic_load(x + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); sacc += t * t; }
            if (tid < 256) redn[tid] = sacc;
            __syncthreads();
            for (int k = 128; k > 0; k >>= 1) {
                if (tid < k) redn[tid] += redn[tid + k];
                __syncthreads();
            }
            if (tid == 0) a.norms[row] = redn[0];
            __syncthreads();
        }
        SVL_CLK(5);
        if (blk == 0 && tid == 0) { a.norms[a.nvp] = (double)sweeps; a.norms[a.nvp + 1] = converged ? 1.0 : 0.0; }
    }
    if (tid == 0) {
        if (!alive) atomicAdd(&a.stt->gaveup, 1);
        __threadfence();
        const int prev = atomicAdd(&a.stt->exits, 1);
        if (prev == nwg - 1) {
            const int gu = atomicAdd(&a.stt->gaveup, 0);
            a.norms[a.nvp + 2] = (double)gu;
            a.stt->exits = 0; a.stt->counter = 0; a.stt->gaveup = 0;
            __threadfence();
        }
    }
}

static SvdjState* svdj_state_of(int slot) {
    static std::mutex mu;
    static char* base[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!base[dev]) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(svdj_state_pool)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        base[dev] = (char*)p;
    }
    return (SvdjState*)base[dev] + slot;
}

// 0: the launch is in the stream (read norms[nvp .. nvp + 2] back: sweeps, converged, workgroups that gave up); 1: not taken (the
// caller runs the rounds as separate launches); else an error
int svd_rounds_fused(hipStream_t st, const SvdRoundsJob& j) {
    {
        const char* e = getenv("TN_SVD_FUSED");                       // read per call: the tests switch it
        if (e && e[0] == '0') return 1;
    }
    static const int gen = [] { const char* e = getenv("TN_EIG_PIPELINED"); return e ? atoi(e) : 2; }();
    if (gen < 2) return 1;
    static const int dbg = [] { const char* e = getenv("TN_EIG_DBG"); return e ? atoi(e) : 0; }();
    static const double fast_thr = [] { const char* e = getenv("TN_EIG_FAST"); return e ? atof(e) : 1e-2; }();
    static const int eig_naps = [] { const char* e = getenv("TN_SVDJ_NAPS"); return e ? atoi(e) : 2; }();
    const int slot = cholqr_stream_slot(st);
    if (slot >= CHOLQR_SLOTS) return 1;
    if (2 * j.w != 64 || j.ng < 1 || j.ng > 32 || j.nr < 1 || j.nvp > 256) return 1;
    // the chunks of the kernel must be the splits the GEMM of the separate launches would use (gram_nchunk cuts at 64 up to L = 4096)
    int64_t kchunk = 0;
    const int nchunk = gemm_forced_split(j.L, j.nchunk, &kchunk);
    const int ncw = (int)cdiv(j.pitch, 64);
    if (!(kchunk == 64 || nchunk == 1) || nchunk < 1 || nchunk > ncw) return 1;
    if ((int64_t)j.ng * nchunk * 64 * 64 * 8 > j.part_bytes) return 1;
    if (!fused_forms_allowed(st, ncw + j.ng)) return 1;
    SvdlArgs l;
    l.X = j.X; l.pitch = j.pitch; l.L = j.L; l.nvp = j.nvp; l.ncw = ncw; l.ng = j.ng; l.nr = j.nr; l.nchunk = nchunk;
    l.pairs = j.pairs; l.part = j.part; l.Js = j.Js; l.nrot = j.nrot; l.maxoff = j.maxoff;
    l.relevant2 = j.relevant2; l.fast_thr = fast_thr; l.last_tol = j.last_tol;
    l.inner_first = j.inner_first; l.inner_later = j.inner_later; l.dbg = dbg;
    l.norms = j.norms;
    l.stt = svdj_state_of(slot);
    if (!l.stt) return 1;
    l.spin_limit = 1u << 22;
    if (const char* e = getenv("TN_PANEL_SPIN_LIMIT")) l.spin_limit = (unsigned)strtoul(e, nullptr, 10);      // tests: force the barriers to give up
    l.magic = SVDJ_MAGIC;
    l.eig_naps = eig_naps;
    prof_begin(st, PROF_EIG);
    hipLaunchKernelGGL(svdl_kernel, dim3(ncw + j.ng), dim3(512), 0, st, l);
    TN_CHECK_LAUNCH("svdl_kernel");
    prof_end(st, PROF_EIG, 0.0, 16.0 * (double)j.nvp * (double)j.pitch);      // the vectors in and out; what the rounds exchange is not compulsory
    return 0;
}

// ------------------------------------------------------------------------------------------ rows_times_small
template <int NB>
__global__ __launch_bounds__(256) void rows_times_small_kernel(double* __restrict__ X, int64_t rs, int64_t cs,
                                                               int64_t nrows, int b, const double* __restrict__ S) {
    __shared__ double Ss[NB * NB];
    const int tid = threadIdx.x;
    for (int e = tid; e < NB * NB; e += 256) {
        const int i = e / NB, j = e % NB;
        Ss[e] = (i < b && j < b) ? S[i * b + j] : 0.0;
    }
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * 256 + tid;
    if (r >= nrows) return;
    double* row = X + r * rs;
    double x[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) x[i] = (i < b) ? row[i * cs] : 0.0;
    for (int j = 0; j < b; ++j) {
        double y = 0.0;
#pragma unroll
        for (int i = 0; i < NB; ++i) y += x[i] * Ss[i * NB + j];
        row[j * cs] = y;
    }
}

int rows_times_small(hipStream_t st, double* X, int64_t rs, int64_t cs, int64_t nrows, int b, const double* S) {
    TN_CHECK_ARG(b >= 1 && b <= NBMAX, "b out of range");
    if (nrows <= 0) return 0;
    dim3 grid((unsigned)cdiv(nrows, 256));
    prof_begin(st, PROF_ROWS_SMALL);
    if (b <= 32) hipLaunchKernelGGL((rows_times_small_kernel<32>), grid, dim3(256), 0, st, X, rs, cs, nrows, b, S);
    else hipLaunchKernelGGL((rows_times_small_kernel<64>), grid, dim3(256), 0, st, X, rs, cs, nrows, b, S);
    TN_CHECK_LAUNCH("rows_times_small_kernel");
    prof_end(st, PROF_ROWS_SMALL, 2.0 * nrows * b * b, 16.0 * nrows * b);
    return 0;
}

// ------------------------------------------------------------------------------------------ small_t_times_vecs
template <int NV>
__global__ __launch_bounds__(256) void small_t_times_vecs_kernel(const double* __restrict__ S, double* __restrict__ X,
                                                                 int64_t vs, int64_t es, int64_t L, int nvec, int w,
                                                                 const int* __restrict__ pairs,
                                                                 const int* __restrict__ nrot) {
    __shared__ double Ss[NV * NV];
    const int tid = threadIdx.x, grp = blockIdx.y;
    if (nrot && nrot[grp] == 0) return;
    const double* s = S + (int64_t)grp * nvec * nvec;
    for (int e = tid; e < NV * NV; e += 256) {
        const int u = e / NV, v = e % NV;
        Ss[e] = (u < nvec && v < nvec) ? s[u * nvec + v] : 0.0;
    }
    __syncthreads();
    const int blk0 = pairs ? pairs[2 * grp] : 0, blk1 = pairs ? pairs[2 * grp + 1] : 1;
    const int64_t c = (int64_t)blockIdx.x * 256 + tid;
    if (c >= L) return;
    double x[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) x[u] = (u < nvec) ? X[vec_offset(u, w, blk0, blk1, vs) + c * es] : 0.0;
    for (int v = 0; v < nvec; ++v) {
        double y = 0.0;
#pragma unroll
        for (int u = 0; u < NV; ++u) y += Ss[u * NV + v] * x[u];
        X[vec_offset(v, w, blk0, blk1, vs) + c * es] = y;
    }
}

int small_t_times_vecs(hipStream_t st, const double* S, double* X, int64_t vs, int64_t es, int64_t L, int nvec, int w,
                       const int* pairs, int ngroups, const int* nrot) {
    TN_CHECK_ARG(nvec >= 1 && nvec <= NBMAX, "nvec out of range");
    if (ngroups <= 0 || L <= 0) return 0;
    dim3 grid((unsigned)cdiv(L, 256), ngroups);
    prof_begin(st, PROF_VECS_SMALL);
    if (nvec <= 32)
        hipLaunchKernelGGL((small_t_times_vecs_kernel<32>), grid, dim3(256), 0, st, S, X, vs, es, L, nvec, w, pairs, nrot);
    else
        hipLaunchKernelGGL((small_t_times_vecs_kernel<64>), grid, dim3(256), 0, st, S, X, vs, es, L, nvec, w, pairs, nrot);
    TN_CHECK_LAUNCH("small_t_times_vecs_kernel");
    prof_end(st, PROF_VECS_SMALL, 2.0 * nvec * nvec * (double)L * ngroups, 16.0 * nvec * (double)L * ngroups);
    return 0;
}

}  // namespace tn
