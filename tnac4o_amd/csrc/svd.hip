// K4 / K5 — one-sided block-Jacobi SVD with the reference's truncation rule (mps.py:802-811), sign gauge
// (mps.py:35-39) and a values-only variant (mps.py:62-73).
//
// The shorter side of C supplies the vectors (nv <= L).  Vectors whose norm is below 2^-56 of the largest are
// deflated up front (they cannot move any singular value by more than sqrt(nv) 2^-56 S0, far below the eps S0
// truncation threshold); the centre matrices of this path are numerically low-rank, so this typically shrinks a
// 1024-vector problem to 100-300.  The live vectors X (nvp x L, blocks of w = 32) are orthogonalised by rounds of a
// round-robin tournament over block pairs; each round is three launches over all pairs at once:
//   gram_partial (64 x 64 Gram of a pair, split over L)  ->  eig_small (parallel-order Jacobi, orthogonal J)
//   ->  small_t_times_vecs  (X_pair <- J^T X_pair, and the same on the accumulator P).
// Rotations are exactly orthogonal to rounding; the Gram matrix only steers them, so small singular values keep the
// one-sided Jacobi accuracy.  Host syncs: one for deflation, one per sweep (convergence), one for the kept rank.
#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"

namespace tn {

constexpr int SVD_W = 32;

__global__ __launch_bounds__(256) void vec_norm2_kernel(const double* __restrict__ X, int64_t vs, int64_t es, int64_t L,
                                                        double* __restrict__ out) {
    __shared__ double red[256];
    const int v = blockIdx.x, tid = threadIdx.x;
    const double* x = X + (int64_t)v * vs;
    double s = 0.0;
    for (int64_t c = tid; c < L; c += 256) { const double t = x[c * es]; s += t * t; }
    red[tid] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (tid < k) red[tid] += red[tid + k];
        __syncthreads();
    }
    if (tid == 0) out[v] = red[0];
}

// X[i] <- M[live[i]] (zero rows beyond nvl); P[i] <- one-hot(live[i]) when P != nullptr
__global__ __launch_bounds__(256) void svd_init_kernel(const double* __restrict__ M, int64_t vs, int64_t es, int64_t L,
                                                       int64_t nv, const int* __restrict__ live, int nvl,
                                                       double* __restrict__ X, double* __restrict__ P, int64_t pitch) {
    const int i = blockIdx.x, tid = threadIdx.x;
    const bool on = i < nvl;
    const int src = on ? live[i] : 0;
    for (int64_t c = tid; c < L; c += 256) X[(int64_t)i * pitch + c] = on ? M[(int64_t)src * vs + c * es] : 0.0;
    if (P)
        for (int64_t c = tid; c < nv; c += 256) P[(int64_t)i * pitch + c] = (on && c == src) ? 1.0 : 0.0;
}

// One block per kept vector j (source row order[j]):  left[:, j] = sgn * P[row, :],  right[j, :] = sgn * X[row, :] / S_j
// with the reference's sign gauge: flip when in both vectors the most negative entry outweighs the most positive.
// Up to 64 kept vectors (every truncation to chi <= 64): their values and rows travel by value in the kernel arguments (`byval`),
// no host-to-device copy; more: `order` / `Ssorted` in device memory.
struct GatherList { double S[64]; int order[64]; };
__global__ __launch_bounds__(256) void svd_gather_kernel(const double* __restrict__ X, int64_t L, const double* __restrict__ P,
                                                         int64_t nv, int64_t pitch, const int* __restrict__ order,
                                                         const double* __restrict__ Ssorted, double* __restrict__ Sout,
                                                         double* __restrict__ left, int64_t lrs, int64_t lcs,
                                                         double* __restrict__ right, int64_t rrs, int64_t rcs, GatherList gl, int byval) {
    __shared__ double rmin[256], rmax[256];
    __shared__ double sgn;
    const int j = blockIdx.x, tid = threadIdx.x;
    const int row = byval ? gl.order[j] : order[j];
    const double* x = X + (int64_t)row * pitch;
    const double* p = P + (int64_t)row * pitch;
    double xmin = 0.0, xmax = 0.0, pmin = 0.0, pmax = 0.0;
    bool first = true;
    for (int64_t c = tid; c < L; c += 256) { const double t = x[c]; xmin = first ? t : fmin(xmin, t); xmax = first ? t : fmax(xmax, t); first = false; }
    if (first) { xmin = 1e308; xmax = -1e308; }
    rmin[tid] = xmin; rmax[tid] = xmax;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (tid < k) { rmin[tid] = fmin(rmin[tid], rmin[tid + k]); rmax[tid] = fmax(rmax[tid], rmax[tid + k]); }
        __syncthreads();
    }
    xmin = rmin[0]; xmax = rmax[0];
    __syncthreads();
    first = true;
    for (int64_t c = tid; c < nv; c += 256) { const double t = p[c]; pmin = first ? t : fmin(pmin, t); pmax = first ? t : fmax(pmax, t); first = false; }
    if (first) { pmin = 1e308; pmax = -1e308; }
    rmin[tid] = pmin; rmax[tid] = pmax;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (tid < k) { rmin[tid] = fmin(rmin[tid], rmin[tid + k]); rmax[tid] = fmax(rmax[tid], rmax[tid + k]); }
        __syncthreads();
    }
    if (tid == 0) {
        pmin = rmin[0]; pmax = rmax[0];
        // the sign of x/S equals the sign of x (S > 0), so the rule can be evaluated on x directly
        sgn = (fabs(pmin) > pmax && fabs(xmin) > xmax) ? -1.0 : 1.0;
    }
    __syncthreads();
    const double s = sgn, sv = byval ? gl.S[j] : Ssorted[j];
    if (tid == 0) Sout[j] = sv;
    const double inv = sv > 0.0 ? s / sv : 0.0;
    for (int64_t c = tid; c < L; c += 256) right[(int64_t)j * rrs + c * rcs] = x[c] * inv;
    for (int64_t c = tid; c < nv; c += 256) left[c * lrs + (int64_t)j * lcs] = p[c] * s;
}

// Values-only SVD of a small centre matrix (both dimensions <= 64: the Schmidt-value checks of the final variational
// sweeps, mps.py:550-560) in ONE launch: Hestenes one-sided Jacobi on the rows held in LDS.  Round-robin pairing, 32
// pairs per step, 8 threads per pair (8 consecutive elements of both rows each; the three inner products meet through
// three xor-shuffles, every lane derives the same rotation).  Rows below 2^-56 of the largest are zeroed first, sweeps
// repeat until one passes without a rotation above 4e-15 (the block path's criterion).  out: 64 values sorted
// descending, out[64] = sweeps, out[65] = 1 if converged.  (A 128-row variant was tried: one workgroup then spends 2 ms on a
// 128 x 128 matrix against 0.9 ms for the block path, so 64 is the limit of the fused form.)
__device__ __forceinline__ void rr_pair64(int s, int a, int& p, int& q) {
    if (a == 0) { p = 63; q = s; }
    else { p = (s + a) % 63; q = (s - a + 63) % 63; }
    if (p > q) { const int t = p; p = q; q = t; }
}

__device__ __forceinline__ void svd_vals_small_body(const double* __restrict__ M, int64_t vs, int64_t es, int nv, int L,
                                                    double* __restrict__ out, double* X, double* nrm, int* flags) {
    constexpr int NV = 64, P = 66;            // even pitch: 16-byte aligned 8-element segments
    const int tid = threadIdx.x, slot = tid >> 3, sub = tid & 7;
    double relevant2 = 0.0;
    {
        double xv[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {        // one memory round trip
            const int e = tid + 256 * t, r = e >> 6, c = e & 63;
            xv[t] = (r < nv && c < L) ? M[(int64_t)r * vs + (int64_t)c * es] : 0.0;
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int e = tid + 256 * t;
            X[(e >> 6) * P + (e & 63)] = xv[t];
        }
    }
    __syncthreads();
    auto seg_dot3 = [&](const double* xp, const double* xq, double& a, double& b, double& g) {
        a = 0.0; b = 0.0; g = 0.0;
#pragma unroll
        for (int e = 0; e < 8; ++e) { a += xp[e] * xp[e]; b += xq[e] * xq[e]; g += xp[e] * xq[e]; }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); g += __shfl_xor(g, o, 64); }
    };
    {   // squared norms of rows 2 slot, 2 slot + 1; deflation
        double xp[8], xq[8], a, b, g;
#pragma unroll
        for (int e = 0; e < 8; ++e) { xp[e] = X[(2 * slot) * P + sub * 8 + e]; xq[e] = X[(2 * slot + 1) * P + sub * 8 + e]; }
        seg_dot3(xp, xq, a, b, g);
        if (sub == 0) { nrm[2 * slot] = a; nrm[2 * slot + 1] = b; }
        __syncthreads();
        double nmax = 0.0;
        for (int r = 0; r < NV; ++r) nmax = fmax(nmax, nrm[r]);
        const double thr = nmax * 1.9259299443872359e-34;       // (2^-56)^2
        relevant2 = nmax * 3.0814879110195774e-33;              // (2^-54)^2: see jacobi_core
        if (!(a > thr)) {
#pragma unroll
            for (int e = 0; e < 8; ++e) X[(2 * slot) * P + sub * 8 + e] = 0.0;
        }
        if (!(b > thr)) {
#pragma unroll
            for (int e = 0; e < 8; ++e) X[(2 * slot + 1) * P + sub * 8 + e] = 0.0;
        }
    }
    const double tol2 = 7.888609052210118e-31;                   // (2^-50)^2
    const double conv2 = 1.6e-29;                                // (4e-15)^2
    int sweeps = 0, converged = 0;
    for (int sweep = 0; sweep < 40; ++sweep) {
        if (tid < 2) flags[tid] = 0;
        __syncthreads();
        int any = 0, big = 0;
        for (int s = 0; s < NV - 1; ++s) {
            int p, q;
            rr_pair64(s, slot, p, q);
            double xp[8], xq[8], a, b, g;
#pragma unroll
            for (int e = 0; e < 8; ++e) { xp[e] = X[p * P + sub * 8 + e]; xq[e] = X[q * P + sub * 8 + e]; }
            seg_dot3(xp, xq, a, b, g);
            const double g2 = g * g, ab = a * b;
            if (g2 > tol2 * ab) {
                const double d = b - a;
                const double rh = fast_rsqrt(d * d + 4.0 * g2);
                const double c2 = 0.5 + 0.5 * fabs(d) * rh;
                const double rcv = fast_rsqrt(c2);
                const double sabs = fabs(g) * rh * rcv;
                if (sabs <= 1.0 && c2 <= 1.0000000000000002) {
                    const double c = c2 * rcv, sn = ((d >= 0.0) == (g >= 0.0)) ? sabs : -sabs;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        X[p * P + sub * 8 + e] = c * xp[e] - sn * xq[e];
                        X[q * P + sub * 8 + e] = sn * xp[e] + c * xq[e];
                    }
                    any = 1;
                    if (g2 > conv2 * ab && fmin(a, b) > relevant2) big = 1;
                }
            }
            __syncthreads();
        }
        if (any && sub == 0) atomicOr(&flags[0], 1);
        if (big && sub == 0) atomicOr(&flags[1], 1);
        __syncthreads();
        ++sweeps;
        const int f0 = flags[0], f1 = flags[1];
        __syncthreads();
        if (!f1) { converged = 1; if (!f0) break; break; }
    }
    {   // singular values = row norms; sorted descending by rank counting
        double xp[8], xq[8], a, b, g;
#pragma unroll
        for (int e = 0; e < 8; ++e) { xp[e] = X[(2 * slot) * P + sub * 8 + e]; xq[e] = X[(2 * slot + 1) * P + sub * 8 + e]; }
        seg_dot3(xp, xq, a, b, g);
        if (sub == 0) { nrm[2 * slot] = a; nrm[2 * slot + 1] = b; }
        __syncthreads();
        if (tid < NV) {
            const double mine = nrm[tid];
            int rank = 0;
            for (int r = 0; r < NV; ++r) rank += (nrm[r] > mine || (nrm[r] == mine && r < tid)) ? 1 : 0;
            out[rank] = sqrt(mine);
        }
        if (tid == 0) { out[64] = (double)sweeps; out[65] = (double)converged; }
    }
}

__global__ __launch_bounds__(256) void svd_vals_small_kernel(const double* __restrict__ M, int64_t vs, int64_t es, int nv, int L,
                                                             double* __restrict__ out) {
    __shared__ double X[64 * 66];
    __shared__ double nrm[64];
    __shared__ int flags[2];                  // [0] rotations this sweep, [1] rotations above the convergence threshold
    svd_vals_small_body(M, vs, es, nv, L, out, X, nrm, flags);
}

// One workgroup per item: desc[5 i .. 5 i + 4] = {device address of the matrix, vector stride, element stride, vectors, length}
// (the orientation is resolved by the host); out + 66 i receives the item's 64 values, sweeps and convergence flag.
__global__ __launch_bounds__(256) void svd_vals_small_batched_kernel(const int64_t* __restrict__ desc, double* __restrict__ out) {
    __shared__ double X[64 * 66];
    __shared__ double nrm[64];
    __shared__ int flags[2];
    const int64_t* d = desc + 5 * (int64_t)blockIdx.x;
    svd_vals_small_body(reinterpret_cast<const double*>(d[0]), d[1], d[2], (int)d[3], (int)d[4], out + 66 * (int64_t)blockIdx.x, X, nrm, flags);
}

// ---- truncated SVD of a small centre matrix (both dimensions <= 64) in ONE launch ---------------------------------------------
// A sweep truncates ~150 centre matrices of at most 64 x 64 (the bonds next to the edges of the boundary MPS: 1 x 1, 16 x 16,
// 60 x 60 ...); through the block path each costs a dozen launches and four read-backs, 80-330 us of pure latency.  Here: the
// Hestenes sweeps of svd_vals_small_body with the accumulated rotations kept beside the vectors (Pm starts as the identity and is
// rotated along: U = Pm^T), then -- all in the same workgroup -- the values sorted by rank counting, the truncation rule of
// mps.py:805-806 (keep S > S0 max(eps, tol), at most Dmax; discarded weight added up from the smallest value as the host does),
// the reference's sign gauge (mps.py:35-39) and the three outputs.  res (DEVICE, 4 doubles): keep, discarded, sweeps, converged.
__global__ __launch_bounds__(256) void svd_trunc_small_kernel(const double* __restrict__ M, int64_t vs, int64_t es, int nv, int L, int Dmax, double t,
                                                              double* __restrict__ Sout, double* __restrict__ left, int64_t lrs, int64_t lcs,
                                                              double* __restrict__ right, int64_t rrs, int64_t rcs, double* __restrict__ res) {
    constexpr int NV = 64, P = 66;
    __shared__ double X[NV * P];
    __shared__ double Pm[NV * P];
    __shared__ double nrm[NV];
    __shared__ double ssort[NV];
    __shared__ int order[NV];
    __shared__ int flags[2];
    __shared__ int s_keep;
    const int tid = threadIdx.x, slot = tid >> 3, sub = tid & 7;
    double relevant2 = 0.0, dscale = 1.0;
    {
        double xv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = tid + 256 * u, r = e >> 6, c = e & 63;
            xv[u] = (r < nv && c < L) ? M[(int64_t)r * vs + (int64_t)c * es] : 0.0;
        }
        // entries scaled into [0.5, 1) by a power of two (squares of squares must neither overflow nor underflow); NaN / Inf pass
        // through to the norm check below
        double mx = 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) { const double a = fabs(xv[u]); mx = (a == a) ? fmax(mx, a) : mx; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
        if ((tid & 63) == 0) nrm[tid >> 6] = mx;
        __syncthreads();
        mx = fmax(fmax(nrm[0], nrm[1]), fmax(nrm[2], nrm[3]));
        int ex = 0;
        if (mx > 0.0 && mx < 1.7e308) frexp(mx, &ex);
        dscale = ldexp(1.0, ex);
        const double scl = ldexp(1.0, -ex);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = tid + 256 * u, r = e >> 6, c = e & 63;
            X[r * P + c] = xv[u] * scl;
            Pm[r * P + c] = (r == c) ? 1.0 : 0.0;
        }
    }
    __syncthreads();
    auto seg_dot3 = [&](const double* xp, const double* xq, double& a, double& b, double& g) {
        a = 0.0; b = 0.0; g = 0.0;
#pragma unroll
        for (int e = 0; e < 8; ++e) { a += xp[e] * xp[e]; b += xq[e] * xq[e]; g += xp[e] * xq[e]; }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); g += __shfl_xor(g, o, 64); }
    };
    {   // squared norms of rows 2 slot, 2 slot + 1; rows below 2^-56 of the largest are dropped (as jacobi_core does)
        double xp[8], xq[8], a, b, g;
#pragma unroll
        for (int e = 0; e < 8; ++e) { xp[e] = X[(2 * slot) * P + sub * 8 + e]; xq[e] = X[(2 * slot + 1) * P + sub * 8 + e]; }
        seg_dot3(xp, xq, a, b, g);
        if (sub == 0) { nrm[2 * slot] = a; nrm[2 * slot + 1] = b; }
        __syncthreads();
        double nmax = 0.0;
        bool bad = false;
        for (int r = 0; r < NV; ++r) { nmax = fmax(nmax, nrm[r]); bad = bad || !(nrm[r] == nrm[r]) || nrm[r] > 1.7e308; }
        if (bad) {                                               // non-finite input (uniform): reported through a NaN discarded weight
            if (tid == 0) { res[0] = 0.0; res[1] = __longlong_as_double(0x7ff8000000000000LL); res[2] = 0.0; res[3] = 0.0; }
            return;
        }
        const double thr = nmax * 1.9259299443872359e-34;       // (2^-56)^2
        relevant2 = nmax * 3.0814879110195774e-33;              // (2^-54)^2
        if (!(a > thr)) {
#pragma unroll
            for (int e = 0; e < 8; ++e) X[(2 * slot) * P + sub * 8 + e] = 0.0;
        }
        if (!(b > thr)) {
#pragma unroll
            for (int e = 0; e < 8; ++e) X[(2 * slot + 1) * P + sub * 8 + e] = 0.0;
        }
    }
    const double tol2 = 7.888609052210118e-31;                   // (2^-50)^2
    const double conv2 = 1.6e-29;                                // (4e-15)^2
    int sweeps = 0, converged = 0;
    for (int sweep = 0; sweep < 40; ++sweep) {
        if (tid < 2) flags[tid] = 0;
        __syncthreads();
        int any = 0, big = 0;
        for (int s = 0; s < NV - 1; ++s) {
            int p, q;
            rr_pair64(s, slot, p, q);
            double xp[8], xq[8], a, b, g;
#pragma unroll
            for (int e = 0; e < 8; ++e) { xp[e] = X[p * P + sub * 8 + e]; xq[e] = X[q * P + sub * 8 + e]; }
            seg_dot3(xp, xq, a, b, g);
            const double g2 = g * g, ab = a * b;
            if (g2 > tol2 * ab) {
                const double d = b - a;
                const double rh = fast_rsqrt(d * d + 4.0 * g2);
                const double c2 = 0.5 + 0.5 * fabs(d) * rh;
                const double rcv = fast_rsqrt(c2);
                const double sabs = fabs(g) * rh * rcv;
                if (sabs <= 1.0 && c2 <= 1.0000000000000002) {
                    const double c = c2 * rcv, sn = ((d >= 0.0) == (g >= 0.0)) ? sabs : -sabs;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        X[p * P + sub * 8 + e] = c * xp[e] - sn * xq[e];
                        X[q * P + sub * 8 + e] = sn * xp[e] + c * xq[e];
                        const double up = Pm[p * P + sub * 8 + e], uq = Pm[q * P + sub * 8 + e];
                        Pm[p * P + sub * 8 + e] = c * up - sn * uq;
                        Pm[q * P + sub * 8 + e] = sn * up + c * uq;
                    }
                    any = 1;
                    if (g2 > conv2 * ab && fmin(a, b) > relevant2) big = 1;
                }
            }
            __syncthreads();
        }
        if (any && sub == 0) atomicOr(&flags[0], 1);
        if (big && sub == 0) atomicOr(&flags[1], 1);
        __syncthreads();
        ++sweeps;
        const int f1 = flags[1];
        __syncthreads();
        if (!f1) { converged = 1; break; }
    }
    {   // singular values = row norms, sorted descending by rank counting (ties: the lower row first)
        double xp[8], xq[8], a, b, g;
#pragma unroll
        for (int e = 0; e < 8; ++e) { xp[e] = X[(2 * slot) * P + sub * 8 + e]; xq[e] = X[(2 * slot + 1) * P + sub * 8 + e]; }
        seg_dot3(xp, xq, a, b, g);
        if (sub == 0) { nrm[2 * slot] = a; nrm[2 * slot + 1] = b; }
        __syncthreads();
        if (tid < NV) {
            const double mine = nrm[tid];
            int rank = 0;
            for (int r = 0; r < NV; ++r) rank += (nrm[r] > mine || (nrm[r] == mine && r < tid)) ? 1 : 0;
            ssort[rank] = sqrt(mine);
            order[rank] = tid;
        }
        __syncthreads();
        if (tid == 0) {
            const double s0 = ssort[0];
            int keep = 0;
            for (int i = 0; i < nv; ++i) keep += (ssort[i] > s0 * t) ? 1 : 0;
            if (keep > Dmax) keep = Dmax;
            double d2 = 0.0;
            for (int i = nv - 1; i >= keep; --i) d2 += ssort[i] * ssort[i];
            s_keep = keep;
            res[0] = (double)keep;
            res[1] = s0 > 0.0 ? sqrt(d2) / s0 : 0.0;
            res[2] = (double)sweeps;
            res[3] = (double)converged;
        }
        __syncthreads();
    }
    // kept vector j (source row order[j]): 8 threads per vector, two rounds of 32 vectors
    const int keep = s_keep;
    for (int j = slot; j < keep; j += 32) {
        const int row = order[j];
        const double sv = ssort[j];
        double xmin = 1e308, xmax = -1e308, pmin = 1e308, pmax = -1e308;
        for (int c = sub; c < L; c += 8) { const double v = X[row * P + c]; xmin = fmin(xmin, v); xmax = fmax(xmax, v); }
        for (int c = sub; c < nv; c += 8) { const double v = Pm[row * P + c]; pmin = fmin(pmin, v); pmax = fmax(pmax, v); }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            xmin = fmin(xmin, __shfl_xor(xmin, o, 64)); xmax = fmax(xmax, __shfl_xor(xmax, o, 64));
            pmin = fmin(pmin, __shfl_xor(pmin, o, 64)); pmax = fmax(pmax, __shfl_xor(pmax, o, 64));
        }
        const double sg = (fabs(pmin) > pmax && fabs(xmin) > xmax) ? -1.0 : 1.0;
        const double inv = sv > 0.0 ? sg / sv : 0.0;
        if (sub == 0) Sout[j] = sv * dscale;
        for (int c = sub; c < L; c += 8) right[(int64_t)j * rrs + (int64_t)c * rcs] = X[row * P + c] * inv;
        for (int c = sub; c < nv; c += 8) left[(int64_t)c * lrs + (int64_t)j * lcs] = Pm[row * P + c] * sg;
    }
}

// Page-locked slot 3 stages the asynchronous uploads of this file (tournament schedule, kept values / order).  An upload is only
// guaranteed to have been consumed once ITS stream has passed a synchronisation; a thread that moves on to another stream must not
// overwrite the slot while the previous stream may still be waiting to read it.
static int upload_slot_guard(hipStream_t st) {
    thread_local hipStream_t last = nullptr;
    thread_local bool pending = false;
    if (pending && last != st) {
        const hipError_t e = hipStreamSynchronize(last);
        if (e != hipSuccess) return hip_fail(e, "sync previous upload stream");
    }
    last = st;
    pending = true;
    return 0;
}

static void round_robin(int nblk, std::vector<int>& pairs) {     // (nblk-1) rounds x (nblk/2) pairs x 2
    std::vector<int> idx(nblk);
    for (int i = 0; i < nblk; ++i) idx[i] = i;
    pairs.clear();
    for (int r = 0; r < nblk - 1; ++r) {
        for (int i = 0; i < nblk / 2; ++i) {
            int a = idx[i], b = idx[nblk - 1 - i];
            pairs.push_back(a < b ? a : b);
            pairs.push_back(a < b ? b : a);
        }
        const int last = idx[nblk - 1];
        for (int i = nblk - 1; i > 1; --i) idx[i] = idx[i - 1];
        idx[1] = last;
    }
}

struct SvdWs {
    double *X, *P, *part, *Js, *maxoff, *norms, *Ssorted;
    int *live, *pairs, *nrot, *order;
};

static int64_t svd_layout(int64_t nv, int64_t L, bool vectors, char* base, SvdWs* w) {
    const int64_t nvp = align_up(nv, 2 * SVD_W), nblk = nvp / SVD_W, ng = nblk / 2, nr = nblk - 1;
    const int nchunk = gram_nchunk(L);
    int64_t off = 0;
    auto take = [&](int64_t bytes) { int64_t o = off; off += align_up(bytes, 256); return base ? base + o : nullptr; };
    // X (nvp x L) and the accumulator P (nvp x nv) share rows of one (nvp x (L + nv)) array, so that one GEMM applies a
    // round's rotations to both
    double* X = (double*)take(nvp * (L + (vectors ? nv : 0)) * 8);
    double* P = X ? X + L : nullptr;        // size queries run the layout with a null base
    double* part = (double*)take(ng * nchunk * 4 * SVD_W * SVD_W * 8);
    double* Js = (double*)take(ng * 4 * SVD_W * SVD_W * 8);
    double* maxoff = (double*)take(nr * ng * 8);
    double* norms = (double*)take(nvp * 8);
    double* Ss = (double*)take(nvp * 8);
    int* live = (int*)take(nvp * 4);
    int* pairs = (int*)take(nr * ng * 2 * 4);
    int* nrot = (int*)take(ng * 4);
    int* order = (int*)take(nvp * 4);
    if (w) { w->X = X; w->P = P; w->part = part; w->Js = Js; w->maxoff = maxoff; w->norms = norms; w->Ssorted = Ss;
             w->live = live; w->pairs = pairs; w->nrot = nrot; w->order = order; }
    return off;
}

int64_t svd_ws_bytes(int64_t k, int64_t n, int vectors) {
    const int64_t nv = k <= n ? k : n, L = k <= n ? n : k;
    return svd_layout(nv, L, vectors != 0, nullptr, nullptr);
}

// Core: orthogonalise the rows of M (nv x L, strides vs/es).  On return hostS holds the singular values sorted
// descending (length nvl), *nvl_out the number of live vectors, order[] (device) the matching row permutation.
// rel_tol: the relative threshold below which singular values are of no interest to the caller (the truncation rule keeps
// S > S0 max(eps, tol), mps.py:805-806).  Vectors a factor 4 below it (relative to the largest input row) are still rotated
// whenever a group needs rotating, but they do not hold up convergence: the rows of a triangular factor beyond its
// numerical rank are rounding noise of the QR (typically 2/3 of the live rows at 1e-16..1e-17 of the scale), a generic
// dense cluster on which cyclic Jacobi needs 15-25 sweeps, and every one of them is discarded afterwards.  What is returned
// for the kept vectors is unaffected: U and V^T stay orthonormal (J is orthogonal, the kept rows are mutually orthogonal),
// and the discarded weight is the Frobenius norm of the remaining rows, which rotations do not change.
static int jacobi_core(hipStream_t st, const double* M, int64_t vs, int64_t es, int64_t nv, int64_t L, bool vectors,
                       SvdWs& w, std::vector<double>& hostS, std::vector<int>& hostOrder, int* sweeps_out, int* info,
                       double rel_tol) {
    hipError_t e;
    int rc;
    TN_PROF_LAUNCH(st, PROF_SVD_AUX, hipLaunchKernelGGL(vec_norm2_kernel, dim3((unsigned)nv), dim3(256), 0, st, M, vs, es, L, w.norms));
    TN_CHECK_LAUNCH("vec_norm2_kernel");
    std::vector<double> hn(nv);
    {   // read-backs go through a page-locked staging buffer (see pinned_host): queued right behind the producing kernel
        double* stage = (double*)pinned_host((size_t)nv * 8, 2);
        if ((e = hipMemcpyAsync(stage ? stage : hn.data(), w.norms, nv * 8, hipMemcpyDeviceToHost, st)) != hipSuccess) return hip_fail(e, "memcpy norms");
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return hip_fail(e, "sync norms");
        if (stage) std::memcpy(hn.data(), stage, (size_t)nv * 8);
    }
    double nmax = 0.0;
    for (int64_t i = 0; i < nv; ++i) {
        if (!(hn[i] == hn[i]) || hn[i] > 1.7e308) { set_error("svd: non-finite input"); return -2; }
        nmax = std::max(nmax, hn[i]);
    }
    std::vector<int> live;
    const double thr = nmax * (1.9259299443872359e-34);      // (2^-56)^2 on squared norms
    for (int64_t i = 0; i < nv; ++i)
        if (hn[i] > thr) live.push_back((int)i);
    if (live.empty()) live.push_back(0);                      // all-zero input: one (zero) vector
    {   // TN_SVD_NORMS=1 (diagnostics): how the input vectors' norms are spread below the largest one
        static const bool norms_trace = [] { const char* e = getenv("TN_SVD_NORMS"); return e && e[0] == '1'; }();
        if (norms_trace) {
            int c[7] = {0, 0, 0, 0, 0, 0, 0};
            const int ex[7] = {-56, -54, -53, -52, -50, -44, -30};
            for (int64_t i = 0; i < nv; ++i)
                for (int t = 0; t < 7; ++t)
                    if (hn[i] > nmax * std::ldexp(1.0, 2 * ex[t])) ++c[t];
            fprintf(stderr, "[tn_svd norms] nv=%lld L=%lld above 2^-56:%d -54:%d -53:%d -52:%d -50:%d -44:%d -30:%d\n", (long long)nv, (long long)L,
                    c[0], c[1], c[2], c[3], c[4], c[5], c[6]);
        }
    }
    // de Rijk ordering: vectors enter the tournament sorted by decreasing norm, so that the blocks are graded (large
    // vectors meet large ones first) -- the classical remedy for the slow start of Jacobi on strongly graded matrices.
    // Only the initial order changes (TN_SVD_SORT=0 keeps the input order for A/B measurements).
    static const bool sort_live = [] { const char* e = getenv("TN_SVD_SORT"); return !(e && e[0] == '0'); }();
    if (sort_live) std::stable_sort(live.begin(), live.end(), [&](int a, int b) { return hn[a] > hn[b]; });
    const int nvl = (int)live.size();
    const int64_t nvp = align_up(nvl, 2 * SVD_W);
    const int nblk = (int)(nvp / SVD_W), ng = nblk / 2, nr = nblk - 1;
    // live list and round-robin schedule go up in ONE copy: w.live and w.pairs are neighbours in the workspace; the host
    // buffer lives until the end of this function, past the first stream synchronisation below
    std::vector<int> pairs;
    round_robin(nblk, pairs);
    const size_t gap = (size_t)((char*)w.pairs - (char*)w.live) / 4;
    std::vector<int> hinit(gap + pairs.size(), 0);
    std::copy(live.begin(), live.end(), hinit.begin());
    std::copy(pairs.begin(), pairs.end(), hinit.begin() + gap);
    {
        if ((rc = upload_slot_guard(st))) return rc;
        int* stage = (int*)pinned_host(hinit.size() * 4, 3);
        if (stage) std::memcpy(stage, hinit.data(), hinit.size() * 4);
        if ((e = hipMemcpyAsync(w.live, stage ? stage : hinit.data(), hinit.size() * 4, hipMemcpyHostToDevice, st)) != hipSuccess) return hip_fail(e, "memcpy init");
    }
    const int64_t pitch = L + (vectors ? nv : 0);
    TN_PROF_LAUNCH(st, PROF_SVD_AUX, hipLaunchKernelGGL(svd_init_kernel, dim3((unsigned)nvp), dim3(256), 0, st, M, vs, es, L, nv, w.live, nvl, w.X,
                       vectors ? w.P : nullptr, pitch));
    TN_CHECK_LAUNCH("svd_init_kernel");

    const int nchunk = gram_nchunk(L);
    const int nvec = 2 * SVD_W;
    std::vector<double> hoff((size_t)nr * ng);
    // a single block pair is diagonalised completely inside eig_small (one round + one verification round);
    // with several pairs two inner sweeps per visit give the fewest total Jacobi steps
    static const int inner_env = [] { const char* e = getenv("TN_SVD_INNER"); return e ? atoi(e) : 2; }();
    // ... in the FIRST outer sweep, where the rotations are large; from the second outer sweep on the off-diagonals a visit meets are
    // small (quadratic convergence), a second inner sweep finds next to nothing and the next outer sweep has to look at the pair again
    // anyway -- measured neutral at L = 2048 (eig_small 481 -> 440 ms per sweep, but 1964 -> 2189 outer sweeps): TN_SVD_INNER_LATER
    // stays at 2
    static const int inner_later = [] { const char* e = getenv("TN_SVD_INNER_LATER"); return e ? atoi(e) : 2; }();
    static const bool restrict_conv = [] { const char* e = getenv("TN_SVD_RELEVANT"); return !(e && e[0] == '0'); }();
    const double rel4 = 0.25 * rel_tol;
    const double relevant2 = restrict_conv ? nmax * rel4 * rel4 : 0.0;
    static const double last_tol = [] { const char* e = getenv("TN_SVD_LAST"); return e ? atof(e) : 1e-9; }();
    static const bool trace = [] { const char* e = getenv("TN_SVD_TRACE"); return e && e[0] == '1'; }();
    int sweeps = 0;
    bool converged = false;
    std::vector<double> hs(nvp);
    // All rounds of all sweeps, the convergence tests and the final norms in ONE launch (svdl_kernel, small.hip) -- the same arithmetic
    // as the loop below, bit for bit.  If one of its barriers gave up (workgroups not co-resident) the vectors are set up again and the
    // loop below does the work; the stream stays off the single-launch forms from then on.
    bool fused_done = false;
    if (!trace) {
        SvdRoundsJob j;
        j.X = w.X; j.pitch = pitch; j.L = L; j.nvp = (int)nvp; j.w = SVD_W;
        j.pairs = w.pairs; j.ng = ng; j.nr = nr; j.nchunk = nchunk;
        j.part = w.part; j.part_bytes = (int64_t)ng * nchunk * nvec * nvec * 8; j.Js = w.Js; j.nrot = w.nrot; j.maxoff = w.maxoff;
        j.relevant2 = relevant2; j.last_tol = last_tol; j.inner_first = inner_env; j.inner_later = inner_later;
        j.norms = w.norms;                                            // (w.Ssorted follows it: room for the status words)
        rc = svd_rounds_fused(st, j);
        if (rc != 0 && rc != 1) return rc;
        if (rc == 0) {
            std::vector<double> hb(nvp + 4);
            double* stage = (double*)pinned_host((size_t)(nvp + 4) * 8, 2);
            if ((e = hipMemcpyAsync(stage ? stage : hb.data(), w.norms, (nvp + 4) * 8, hipMemcpyDeviceToHost, st)) != hipSuccess) return hip_fail(e, "memcpy S");
            if ((e = hipStreamSynchronize(st)) != hipSuccess) return hip_fail(e, "sync S");
            if (stage) std::memcpy(hb.data(), stage, (size_t)(nvp + 4) * 8);
            if (hb[nvp + 2] != 0.0) {
                fprintf(stderr, "[libtnpeps] the one-launch Jacobi rounds of an SVD gave up at a barrier on stream %p (%d workgroup(s); workgroups not "
                        "co-resident: is the device shared? see TN_PANEL_CU_BUDGET); the rounds are redone as separate launches, which this stream "
                        "uses from now on\n", (void*)st, (int)hb[nvp + 2]);
                fused_forms_disable(st);
                TN_PROF_LAUNCH(st, PROF_SVD_AUX, hipLaunchKernelGGL(svd_init_kernel, dim3((unsigned)nvp), dim3(256), 0, st, M, vs, es, L, nv, w.live, nvl, w.X,
                                   vectors ? w.P : nullptr, pitch));
                TN_CHECK_LAUNCH("svd_init_kernel");
            } else {
                sweeps = (int)hb[nvp];
                converged = hb[nvp + 1] != 0.0;
                std::copy(hb.begin(), hb.begin() + nvp, hs.begin());
                fused_done = true;
            }
        }
    }
    for (int outer = 0; outer < 40 && !converged && !fused_done; ++outer) {
        for (int r = 0; r < nr; ++r) {
            const int* pr = w.pairs + (int64_t)r * ng * 2;
            // Gram matrices of all block pairs on the matrix cores: G = Xp Xp^T, split over L with the partial sums left
            // in w.part (eig_small adds them up); then X_pair <- J^T X_pair (and the same on P) as in-place GEMMs
            GemmExtra xg;
            int used = 1;
            xg.pairs = pr; xg.pw = SVD_W; xg.mapA = 1; xg.mapB = 2; xg.force_splitk = nchunk; xg.raw_partials = true;
            xg.splitk_used = &used;
            if ((rc = gemm_ex(st, nvec, nvec, L, 1.0, w.X, pitch, 1, w.X, 1, pitch, 0.0, nullptr, 0, 0, ng, 0, 0, 0, w.part,
                              (int64_t)ng * nchunk * nvec * nvec * 8, &xg)))
                return rc;
            if ((rc = eig_small(st, w.part, used, nvec, ng, 2, (ng == 1) ? 12 : (outer == 0 ? inner_env : inner_later), 0.0, w.Js, nullptr, w.nrot,
                                w.maxoff + (int64_t)r * ng, relevant2, outer < 8 ? 1 : 0)))      // (a call that drags on goes back to the plain sweeps)
                return rc;
            GemmExtra xa;
            xa.pairs = pr; xa.pw = SVD_W; xa.mapB = 1; xa.mapC = 1; xa.skip = w.nrot;
            if ((rc = gemm_ex(st, nvec, pitch, nvec, 1.0, w.Js, 1, nvec, w.X, pitch, 1, 0.0, w.X, pitch, 1, ng, (int64_t)nvec * nvec,
                              0, 0, nullptr, 0, &xa)))                 // [X | P] <- J^T [X | P] in one launch
                return rc;
        }
        ++sweeps;
        {
            double* stage = (double*)pinned_host(hoff.size() * 8, 2);
            if ((e = hipMemcpyAsync(stage ? stage : hoff.data(), w.maxoff, hoff.size() * 8, hipMemcpyDeviceToHost, st)) != hipSuccess) return hip_fail(e, "memcpy maxoff");
            if ((e = hipStreamSynchronize(st)) != hipSuccess) return hip_fail(e, "sync sweep");
            if (stage) std::memcpy(hoff.data(), stage, hoff.size() * 8);
        }
        double worst = 0.0;
        for (double v : hoff) worst = std::max(worst, v);
        if (trace) {
            fprintf(stderr, "[tn_svd] nv=%lld L=%lld live=%d sweep=%d worst=%.3e offs", (long long)nv, (long long)L, nvl, sweeps, worst);
            for (double v : hoff) fprintf(stderr, " %.2e", v);
            fprintf(stderr, "\n");
        }
        // Quadratic convergence: a sweep that met nothing above TN_SVD_LAST (1e-9) leaves off-diagonals of its square (each visited pair is
        // diagonalised to rounding by the two inner sweeps, cross terms are products of two such numbers), so the verification sweep
        // that would follow finds every pair below 4e-15 and rotates nothing -- it is skipped (8-10 % of the rounds of a sweep of the
        // headline workload); the results are the same bit for bit unless that sweep would have found something, which 1e-18 cannot be.
        converged = worst < 4.0e-15 || worst <= last_tol;
    }
    if (sweeps_out) *sweeps_out = sweeps;
    if (info) *info = converged ? 0 : 1;
    prof_note(PROF_SVD_ROUNDS, (double)sweeps * nr, (double)sweeps * nr * ng, 0.0);
    if (!fused_done) {
        TN_PROF_LAUNCH(st, PROF_SVD_AUX, hipLaunchKernelGGL(vec_norm2_kernel, dim3((unsigned)nvp), dim3(256), 0, st, w.X, pitch, 1, L, w.norms));
        TN_CHECK_LAUNCH("vec_norm2_kernel");
        double* stage = (double*)pinned_host((size_t)nvp * 8, 2);
        if ((e = hipMemcpyAsync(stage ? stage : hs.data(), w.norms, nvp * 8, hipMemcpyDeviceToHost, st)) != hipSuccess) return hip_fail(e, "memcpy S");
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return hip_fail(e, "sync S");
        if (stage) std::memcpy(hs.data(), stage, (size_t)nvp * 8);
    }
    hostOrder.resize(nvp);
    for (int i = 0; i < nvp; ++i) hostOrder[i] = i;
    std::stable_sort(hostOrder.begin(), hostOrder.end(), [&](int a, int b) { return hs[a] > hs[b]; });
    hostS.resize(nvl);
    hostOrder.resize(nvl);     // padding rows are exactly zero and sort last
    for (int i = 0; i < nvl; ++i) hostS[i] = std::sqrt(hs[hostOrder[i]]);
    return 0;
}

// C is k x n (element strides crs, ccs).  U: k x keep (urs, ucs), Vt: keep x n (vrs, vcs), S: keep values.
// keep_out / discarded_out / hostS_out are host pointers.
// TN_SVD_CENSUS=1 (diagnostics): every tn_svd_trunc call is timed synchronously and booked under its shape; table at exit.
namespace {
struct SvdCensus {
    bool on;
    std::mutex mu;
    std::map<std::pair<int64_t, int64_t>, std::tuple<double, long, double>> tab;       // (k, n) -> ms, calls, sweeps
    SvdCensus() { const char* e = getenv("TN_SVD_CENSUS"); on = e && e[0] == '1'; }
    ~SvdCensus() {
        if (!on || tab.empty()) return;
        std::vector<std::pair<double, std::pair<int64_t, int64_t>>> v;
        double tot = 0.0;
        long calls = 0;
        for (auto& kv : tab) { v.push_back({std::get<0>(kv.second), kv.first}); tot += std::get<0>(kv.second); calls += std::get<1>(kv.second); }
        std::sort(v.begin(), v.end(), [](auto& a, auto& b) { return a.first > b.first; });
        fprintf(stderr, "[tn_svd census] %zu shapes, %ld calls, %.1f ms in total; k n : calls, ms, us/call, mean sweeps\n", v.size(), calls, tot);
        for (auto& e : v) {
            auto& t = tab[e.second];
            fprintf(stderr, "  %5lld %5lld : %5ld  %8.2f  %8.2f  %5.2f\n", (long long)e.second.first, (long long)e.second.second, std::get<1>(t), std::get<0>(t),
                    1e3 * std::get<0>(t) / std::get<1>(t), std::get<2>(t) / std::get<1>(t));
        }
    }
};
SvdCensus g_svd_census;
}  // namespace

static int svd_trunc_impl(hipStream_t st, const double* C, int64_t crs, int64_t ccs, int64_t k, int64_t n, int64_t Dmax, double tol,
                          double* U, int64_t urs, int64_t ucs, double* S, double* Vt, int64_t vrs, int64_t vcs, int64_t* keep_out,
                          double* discarded_out, int* sweeps_out, int* info, void* ws, int64_t ws_bytes);

int svd_trunc(hipStream_t st, const double* C, int64_t crs, int64_t ccs, int64_t k, int64_t n, int64_t Dmax, double tol,
              double* U, int64_t urs, int64_t ucs, double* S, double* Vt, int64_t vrs, int64_t vcs, int64_t* keep_out,
              double* discarded_out, int* sweeps_out, int* info, void* ws, int64_t ws_bytes) {
    if (!g_svd_census.on)
        return svd_trunc_impl(st, C, crs, ccs, k, n, Dmax, tol, U, urs, ucs, S, Vt, vrs, vcs, keep_out, discarded_out, sweeps_out, info, ws, ws_bytes);
    thread_local hipEvent_t e0 = nullptr, e1 = nullptr;
    if (!e0) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); }
    int sw = 0;
    (void)hipEventRecord(e0, st);
    const int rc = svd_trunc_impl(st, C, crs, ccs, k, n, Dmax, tol, U, urs, ucs, S, Vt, vrs, vcs, keep_out, discarded_out, &sw, info, ws, ws_bytes);
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    if (sweeps_out) *sweeps_out = sw;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::lock_guard<std::mutex> lk(g_svd_census.mu);
    auto& t = g_svd_census.tab[{k, n}];
    std::get<0>(t) += ms; std::get<1>(t) += 1; std::get<2>(t) += sw;
    return rc;
}

static int svd_trunc_impl(hipStream_t st, const double* C, int64_t crs, int64_t ccs, int64_t k, int64_t n, int64_t Dmax, double tol,
                          double* U, int64_t urs, int64_t ucs, double* S, double* Vt, int64_t vrs, int64_t vcs, int64_t* keep_out,
                          double* discarded_out, int* sweeps_out, int* info, void* ws, int64_t ws_bytes) {
    TN_CHECK_ARG(k >= 1 && n >= 1 && Dmax >= 1, "bad dimensions");
    TN_CHECK_ARG(ws_bytes >= svd_ws_bytes(k, n, 1), "workspace too small");
    const bool rows = k <= n;
    const int64_t nv = rows ? k : n, L = rows ? n : k;
    const int64_t vs = rows ? crs : ccs, es = rows ? ccs : crs;
    {   // both dimensions <= 64: the whole truncated SVD in one launch and one read-back (TN_SVD_SMALL=0: the block path; read per call)
        const char* e_small = getenv("TN_SVD_SMALL");
        if (k <= 64 && n <= 64 && !(e_small && e_small[0] == '0')) {
            const double eps = 2.220446049250313e-16;
            const double t = tol > eps ? tol : eps;
            double* res = (double*)ws;
            // rows: left = U (k x keep), right = Vt.  columns (C^T was factored): left = Vt^T, right = U^T.
            if (rows)
                TN_PROF_LAUNCH(st, PROF_SVD_AUX, hipLaunchKernelGGL(svd_trunc_small_kernel, dim3(1), dim3(256), 0, st, C, vs, es, (int)nv, (int)L, (int)std::min<int64_t>(Dmax, 64), t,
                                   S, U, urs, ucs, Vt, vrs, vcs, res));
            else
                TN_PROF_LAUNCH(st, PROF_SVD_AUX, hipLaunchKernelGGL(svd_trunc_small_kernel, dim3(1), dim3(256), 0, st, C, vs, es, (int)nv, (int)L, (int)std::min<int64_t>(Dmax, 64), t,
                                   S, Vt, vcs, vrs, U, ucs, urs, res));
            TN_CHECK_LAUNCH("svd_trunc_small_kernel");
            double h[4] = {0.0, 0.0, 0.0, 0.0};
            double* stage = (double*)pinned_host(sizeof(h), 2);
            hipError_t e;
            if ((e = hipMemcpyAsync(stage ? stage : h, res, sizeof(h), hipMemcpyDeviceToHost, st)) != hipSuccess) return hip_fail(e, "memcpy result");
            if ((e = hipStreamSynchronize(st)) != hipSuccess) return hip_fail(e, "sync result");
            if (stage) std::memcpy(h, stage, sizeof(h));
            if (!(h[1] == h[1])) { set_error("svd: non-finite input"); return -2; }
            if (keep_out) *keep_out = (int64_t)h[0];
            if (discarded_out) *discarded_out = h[1];
            if (sweeps_out) *sweeps_out = (int)h[2];
            if (info) *info = h[3] != 0.0 ? 0 : 1;
            return 0;
        }
    }
    SvdWs w;
    svd_layout(nv, L, true, (char*)ws, &w);
    std::vector<double> hS;
    std::vector<int> hO;
    const double eps = 2.220446049250313e-16;
    const double t = tol > eps ? tol : eps;
    int rc = jacobi_core(st, C, vs, es, nv, L, true, w, hS, hO, sweeps_out, info, t);
    if (rc) return rc;
    const int nvl = (int)hS.size();
    int64_t keep = 0;
    for (int i = 0; i < nvl; ++i) keep += (hS[i] > hS[0] * t) ? 1 : 0;
    if (keep > Dmax) keep = Dmax;
    double d2 = 0.0;
    for (int i = nvl - 1; i >= keep; --i) d2 += hS[i] * hS[i];
    if (keep_out) *keep_out = keep;
    if (discarded_out) *discarded_out = std::sqrt(d2) / hS[0];
    if (keep == 0) return 0;
    hipError_t e;
    GatherList gl = {};
    const int byval = keep <= 64 ? 1 : 0;
    double* dS = w.norms;
    const int* dO = (const int*)((const char*)w.norms + (size_t)keep * 8);
    char* stage = nullptr;
    if (byval) {
        for (int64_t i = 0; i < keep; ++i) { gl.S[i] = hS[(size_t)i]; gl.order[i] = hO[(size_t)i]; }
    } else {
        // kept singular values and their row order go up in one copy into the (now free) norms | Ssorted area: [S | order]
        std::vector<char> hpack((size_t)keep * 12);
        std::memcpy(hpack.data(), hS.data(), (size_t)keep * 8);
        std::memcpy(hpack.data() + (size_t)keep * 8, hO.data(), (size_t)keep * 4);
        if ((rc = upload_slot_guard(st))) return rc;
        stage = (char*)pinned_host(hpack.size(), 3);           // (the schedule uploaded from this slot completed several synchronisations ago)
        if (stage) std::memcpy(stage, hpack.data(), hpack.size());
        if ((e = hipMemcpyAsync(w.norms, stage ? stage : hpack.data(), hpack.size(), hipMemcpyHostToDevice, st)) != hipSuccess) return hip_fail(e, "memcpy S/order");
        if (!stage && (e = hipStreamSynchronize(st)) != hipSuccess) return hip_fail(e, "sync S/order");      // pageable source: must outlive the copy
    }
    // rows: left = U (k x keep), right = Vt.  columns (C^T was factored): left = Vt^T, right = U^T.
    if (rows)
        TN_PROF_LAUNCH(st, PROF_SVD_AUX, hipLaunchKernelGGL(svd_gather_kernel, dim3((unsigned)keep), dim3(256), 0, st, w.X, L, w.P, nv, L + nv, dO, dS, S,
                           U, urs, ucs, Vt, vrs, vcs, gl, byval));
    else
        TN_PROF_LAUNCH(st, PROF_SVD_AUX, hipLaunchKernelGGL(svd_gather_kernel, dim3((unsigned)keep), dim3(256), 0, st, w.X, L, w.P, nv, L + nv, dO, dS, S,
                           Vt, vcs, vrs, U, ucs, urs, gl, byval));
    TN_CHECK_LAUNCH("svd_gather_kernel");
    return 0;
}

// Values-only SVD of a centre matrix with both dimensions <= 64, without any read-back: out (DEVICE, 66 doubles) receives the
// 64 values sorted descending (zero padded), the executed sweeps and the convergence flag.  One launch, asynchronous.
int svd_vals_small_async(hipStream_t st, const double* C, int64_t crs, int64_t ccs, int64_t k, int64_t n, double* out) {
    TN_CHECK_ARG(k >= 1 && n >= 1 && k <= 64 && n <= 64, "both dimensions must be in 1..64");
    const bool rows = k <= n;
    const int64_t nv = rows ? k : n, L = rows ? n : k;
    const int64_t vs = rows ? crs : ccs, es = rows ? ccs : crs;
    TN_PROF_LAUNCH(st, PROF_SVD_AUX, hipLaunchKernelGGL(svd_vals_small_kernel, dim3(1), dim3(256), 0, st, C, vs, es, (int)nv, (int)L, out));
    TN_CHECK_LAUNCH("svd_vals_small_kernel");
    return 0;
}

// `batch` centre matrices (both dimensions <= 64 each) in ONE launch, one workgroup per matrix: desc (DEVICE, 5 int64 per item:
// address, vector stride, element stride, number of vectors <= length, length) as prepared by the caller from
// tn_svdvals_small_desc; out (DEVICE, 66 doubles per item) as for svd_vals_small_async.
int svd_vals_small_batched(hipStream_t st, const int64_t* desc, int64_t batch, double* out) {
    TN_CHECK_ARG(batch >= 0, "negative batch");
    if (batch == 0) return 0;
    TN_PROF_LAUNCH(st, PROF_SVD_AUX, hipLaunchKernelGGL(svd_vals_small_batched_kernel, dim3((unsigned)batch), dim3(256), 0, st, desc, out));
    TN_CHECK_LAUNCH("svd_vals_small_batched_kernel");
    return 0;
}

// Singular values only, sorted descending, min(k,n) of them (deflated ones reported as 0).  hostS: host pointer.
int svd_vals(hipStream_t st, const double* C, int64_t crs, int64_t ccs, int64_t k, int64_t n, double* hostS,
             int* sweeps_out, int* info, void* ws, int64_t ws_bytes) {
    TN_CHECK_ARG(k >= 1 && n >= 1, "bad dimensions");
    TN_CHECK_ARG(ws_bytes >= svd_ws_bytes(k, n, 0), "workspace too small");
    const bool rows = k <= n;
    const int64_t nv = rows ? k : n, L = rows ? n : k;
    const int64_t vs = rows ? crs : ccs, es = rows ? ccs : crs;
    if (L <= 64) {               // small centre matrix: everything in one launch and one read-back
        double* dout = (double*)ws;
        TN_PROF_LAUNCH(st, PROF_SVD_AUX, hipLaunchKernelGGL(svd_vals_small_kernel, dim3(1), dim3(256), 0, st, C, vs, es, (int)nv, (int)L, dout));
        TN_CHECK_LAUNCH("svd_vals_small_kernel");
        double h[66];
        hipError_t e;
        double* stage = (double*)pinned_host(sizeof(h), 2);
        if ((e = hipMemcpyAsync(stage ? stage : h, dout, sizeof(h), hipMemcpyDeviceToHost, st)) != hipSuccess) return hip_fail(e, "memcpy S");
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return hip_fail(e, "sync S");
        if (stage) std::memcpy(h, stage, sizeof(h));
        for (int64_t i = 0; i < nv; ++i) {
            if (!(h[i] == h[i]) || h[i] > 1.7e308) { set_error("svd: non-finite input"); return -2; }
            hostS[i] = h[i];
        }
        if (sweeps_out) *sweeps_out = (int)h[64];
        if (info) *info = (h[65] != 0.0) ? 0 : 1;
        return 0;
    }
    SvdWs w;
    svd_layout(nv, L, false, (char*)ws, &w);
    std::vector<double> hS;
    std::vector<int> hO;
    int rc = jacobi_core(st, C, vs, es, nv, L, false, w, hS, hO, sweeps_out, info, 2.220446049250313e-16);
    if (rc) return rc;
    for (int64_t i = 0; i < nv; ++i) hostS[i] = i < (int64_t)hS.size() ? hS[i] : 0.0;
    return 0;
}

}  // namespace tn
