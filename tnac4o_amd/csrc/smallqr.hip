// Small factorisations in ONE launch: explicit-Q iterated Cholesky-QR of an m x n matrix with n <= 64 columns.
//
// Replaces, for these shapes, the blocked Householder path of qr.hip behind the reference's `qr` (mps.py:43-59: economic QR with
// diag(R) >= 0) as called from the variational sweeps (mps.py:238-279) and the canonisation passes (mps.py:202-236).  A sweep of
// the headline workload makes ~900 such factorisations (1024 x 64-class site matrices); through the blocked path each is two panel
// launches plus ten auxiliary ones (trailing update, Q accumulation, R assembly, normalisation): 130-440 us of launch latency.
//
// Algorithm (the panel step of cholqr.hip at the full width, with the triangular factor kept):
//   every workgroup owns TR rows, held in LDS from the first load to the last store;
//   pass t:  G = X^T X (matrix cores; per-workgroup partials, summed in block order by slices: workgroup w adds slice w of all
//            partials, the slices meet in one n x n matrix)  ->  every workgroup factors G = R_t^T R_t itself (one wave, Cholesky with
//            deferral, same bits everywhere)  ->  X <- X R_t^-1 (substitution on the rows of the tile)  ->  R <- R_t R;
//   passes repeat until G is the identity to rounding (cholqr.hip: CQ_DONE / CQ_LAST), then Q = X and R go out, R optionally
//   divided by its power-of-two norm factor (mps.py:76-85, what site_qr does next anyway).
// What makes this safe on rank-deficient / graded inputs is cholqr.hip's scheme: block-wise power-of-two scaling, deferral of
// pivots that have lost their digits (row j of R_t = e_j: the substitution leaves the exact residual, the next pass treats it as
// an ordinary column), exactly zero columns refilled with noise (and their row of R zeroed: A = Q R stays exact).  The
// substitution is backward stable row by row, so A = Q (R_p ... R_1) holds column-wise to rounding whatever the conditioning;
// orthogonality is what the passes iterate on.  If SQ_MAXPASS passes do not converge, workgroup 0 redoes the factorisation with
// Householder reflections in global memory (slow, never seen on the contraction path; driven in the tests with TN_PANEL_MAXPASS).
//
// Workgroups meet at in-kernel barriers (monotone counter, one polling lane, agent-scope stores / loads: the protocol of
// cq_fused_kernel, same co-residency budget: at most 32 workgroups per launch).  Bounded spins; a launch that gives up poisons its
// outputs with NaN and books a time-out (cholqr_timeouts), which the callers check.
#include <stdlib.h>

#include <mutex>

#include "common.h"

// -DSQ_CLOCKS: thread 0 of workgroup 0 records the 100 MHz wall clock at phase boundaries (diagnostics only: tools/smallqr_clocks.py)
#ifdef SQ_CLOCKS
__device__ long long sq_clk[32];
extern "C" int tn_debug_sq_clocks(long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sq_clk), sizeof(long long) * (n < 32 ? n : 32));
}
#define SQ_CLK(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) sq_clk[k] = wall_clock64(); } while (0)
#else
#define SQ_CLK(k) do {} while (0)
#endif

namespace tn {

typedef double d4s __attribute__((ext_vector_type(4)));

constexpr int SQ_MAXBLK = 32;
constexpr int SQ_MAXPASS = 4;
constexpr double SQ_THETA = 1e-10;
constexpr double SQ_DONE = 5e-15;
constexpr double SQ_LAST = 1e-8;

struct SqState { int counter; int exits; int pad0; int pad1; };
__device__ SqState sq_state_pool[CHOLQR_SLOTS];
// [0] calls  [1] passes applied  [2] Householder fallbacks  [3] launches that gave up at a barrier
__device__ unsigned long long sq_stats[CHOLQR_SLOTS * 4];

__device__ __forceinline__ double sq_ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sq_st(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int sq_ldi(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sq_sti(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sq_publish_wait() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_waitcnt(0); }

__device__ __forceinline__ double sq_readlane(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sq_rsqrt2(double x) {
    double r = __builtin_amdgcn_rsq(x);
    r = r * (1.5 - 0.5 * x * r * r);
    r = r * (1.5 - 0.5 * x * r * r);
    return r;
}
__device__ __forceinline__ void sq_fnma(double& acc, double a, double b) {
    asm volatile("v_fma_f64 %0, -%1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ double sq_hash_unit(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return ((double)(x >> 11) * (1.0 / 9007199254740992.0)) - 0.5;
}
// value of lane (quad base + SEL) of every quad (TPR = 4) / of lane (pair base + SEL) of every pair (TPR = 2), through the DPP network
template <int CTRL>
__device__ __forceinline__ double sq_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int TPR>
__device__ __forceinline__ double sq_group_bcast(double v, int owner) {      // owner: compile-time after unrolling
    if constexpr (TPR == 1) return v;
    else if constexpr (TPR == 2) return owner == 0 ? sq_dpp<0xA0>(v) : sq_dpp<0xF5>(v);       // quad_perm [0,0,2,2] / [1,1,3,3]
    else return owner == 0 ? sq_dpp<0x00>(v) : owner == 1 ? sq_dpp<0x55>(v) : owner == 2 ? sq_dpp<0xAA>(v) : sq_dpp<0xFF>(v);
}

__device__ __forceinline__ void sq_block_rows(int64_t nrows, int nblk, int blk, int64_t& r0, int& nr) {
    const int64_t base = nrows / nblk, rem = nrows % nblk;
    r0 = blk * base + (blk < rem ? blk : rem);
    nr = (int)(base + (blk < rem ? 1 : 0));
}

__device__ __forceinline__ bool sq_grid_barrier(int* counter, int target, int* s_flag, int tid, unsigned spin_limit) {
    sq_publish_wait();
    __syncthreads();
    if (tid == 0) {
        atomicAdd(counter, 1);
        int ok = 1;
        unsigned spins = 0;
        while (sq_ldi(counter) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > spin_limit) { ok = 0; break; }
        }
        *s_flag = ok;
    }
    __syncthreads();
    return *s_flag != 0;
}

__device__ __forceinline__ double sq_block_sum(double v, double* red, int tid) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// Last resort, one workgroup, everything in global memory: Householder QR (dgeqr2) of 2^-emax A in Q's storage, R = its upper
// triangle, then the explicit Q (dorg2r) in place, signs fixed so that diag(R) >= 0.  lds: >= 200 doubles.
__device__ void sq_fallback_householder(const double* A, int64_t ars, int64_t acs, double* Q, int64_t rs, int64_t cs, double* R, int64_t rrs,
                                        int64_t rcs, int64_t m, int n, int emax, double* lds, int tid) {
    double* red = lds;              // 4
    double* wv = lds + 8;           // 64 products
    double* taus = lds + 72;        // 64
    double* sg = lds + 136;         // 64 signs of diag(R)
    const double scl = (emax > -2000) ? ldexp(1.0, -emax) : 0.0, back = (emax > -2000) ? ldexp(1.0, emax) : 0.0;
    for (int64_t e = tid; e < m * n; e += 256) {
        const int64_t i = e / n, j = e % n;
        Q[i * rs + j * cs] = A[i * ars + j * acs] * scl;
    }
    __threadfence();
    __syncthreads();
    for (int j = 0; j < n; ++j) {
        double s = 0.0;
        for (int64_t r = j + 1 + tid; r < m; r += 256) { const double y = Q[r * rs + j * cs]; s += y * y; }
        s = sq_block_sum(s, red, tid);
        const double alpha = Q[j * rs + j * cs];
        double tau = 0.0;
        if (s > 1e-300) {
            const double beta = -copysign(sqrt(alpha * alpha + s), alpha);
            tau = (beta - alpha) / beta;
            const double inv = 1.0 / (alpha - beta);
            for (int64_t r = j + 1 + tid; r < m; r += 256) Q[r * rs + j * cs] *= inv;
            __threadfence();
            __syncthreads();
            for (int c = j + 1; c < n; ++c) {
                double w = 0.0;
                for (int64_t r = j + 1 + tid; r < m; r += 256) w += Q[r * rs + j * cs] * Q[r * rs + c * cs];
                w = sq_block_sum(w, red, tid);
                if (tid == 0) wv[c] = tau * (w + Q[j * rs + c * cs]);
                __syncthreads();
                const double tw = wv[c];
                for (int64_t r = j + 1 + tid; r < m; r += 256) Q[r * rs + c * cs] -= Q[r * rs + j * cs] * tw;
                if (tid == 0) Q[j * rs + c * cs] -= tw;
            }
            if (tid == 0) Q[j * rs + j * cs] = beta;
        }
        if (tid == 0) taus[j] = tau;
        __threadfence();
        __syncthreads();
    }
    // R = upper triangle (rows with a negative diagonal are negated, and so is the matching column of Q below)
    for (int e = tid; e < n * n; e += 256) {
        const int i = e / n, c = e % n;
        const double sg = (Q[(int64_t)i * rs + (int64_t)i * cs] < 0.0) ? -1.0 : 1.0;
        R[(int64_t)i * rrs + (int64_t)c * rcs] = (c >= i) ? sg * Q[(int64_t)i * rs + (int64_t)c * cs] * back : 0.0;
    }
    if (tid < n) sg[tid] = (Q[(int64_t)tid * rs + (int64_t)tid * cs] < 0.0) ? -1.0 : 1.0;       // read before Q is overwritten
    __threadfence();
    __syncthreads();
    for (int j = n - 1; j >= 0; --j) {
        const double tau = taus[j];
        for (int c = j + 1; c < n; ++c) {
            double w = 0.0;
            for (int64_t r = j + 1 + tid; r < m; r += 256) w += Q[r * rs + j * cs] * Q[r * rs + c * cs];
            w = sq_block_sum(w, red, tid);
            const double tw = tau * w;
            for (int64_t r = j + 1 + tid; r < m; r += 256) Q[r * rs + c * cs] -= Q[r * rs + j * cs] * tw;
            if (tid == 0) Q[j * rs + c * cs] = -tw;
            __threadfence();
            __syncthreads();
        }
        for (int64_t r = j + 1 + tid; r < m; r += 256) Q[r * rs + j * cs] *= -tau;
        if (tid == 0) Q[j * rs + j * cs] = 1.0 - tau;
        for (int64_t r = tid; r < j; r += 256) Q[r * rs + j * cs] = 0.0;
        __threadfence();
        __syncthreads();
    }
    for (int64_t e = tid; e < m * n; e += 256) {
        const int64_t i = e / n, j = e % n;
        if (sg[j] < 0.0) Q[i * rs + j * cs] = -Q[i * rs + j * cs];
    }
}

struct SqArgs {
    const double* A; int64_t ars, acs;
    int64_t m; int n;
    double* Q; int64_t qrs, qcs;
    double* R; int64_t rrs, rcs;
    double* nf_out2;            // optional: R /= nfactor(R), [nf, 1/nf] stored here
    double* part;               // 2 x nblk x N*N partial Gram matrices (pass parity)
    double* gsum;               // 2 x N*N summed Gram matrices (pass parity)
    int* bexp;                  // nblk block exponents
    SqState* stt;
    unsigned long long* stats;  // the stream's 4 counters
    uint64_t seed;
    int nblk, maxpass;
    unsigned spin_limit;
};

// N: padded width (32 or 64).  TR: rows per workgroup; 256 / TR threads share a row in the substitution.
template <int N, int TR>
__global__ __launch_bounds__(256) void sq_kernel(SqArgs a) {
    constexpr int P = N + 1;                   // LDS pitch of the tile and of the Gram matrix
    constexpr int TPR = 256 / TR;              // threads per row
    constexpr int NT = N / 16;                 // 16 x 16 tiles per side
    constexpr int NTILE = NT * (NT + 1) / 2;   // unique Gram tiles
    constexpr int EPT = TR * N / 256;          // tile elements per thread
    constexpr int XL = N / TPR;                // columns per thread in the substitution
    constexpr int RP = N + 2;                  // pitch of the factor (rows 16-byte aligned, fragment reads spread over the banks)
    __shared__ double T[TR * P];
    __shared__ double Gs[N * P];
    __shared__ __attribute__((aligned(16))) double Rf[N * RP + N];    // R_t row-major + reciprocal diagonal
    __shared__ double red[8];
    __shared__ int s_out[6];                   // [0] decision [1] final_next [2,3] dead mask [4] emax
    __shared__ int s_flag;
    const int tid = threadIdx.x, blk = blockIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int n = a.n, nblk = a.nblk;
    const bool multi = nblk > 1;
    int64_t r0;
    int nr;
    sq_block_rows(a.m, nblk, blk, r0, nr);
    int nbar = 0;
    bool alive = true;
    // ---- load: rows r0 .. r0+nr-1 into the tile, scaled by the power of two that brings its largest entry into [0.5, 1).
    // 16-byte accesses along the unit-stride direction when the layout allows (the epilogue is bound by store issue otherwise)
    int ex = -2000;
    auto wide_ok = [&](const double* base, int64_t srow, int64_t scol) -> bool {
        if (((uintptr_t)base & 15) != 0) return false;
        if (scol == 1) return (n & 1) == 0 && (srow & 1) == 0;
        if (srow == 1) return (scol & 1) == 0 && (r0 & 1) == 0 && (nr & 1) == 0;
        return false;
    };
    {
        const bool colfast = (a.acs == 1);
        const bool wide = wide_ok(a.A, a.ars, a.acs);
        double xv[EPT];
        double amax = 0.0;
        if (wide) {
#pragma unroll
            for (int u = 0; u < EPT / 2; ++u) {
                const int e = tid + 256 * u;
                const int i = colfast ? e / (N / 2) : 2 * (e % (TR / 2)), j = colfast ? 2 * (e % (N / 2)) : e / (TR / 2);
                double2 v = make_double2(0.0, 0.0);
                if (i < nr && j < n) v = *reinterpret_cast<const double2*>(a.A + (r0 + i) * a.ars + j * a.acs);
                xv[2 * u] = v.x; xv[2 * u + 1] = v.y;
            }
        } else {
#pragma unroll
            for (int u = 0; u < EPT; ++u) {
                const int e = tid + 256 * u;
                const int i = colfast ? e / N : e % TR, j = colfast ? e % N : e / TR;
                xv[u] = (i < nr && j < n) ? a.A[(r0 + i) * a.ars + j * a.acs] : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            const double v = fabs(xv[u]);
            amax = (v == v) ? fmax(amax, v) : 1.7e308;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmax(amax, __shfl_xor(amax, o, 64));
        if (lane == 0) red[wave] = amax;
        __syncthreads();
        amax = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        if (amax > 0.0 && amax < 1.7e308) frexp(amax, &ex);
        const double scl = (ex > -2000) ? ldexp(1.0, -ex) : 0.0;
        if (wide) {
#pragma unroll
            for (int u = 0; u < EPT / 2; ++u) {
                const int e = tid + 256 * u;
                const int i = colfast ? e / (N / 2) : 2 * (e % (TR / 2)), j = colfast ? 2 * (e % (N / 2)) : e / (TR / 2);
                T[i * P + j] = xv[2 * u] * scl;
                T[(colfast ? i : i + 1) * P + (colfast ? j + 1 : j)] = xv[2 * u + 1] * scl;
            }
        } else {
#pragma unroll
            for (int u = 0; u < EPT; ++u) {
                const int e = tid + 256 * u;
                const int i = colfast ? e / N : e % TR, j = colfast ? e % N : e / TR;
                T[i * P + j] = xv[u] * scl;
            }
        }
        __syncthreads();
    }
    // Gram matrix of the tile on the matrix cores: the unique 16 x 16 tiles are dealt to the waves (tile t -> wave t mod 4), every wave
    // runs over all TR rows for its tiles.  dst == nullptr: into Gs (single workgroup), otherwise the published partial (full matrix).
    auto block_gram = [&](double* dst) {
#pragma unroll
        for (int t0 = 0; t0 < NTILE; t0 += 4) {
            const int t = t0 + wave;
            if (t < NTILE) {                       // wave-uniform
                int ti = 0, rem = t;
                while (rem >= NT - ti) { rem -= NT - ti; ++ti; }
                const int tj = ti + rem;
                d4s acc = d4s{0.0, 0.0, 0.0, 0.0};
#pragma unroll 8
                for (int ks = 0; ks < TR / 4; ++ks) {
                    const int row = ks * 4 + lk;
                    const double fa = T[row * P + ti * 16 + li], fb = T[row * P + tj * 16 + li];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb, acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gi = ti * 16 + lk + 4 * r, gj = tj * 16 + li;
                    if (dst) { sq_st(dst + gi * N + gj, acc[r]); if (ti != tj) sq_st(dst + gj * N + gi, acc[r]); }
                    else { Gs[gi * P + gj] = acc[r]; if (ti != tj) Gs[gj * P + gi] = acc[r]; }
                }
            }
        }
    };
    // the summed Gram matrix of pass `par` into Gs: slice w of all partials by workgroup w (block order, optional power-of-two
    // weights), one more barrier, then everybody fetches the n x n sum.  Returns false when a barrier gave up.
    auto gather_gram = [&](int par, bool weights, int& emax_out) -> bool {
        const double* part = a.part + (int64_t)par * nblk * N * N;
        double* gsum = a.gsum + (int64_t)par * N * N;
        int emax = 0;
        if (weights) {
            int e = -100000;
            for (int i = lane; i < nblk; i += 64) { const int x = sq_ldi(a.bexp + i); e = x > e ? x : e; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const int y = __shfl_xor(e, o, 64); e = y > e ? y : e; }
            emax = e;
        }
        emax_out = emax;
        const int sl = (N * N + nblk - 1) / nblk;              // slice length
        for (int e0 = tid; e0 < sl; e0 += 256) {
            const int e = blk * sl + e0;
            if (e < N * N) {
                double acc = 0.0;
                for (int b0 = 0; b0 < nblk; b0 += 8) {
                    double v[8];
                    int xe[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const bool in = b0 + u < nblk;
                        v[u] = in ? sq_ld(part + (int64_t)(b0 + u) * N * N + e) : 0.0;
                        xe[u] = (weights && in) ? sq_ldi(a.bexp + b0 + u) : emax;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc = fma(weights ? ldexp(1.0, 2 * (xe[u] - emax)) : 1.0, v[u], acc);
                }
                sq_st(gsum + e, acc);
            }
        }
        if (!sq_grid_barrier(&a.stt->counter, (++nbar) * nblk, &s_flag, tid, a.spin_limit)) return false;
#pragma unroll
        for (int u = 0; u < N * N / 256; ++u) {
            const int e = tid + 256 * u;
            Gs[(e / N) * P + (e % N)] = sq_ld(gsum + e);
        }
        __syncthreads();
        return true;
    };
    // decision + Cholesky with deferral of the Gram matrix in Gs.  The distance from the identity is measured by all threads; the
    // factorisation runs in wave 0 in blocks of 32 rows (the scheme of cq_tail_fused): 32 right-looking steps on the block row
    // [R11 R12] with lane = column (all N columns at once: 32 registers per lane), then -- N = 64 -- the Schur complement
    // S = G22 - R12^T R12 on the matrix cores and 32 steps on S.  A deferred pivot leaves row j = e_j (R12's row j = 0 with it).
    auto factor = [&](int pass, int emax) {
        double dev = 0.0;
#pragma unroll
        for (int u = 0; u < N * N / 256; ++u) {
            const int e = tid + 256 * u, i = e / N, k = e % N;
            const double gx = (i < n && k < n) ? Gs[i * P + k] : ((i == k) ? 1.0 : 0.0);
            const double x = fabs(gx - (i == k ? 1.0 : 0.0));
            dev = (x == x) ? fmax(dev, x) : 1e300;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dev = fmax(dev, __shfl_xor(dev, o, 64));
        if (lane == 0) red[4 + wave] = dev;
        __syncthreads();
        dev = fmax(fmax(red[4], red[5]), fmax(red[6], red[7]));
        int dec = 0;
        if (pass > 0 && dev <= SQ_DONE) dec = 1;
        else if (pass >= a.maxpass) dec = 2;
        if (tid == 0) {
            s_out[0] = dec; s_out[4] = emax;
            if (dec != 0) { s_out[1] = 0; s_out[2] = 0; s_out[3] = 0; }
        }
        if (dec == 0 && tid < 64) {
            const int k = lane & (N - 1);
            // squared norm of column k before any reduction: zero (underflowing, non-finite) columns are refilled with noise in the
            // next pass, a pivot below SQ_THETA of it is deferred
            const double gd = (k < n) ? Gs[k * P + k] : 1.0;
            const bool zero_k = !(gd > 1e-290) || !(gd < 1e300);
            unsigned long long deadmask = __ballot(zero_k);
            if (N == 32) deadmask &= 0xffffffffull;
            const double thr_k = zero_k ? 1e308 : SQ_THETA * gd;
            unsigned long long badmask = 0ull;
            // 32 steps on rows row0 .. row0+31: g[i] = entry (row0 + i, column kc) of the reduced matrix, thr of column kc; writes rows
            // row0.. of R (columns >= row0) and returns this lane's diagonal entry (lanes row0 <= kc < row0 + 32)
            auto steps32 = [&](double (&g)[32], int row0, int kc, double thr_c, bool store) -> double {
                double dkk = 1.0;
                const int kk = kc - row0;                  // column relative to the block (>= 32: the R12 part)
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    const double d = sq_readlane(g[j], j), thr = sq_readlane(thr_c, j);
                    const bool ok = d > thr;
                    badmask |= ok ? 0ull : (1ull << (row0 + j));
                    const double rinv = sq_rsqrt2(ok ? d : 1.0);
                    double r = (kk >= j) ? g[j] * (ok ? rinv : 0.0) : 0.0;
                    r = (!ok && kk == j) ? 1.0 : r;
                    if (j == kk) dkk = r;
                    if (store) Rf[(row0 + j) * RP + kc] = r;
                    if (j < 31) {
                        const double m1 = sq_readlane(r, j + 1);
                        const int i0 = (j + 3) & ~1;
                        double2 mm[16];
                        double m2 = 0.0;
                        if (j + 2 < 32 && ((j + 2) & 1)) m2 = Rf[(row0 + j) * RP + row0 + j + 2];
#pragma unroll
                        for (int i = i0; i < 32; i += 2) mm[i >> 1] = *reinterpret_cast<const double2*>(&Rf[(row0 + j) * RP + row0 + i]);
                        sq_fnma(g[j + 1], m1, r);
                        if (j + 2 < 32 && ((j + 2) & 1)) sq_fnma(g[j + 2], m2, r);
#pragma unroll
                        for (int i = i0; i < 32; i += 2) {
                            sq_fnma(g[i], mm[i >> 1].x, r);
                            sq_fnma(g[i + 1], mm[i >> 1].y, r);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                return dkk;
            };
            double g[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const double x = Gs[i * P + k];
                g[i] = (i < n && k < n) ? x : ((i == k) ? 1.0 : 0.0);
            }
            double dkk = steps32(g, 0, k, thr_k, lane < N);
            if constexpr (N == 64) {
                __builtin_amdgcn_wave_barrier();
                // S = G22 - R12^T R12 into the lower right block of Gs (tiles (0,0), (0,1), (1,1) of 16 x 16; K = the 32 rows of R12)
                d4s s00 = d4s{0.0, 0.0, 0.0, 0.0}, s01 = s00, s11 = s00;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const double f0 = Rf[(ks * 4 + lk) * RP + 32 + li], f1 = Rf[(ks * 4 + lk) * RP + 48 + li];
                    s00 = __builtin_amdgcn_mfma_f64_16x16x4f64(f0, f0, s00, 0, 0, 0);
                    s01 = __builtin_amdgcn_mfma_f64_16x16x4f64(f0, f1, s01, 0, 0, 0);
                    s11 = __builtin_amdgcn_mfma_f64_16x16x4f64(f1, f1, s11, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = lk + 4 * r;
                    Gs[(32 + i) * P + 32 + li] -= s00[r];
                    Gs[(48 + i) * P + 48 + li] -= s11[r];
                    const double v = Gs[(32 + i) * P + 48 + li] - s01[r];
                    Gs[(32 + i) * P + 48 + li] = v;
                    Gs[(48 + li) * P + 32 + i] = v;
                }
                __builtin_amdgcn_wave_barrier();
                const int k2 = 32 + (lane & 31);           // both halves of the wave run the second block (the upper one idles along)
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const double x = Gs[(32 + i) * P + k2];
                    g[i] = (32 + i < n && k2 < n) ? x : ((32 + i == k2) ? 1.0 : 0.0);
                }
                const double thr2 = __shfl(thr_k, k2, 64);
                const double d2 = steps32(g, 32, k2, thr2, lane < 32);
                if (lane < 32) {
#pragma unroll
                    for (int c = 0; c < 32; ++c) Rf[(32 + lane) * RP + c] = 0.0;       // rows 32.. have nothing left of the diagonal block
                    Rf[N * RP + k2] = fast_rcp(d2);
                    Rf[N * RP + lane] = fast_rcp(dkk);
                }
            } else {
                if (lane < N) Rf[N * RP + k] = fast_rcp(dkk);
            }
            if (lane == 0) {
                s_out[1] = (pass > 0 && dev <= SQ_LAST && badmask == 0ull) ? 1 : 0;
                s_out[2] = (int)(unsigned)(deadmask & 0xffffffffull);
                s_out[3] = (int)(unsigned)(deadmask >> 32);
            }
        }
        __syncthreads();
    };
    // X <- X R_t^-1 on the rows of the tile: row tid / TPR, the 256 / TR threads of a row own column pairs (2q, 2q+1) with
    // q mod TPR = tid mod TPR; x_j travels through the DPP network, the multipliers are LDS broadcasts fetched one step ahead
    auto substitute = [&](unsigned long long deadmask, uint64_t seed, double scl) {
        constexpr int NP = XL / 2;
        const int row = tid / TPR, sub = tid % TPR;
        double x[XL];
#pragma unroll
        for (int l = 0; l < XL; ++l) x[l] = T[row * P + 2 * ((l >> 1) * TPR + sub) + (l & 1)] * scl;
        double2 mc[NP], mn[NP];
        double dc = Rf[N * RP], dn = 0.0;
#pragma unroll
        for (int p = 0; p < NP; ++p) mc[p] = *reinterpret_cast<const double2*>(&Rf[2 * (p * TPR + sub)]);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const int owner = (j >> 1) % TPR, lj = 2 * ((j >> 1) / TPR) + (j & 1);
            if (j < N - 1) {                               // prefetch step j+1
                dn = Rf[N * RP + j + 1];
#pragma unroll
                for (int p = ((j + 1) >> 1) / TPR; p < NP; ++p) mn[p] = *reinterpret_cast<const double2*>(&Rf[(j + 1) * RP + 2 * (p * TPR + sub)]);
            }
            const double xj = sq_group_bcast<TPR>(x[lj] * dc, owner);
#pragma unroll
            for (int p = (j >> 1) / TPR; p < NP; ++p) {
                sq_fnma(x[2 * p], xj, mc[p].x);
                sq_fnma(x[2 * p + 1], xj, mc[p].y);
            }
            if (sub == owner) x[lj] = xj;
            __builtin_amdgcn_sched_barrier(0);
            dc = dn;
#pragma unroll
            for (int p = 0; p < NP; ++p) mc[p] = mn[p];
        }
        if (deadmask) {
#pragma unroll
            for (int l = 0; l < XL; ++l) {
                const int c = 2 * ((l >> 1) * TPR + sub) + (l & 1);
                if ((deadmask >> c) & 1ull) x[l] = (row < nr) ? sq_hash_unit(seed + (uint64_t)(r0 + row) * 64 + c) : 0.0;
            }
        }
#pragma unroll
        for (int l = 0; l < XL; ++l) T[row * P + 2 * ((l >> 1) * TPR + sub) + (l & 1)] = x[l];
    };
    // accumulated factor R = R_t ... R_1 in registers: wave w < NT holds block column w as NT tiles in the MFMA result layout
    // (register r of tile bi = row 16 bi + lk + 4 r, column 16 w + li), which is also the B-operand layout of k-step r
    d4s racc[NT];
    auto racc_load = [&]() {
        if (wave < NT) {
#pragma unroll
            for (int bi = 0; bi < NT; ++bi)
#pragma unroll
                for (int r = 0; r < 4; ++r) racc[bi][r] = Rf[(bi * 16 + lk + 4 * r) * RP + wave * 16 + li];
        }
    };
    auto racc_mul = [&](unsigned long long deadmask) {      // R <- R_t R, then the rows of refilled columns are zeroed
        if (wave < NT) {
            d4s nw[NT];
#pragma unroll
            for (int bi = 0; bi < NT; ++bi) {
                nw[bi] = d4s{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int bk = bi; bk < NT; ++bk)          // R_t is upper triangular: block row bi meets block columns bk >= bi
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const double fa = Rf[(bi * 16 + li) * RP + bk * 16 + ks * 4 + lk];
                        nw[bi] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, racc[bk][ks], nw[bi], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int bi = 0; bi < NT; ++bi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = bi * 16 + lk + 4 * r;
                    racc[bi][r] = ((deadmask >> i) & 1ull) ? 0.0 : nw[bi][r];
                }
        }
    };

    int dec = 0, emax = 0, passes = 0;
    SQ_CLK(0);
    // ---- pass 0: Gram matrix of the input
    if (multi) {
        block_gram(a.part + (int64_t)blk * N * N);
        if (tid == 0) sq_sti(a.bexp + blk, ex);
        alive = sq_grid_barrier(&a.stt->counter, (++nbar) * nblk, &s_flag, tid, a.spin_limit);
        if (alive) alive = gather_gram(0, true, emax);
    } else {
        block_gram(nullptr);
        emax = ex;
        __syncthreads();
    }
    SQ_CLK(1);
    if (alive) {
        factor(0, emax);
        SQ_CLK(2);
        dec = s_out[0];
        const double scl0 = (ex > -2000 && emax > -2000) ? ldexp(1.0, ex - emax) : 0.0;       // tile is 2^-ex A; the passes work on 2^-emax A
        for (int t = 1; dec == 0; ++t) {
            const int fin = s_out[1];
            const unsigned long long deadmask = (unsigned long long)(unsigned)s_out[2] | ((unsigned long long)(unsigned)s_out[3] << 32);
            __syncthreads();
            if (t == 1) SQ_CLK(3);
            substitute(deadmask, a.seed + 0x9E3779B97F4A7C15ULL * (uint64_t)t, t == 1 ? scl0 : 1.0);
            if (t == 1) SQ_CLK(4);
            if (t == 1) racc_load(); else racc_mul(0ull);
            if (deadmask && wave < NT) {
#pragma unroll
                for (int bi = 0; bi < NT; ++bi)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if ((deadmask >> (bi * 16 + lk + 4 * r)) & 1ull) racc[bi][r] = 0.0;
            }
            __syncthreads();
            if (t == 1) SQ_CLK(5);
            passes = t;
            if (fin) { dec = 1; break; }
            if (multi) {
                block_gram(a.part + ((int64_t)(t & 1) * nblk + blk) * N * N);
                alive = sq_grid_barrier(&a.stt->counter, (++nbar) * nblk, &s_flag, tid, a.spin_limit);
                int dummy;
                if (alive) alive = gather_gram(t & 1, false, dummy);
                if (!alive) break;
            } else {
                block_gram(nullptr);
                __syncthreads();
            }
            if (t == 1) SQ_CLK(6);
            factor(t, emax);
            if (t == 1) SQ_CLK(7);
            dec = s_out[0];
        }
    }
    SQ_CLK(8);
    if (alive && dec == 1) {
        // ---- Q = the tile
        const bool ofast = (a.qcs == 1);
        if (wide_ok(a.Q, a.qrs, a.qcs)) {
#pragma unroll
            for (int u0 = 0; u0 < EPT / 2; u0 += 8) {
                double2 ov[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = tid + 256 * (u0 + u);
                    const int i = ofast ? e / (N / 2) : 2 * (e % (TR / 2)), j = ofast ? 2 * (e % (N / 2)) : e / (TR / 2);
                    ov[u] = make_double2(T[i * P + j], T[(ofast ? i : i + 1) * P + (ofast ? j + 1 : j)]);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = tid + 256 * (u0 + u);
                    const int i = ofast ? e / (N / 2) : 2 * (e % (TR / 2)), j = ofast ? 2 * (e % (N / 2)) : e / (TR / 2);
                    if (i < nr && j < n) *reinterpret_cast<double2*>(a.Q + (r0 + i) * a.qrs + j * a.qcs) = ov[u];
                }
            }
        } else {
#pragma unroll
            for (int u0 = 0; u0 < EPT; u0 += 8) {
                double ov[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = tid + 256 * (u0 + u);
                    const int i = ofast ? e / N : e % TR, j = ofast ? e % N : e / TR;
                    ov[u] = T[i * P + j];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = tid + 256 * (u0 + u);
                    const int i = ofast ? e / N : e % TR, j = ofast ? e % N : e / TR;
                    if (i < nr && j < n) a.Q[(r0 + i) * a.qrs + j * a.qcs] = ov[u];
                }
            }
        }
        // ---- R = 2^emax R_acc (workgroup 0), optionally divided by its power-of-two norm factor
        if (blk == 0) {
            const double back = (emax > -2000) ? ldexp(1.0, emax) : 0.0;
            unsigned long long mx = 0ull;
            if (wave < NT) {
#pragma unroll
                for (int bi = 0; bi < NT; ++bi)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = bi * 16 + lk + 4 * r, j = wave * 16 + li;
                        const double v = (i < n && j < n && j >= i) ? racc[bi][r] * back : 0.0;
                        racc[bi][r] = v;
                        const unsigned long long b = (unsigned long long)__double_as_longlong(fabs(v));
                        mx = b > mx ? b : mx;
                    }
            }
            double inv = 1.0;
            if (a.nf_out2) {
                __shared__ unsigned long long mred[4];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { const unsigned long long t2 = __shfl_xor(mx, o, 64); mx = t2 > mx ? t2 : mx; }
                if (lane == 0) mred[wave] = mx;
                __syncthreads();
                unsigned long long m0 = mred[0] > mred[1] ? mred[0] : mred[1], m1 = mred[2] > mred[3] ? mred[2] : mred[3];
                m0 = m0 > m1 ? m0 : m1;
                const double f = ldexp(1.0, (int)((long long)(m0 >> 52) - 1023));
                inv = (m0 == 0ull) ? 1.0 : 1.0 / f;            // an all-zero factor stays zero whatever the denormal mode (f = 2^-1023 as mps.py:76-85)
                if (tid == 0) { a.nf_out2[0] = f; a.nf_out2[1] = inv; }
            }
            if (wave < NT) {
#pragma unroll
                for (int bi = 0; bi < NT; ++bi)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = bi * 16 + lk + 4 * r, j = wave * 16 + li;
                        if (i < n && j < n) a.R[(int64_t)i * a.rrs + (int64_t)j * a.rcs] = racc[bi][r] * inv;
                    }
            }
            if (tid == 0) { atomicAdd(&a.stats[0], 1ull); atomicAdd(&a.stats[1], (unsigned long long)passes); }
        }
    } else if (alive && dec == 2) {
        // out of passes: workgroup 0 redoes the factorisation with Householder reflections from the untouched input
        if (blk == 0) {
            __syncthreads();
            sq_fallback_householder(a.A, a.ars, a.acs, a.Q, a.qrs, a.qcs, a.R, a.rrs, a.rcs, a.m, n, emax, T, tid);
            if (a.nf_out2) {                                   // R /= nfactor(R), from global memory
                __threadfence();
                __syncthreads();
                unsigned long long mx = 0ull;
                for (int e = tid; e < n * n; e += 256) {
                    const unsigned long long b = (unsigned long long)__double_as_longlong(fabs(sq_ld(a.R + (int64_t)(e / n) * a.rrs + (int64_t)(e % n) * a.rcs)));
                    mx = b > mx ? b : mx;
                }
                __shared__ unsigned long long fred[4];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { const unsigned long long t2 = __shfl_xor(mx, o, 64); mx = t2 > mx ? t2 : mx; }
                if (lane == 0) fred[wave] = mx;
                __syncthreads();
                unsigned long long m0 = fred[0] > fred[1] ? fred[0] : fred[1], m1 = fred[2] > fred[3] ? fred[2] : fred[3];
                m0 = m0 > m1 ? m0 : m1;
                const double f = ldexp(1.0, (int)((long long)(m0 >> 52) - 1023)), inv = (m0 == 0ull) ? 1.0 : 1.0 / f;
                if (tid == 0) { a.nf_out2[0] = f; a.nf_out2[1] = inv; }
                for (int e = tid; e < n * n; e += 256) {
                    double* p = a.R + (int64_t)(e / n) * a.rrs + (int64_t)(e % n) * a.rcs;
                    *p = sq_ld(p) * inv;
                }
            }
            if (tid == 0) { atomicAdd(&a.stats[0], 1ull); atomicAdd(&a.stats[1], (unsigned long long)passes); atomicAdd(&a.stats[2], 1ull); }
        }
    } else {
        // a barrier gave up: poison the outputs, book the time-out (the host checks the counter: cholqr_timeouts)
        const double bad = __longlong_as_double(0x7ff8000000000000LL);
        for (int e = tid; e < nr * n; e += 256) a.Q[(r0 + e / n) * a.qrs + (e % n) * a.qcs] = bad;
        if (blk == 0) {
            for (int e = tid; e < n * n; e += 256) a.R[(int64_t)(e / n) * a.rrs + (int64_t)(e % n) * a.rcs] = bad;
            if (tid == 0 && a.nf_out2) { a.nf_out2[0] = bad; a.nf_out2[1] = bad; }
        }
        if (tid == 0) atomicAdd(&a.stats[3], 1ull);
    }
    SQ_CLK(9);
    // the last workgroup to leave clears the barrier counter for the stream's next launch (a launch that gave up leaves it to the
    // host: the stream is taken off the single-launch forms and its state is cleared, see cholqr_timeouts)
    if (multi && tid == 0) {
        __threadfence();
        const int prev = atomicAdd(&a.stt->exits, 1);
        if (prev == nblk - 1) { a.stt->exits = 0; a.stt->counter = 0; __threadfence(); }
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------------

static SqState* sq_state_of(int slot) {
    static std::mutex mu;
    static char* base[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!base[dev]) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(sq_state_pool)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        base[dev] = (char*)p;
    }
    return (SqState*)base[dev] + slot;
}
static unsigned long long* sq_stats_of(int slot) {
    static std::mutex mu;
    static char* base[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!base[dev]) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(sq_stats)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        base[dev] = (char*)p;
    }
    return (unsigned long long*)base[dev] + 4 * slot;
}

int64_t smallqr_ws_bytes(int64_t m, int64_t n) {
    const int N = n <= 32 ? 32 : 64;
    return (int64_t)(2 * SQ_MAXBLK + 2) * N * N * 8 + 1024;
}

// whether smallqr_factor takes this shape (the caller then needs smallqr_ws_bytes of scratch)
bool smallqr_fits(int64_t m, int64_t n) {
    if (n < 1 || n > 64 || m < n) return false;
    const int TR = n <= 32 ? 256 : 128;
    return cdiv(m, TR) <= SQ_MAXBLK;
}

// A (m x n, element strides; read only) = Q (m x n) R (n x n upper triangular, diag >= 0).  nf_out2 != NULL: R is divided by its
// power-of-two norm factor, [nf, 1/nf] stored there.  Returns 0 when done, 1 when the shape / the stream is not taken (the caller
// uses the blocked path), an error code otherwise.
int smallqr_factor(hipStream_t st, const double* A, int64_t rs, int64_t cs, int64_t m, int64_t n, double* Q, int64_t qrs, int64_t qcs, double* R,
                   int64_t rrs, int64_t rcs, double* nf_out2, void* ws, int64_t ws_bytes) {
    {
        const char* e = getenv("TN_QR_SMALL");                        // read per call: the tests switch it
        if (e && e[0] == '0') return 1;
    }
    if (!smallqr_fits(m, n) || !Q || !R || ws_bytes < smallqr_ws_bytes(m, n)) return 1;
    const int slot = cholqr_stream_slot(st);
    if (slot >= CHOLQR_SLOTS) return 1;
    const int N = n <= 32 ? 32 : 64, TR = n <= 32 ? 256 : 128;
    const int nblk = (int)cdiv(m, TR);
    if (nblk > 1 && !fused_forms_allowed(st, nblk)) return 1;
    SqArgs a;
    a.A = A; a.ars = rs; a.acs = cs; a.m = m; a.n = (int)n;
    a.Q = Q; a.qrs = qrs; a.qcs = qcs; a.R = R; a.rrs = rrs; a.rcs = rcs; a.nf_out2 = nf_out2;
    char* p = (char*)ws;
    a.part = (double*)p; p += (int64_t)2 * SQ_MAXBLK * N * N * 8;
    a.gsum = (double*)p; p += (int64_t)2 * N * N * 8;
    a.bexp = (int*)p;
    a.stt = sq_state_of(slot);
    a.stats = sq_stats_of(slot);
    if (!a.stt || !a.stats) return 1;
    a.seed = 0x5bd1e995u + 1315423911ull * (uint64_t)(m * 131 + n);
    a.nblk = nblk;
    static const int maxpass = [] { const char* e = getenv("TN_PANEL_MAXPASS"); const int v = e ? atoi(e) : SQ_MAXPASS; return v >= 1 && v <= SQ_MAXPASS ? v : SQ_MAXPASS; }();
    a.maxpass = maxpass;
    {
        const char* e = getenv("TN_PANEL_SPIN_LIMIT");                 // tests: force the barriers to give up
        a.spin_limit = e ? (unsigned)strtoul(e, nullptr, 10) : (1u << 22);
    }
    if (nblk > 1) fused_note_launch();
    prof_begin(st, PROF_TSQR);
    if (N == 32) hipLaunchKernelGGL((sq_kernel<32, 256>), dim3(nblk), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((sq_kernel<64, 128>), dim3(nblk), dim3(256), 0, st, a);
    TN_CHECK_LAUNCH("sq_kernel");
    const double e = (double)m * (double)n;
    prof_end(st, PROF_TSQR, 2.0 * e * n, 16.0 * e + 8.0 * n * n);
    return 0;
}

// statistics of the stream's single-launch factorisations: out4 = calls, passes, Householder fallbacks, launches that gave up
int smallqr_stats(hipStream_t st, unsigned long long* out4, int reset) {
    const int slot = cholqr_stream_slot(st);
    for (int i = 0; i < 4; ++i) out4[i] = 0;
    if (slot >= CHOLQR_SLOTS) return 0;
    hipError_t e = hipMemcpyFromSymbol(out4, HIP_SYMBOL(sq_stats), 32, (size_t)slot * 32);
    if (e != hipSuccess) return hip_fail(e, "read small-QR statistics");
    if (reset) {             // (word [3], the sticky count of launches that gave up, stays: fused_timeouts compares it with the value it saw last)
        unsigned long long z[4] = {0, 0, 0, out4[3]};
        if ((e = hipMemcpyToSymbol(HIP_SYMBOL(sq_stats), z, 32, (size_t)slot * 32)) != hipSuccess) return hip_fail(e, "reset small-QR statistics");
    }
    return 0;
}
// clears the barrier state of a stream (after a launch gave up)
int smallqr_reset_state(hipStream_t st) {
    const int slot = cholqr_stream_slot(st);
    if (slot >= CHOLQR_SLOTS) return 0;
    SqState* s = sq_state_of(slot);
    if (!s) return 0;
    const hipError_t e = hipMemsetAsync(s, 0, sizeof(SqState), st);
    return e == hipSuccess ? 0 : hip_fail(e, "clear small-QR state");
}

}  // namespace tn
