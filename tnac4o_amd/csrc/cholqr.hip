// Panel orthonormalisation for the blocked QR (K3), second generation: iterated Cholesky-QR with deferral.
//
// tsqr.hip orthonormalises a tall nrows x b panel (b <= 32) with Householder TSQR: unconditionally stable, but every tree
// level is a chain of 32 column steps of ~1.3 us in a single workgroup (3 levels x (41 + 22) us for 16384 rows).  Here
// the serial part shrinks to the Cholesky factorisation of ONE 32 x 32 Gram matrix per pass, done by one wave:
//
//   pass 0 (cq_gram_kernel):   G = X^T X    (every workgroup: its 256 rows on the matrix cores; the last workgroup to finish
//                               sums the per-block partials in block order -- bit-reproducible -- and factors G = R^T R)
//   pass t (cq_pass_kernel):   X <- X R^-1  (row-parallel substitution), G = X^T X of the new panel, the last workgroup
//                               decides: converged / one more pass / factor again.
//
// In exact arithmetic one pass suffices; in floating point the result is orthonormal to eps kappa(X)^2, so passes repeat
// until the Gram matrix of the current panel is the identity to rounding (2 passes for kappa < 1e4, 3 otherwise).  What
// makes it safe for the rank-deficient, graded panels of this path:
//   * Cholesky is invariant under column scaling, so only the angles between columns matter;
//   * deferral: a column whose pivot d_j falls below CQ_THETA of its squared norm (it lies in the span of the columns to its
//     left to within 1e-5 of its norm, so d_j has lost most of its digits to cancellation) is not normalised by that pass
//     and not used to reduce later columns (row j of R = e_j).  The substitution then leaves its exact residual, computed
//     in vector arithmetic; the next pass sees that residual with its true norm and treats it as an ordinary column;
//   * a column that is exactly zero (or whose square underflows) is refilled with hash noise: an arbitrary completion,
//     exactly what a tau = 0 Householder reflector stands for;
//   * the panel only has to be an orthonormal basis of span(X): the blocked QR takes R from H^T A (qr.hip), never from here.
// If the panel is still not orthonormal after CQ_MAXPASS passes, the last workgroup of the last launch redoes it from the
// untouched input with a plain (slow, single-workgroup) Householder QR: unconditional like tsqr.hip, never seen on the
// contraction path so far (tn_debug_panel_stats counts it).
//
// Launches are plain stream-ordered kernels: nothing spins, nothing needs co-residency, so interleaved chains cannot
// deadlock.  A launch that finds the panel converged returns at once (~3 us).
#include <stdlib.h>

#include <map>
#include <mutex>
#include <vector>

#include "common.h"

// -DTN_CLOCKS: thread 0 of the last workgroup of a launch records the 100 MHz wall clock at phase boundaries (diagnostics only)
#ifdef TN_CLOCKS
__device__ long long cq_clk[32];
extern "C" int tn_debug_clocks2(long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(cq_clk), sizeof(long long) * (n < 32 ? n : 32));
}
#define CQ_CLK_DECL long long clk_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define CQ_CLK(k) do { if (threadIdx.x == 0) clk_[k] = wall_clock64(); } while (0)
#define CQ_CLK_DUMP(base) do { if (threadIdx.x == 0) for (int q_ = 0; q_ < 12; ++q_) cq_clk[(base) + q_] = clk_[q_]; } while (0)
#define FQ_CLK(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) cq_clk[k] = wall_clock64(); } while (0)
#else
#define FQ_CLK(k) do {} while (0)
#define CQ_CLK_DECL do {} while (0)
#define CQ_CLK(k) do {} while (0)
#define CQ_CLK_DUMP(base) do {} while (0)
#endif

namespace tn {

typedef double d4c __attribute__((ext_vector_type(4)));

constexpr int CQ_RB = 256;            // rows per workgroup
constexpr int CQ_P = 33;              // LDS pitch of the row tile
constexpr int CQ_PART = 768;          // per-block partial Gram: tiles (0,0), (0,1), (1,1) of 16 x 16
constexpr int CQ_MAXPASS = 4;         // substitution passes enqueued per panel (later ones return at once when converged)
constexpr double CQ_THETA = 1e-10;    // deferral threshold on pivot / squared column norm
constexpr double CQ_DONE = 5e-15;     // Gram matrix = identity to rounding: converged
constexpr double CQ_LAST = 1e-8;      // below this one more pass lands at rounding level without another check

struct CqState {
    int counter;          // arrival ticket of the workgroups of the launch in flight
    int done;             // panel orthonormal: remaining launches return at once
    int final_next;       // the next pass is the last one (no Gram / check after it)
    int pass;             // passes applied so far
    int emax;             // power-of-two exponent of the panel's largest entry (pass 0 scaling)
    unsigned dead;        // columns to refill with noise in the next pass
    int ndefer_total;     // statistics: deferred pivots, refills, Householder fallbacks of this panel
    int nrefill_total;
    int fallback;
    int fcounter;         // arrivals at the in-kernel barriers of the single-launch form (monotone over the panels of a call)
    int timeout;          // sticky: a barrier of the single-launch form gave up (outputs poisoned with NaN)
    double dev_hist[CQ_MAXPASS + 2];
};
constexpr int CQ_STATE_BYTES = 256;
static_assert(sizeof(CqState) <= CQ_STATE_BYTES, "state block too small");

__device__ __forceinline__ double cq_readlane(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double cq_rsqrt2(double x) {      // hardware seed (~2^-26) + two Newton steps
    double r = __builtin_amdgcn_rsq(x);
    r = r * (1.5 - 0.5 * x * r * r);
    r = r * (1.5 - 0.5 * x * r * r);
    return r;
}
// acc -= a * b, pinned in program order: left to itself the compiler sinks the rank-1 updates of the unrolled factorisation
// into 31-long dependent chains at the point of use and spills the multipliers it keeps alive for them
__device__ __forceinline__ void cq_fnma(double& acc, double a, double b) {
    asm volatile("v_fma_f64 %0, -%1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ double cq_hash_unit(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return ((double)(x >> 11) * (1.0 / 9007199254740992.0)) - 0.5;
}
__device__ __forceinline__ void cq_block_rows(int64_t nrows, int nblk, int blk, int64_t& r0, int& nr) {
    const int64_t base = nrows / nblk, rem = nrows % nblk;
    r0 = blk * base + (blk < rem ? blk : rem);
    nr = (int)(base + (blk < rem ? 1 : 0));
}

__device__ __forceinline__ double cq_ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// What the workgroups publish to the last one (partial Gram matrices, block exponents) is written with agent-scope stores:
// they go through to memory, so the publisher only waits for their completion before it takes its ticket -- a release
// fence would also write back every dirty line of the XCD's L2 (the tile just stored: 3.6 us measured against ~1).
__device__ __forceinline__ void cq_st(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void cq_sti(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void cq_publish_wait() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_waitcnt(0); }
__device__ __forceinline__ int cq_ldi(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Partial Gram of the 256 x 32 LDS tile T (pitch CQ_P): wave w covers rows 64w .. 64w+63 on the matrix cores, the four
// partials meet in the LDS scratch S (4 x 768 doubles; callers that are done with the tile pass T itself) and their sum goes
// to part[0 .. 767].
__device__ __forceinline__ void cq_block_gram(const double* T, double* S, double* __restrict__ part, int tid) {
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    d4c g00 = d4c{0, 0, 0, 0}, g01 = g00, g11 = g00;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const int row = wave * 64 + ks * 4 + lk;
        const double f0 = T[row * CQ_P + li], f1 = T[row * CQ_P + 16 + li];
        g00 = __builtin_amdgcn_mfma_f64_16x16x4f64(f0, f0, g00, 0, 0, 0);
        g01 = __builtin_amdgcn_mfma_f64_16x16x4f64(f0, f1, g01, 0, 0, 0);
        g11 = __builtin_amdgcn_mfma_f64_16x16x4f64(f1, f1, g11, 0, 0, 0);
    }
    __syncthreads();                                     // every wave has read its rows of T
    double* sp = S + wave * CQ_PART;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = lk + 4 * r;
        sp[i * 16 + li] = g00[r];
        sp[256 + i * 16 + li] = g01[r];
        sp[512 + i * 16 + li] = g11[r];
    }
    __syncthreads();
    for (int e = tid; e < CQ_PART; e += 256) cq_st(part + e, (S[e] + S[CQ_PART + e]) + (S[2 * CQ_PART + e] + S[3 * CQ_PART + e]));
}

// diagnostic counters (tn_panel_stats / tn_panel_stats_stream), kept PER STREAM -- slot of the launching stream, CQ_STAT_SLOTS for the
// streams beyond that many; concurrent chains do not mix their counts -- one 64-bit word each, no packed sub-fields that could carry
// into each other:
//   [0] panels  [1] substitution passes applied  [2] deferred pivots  [3] refilled columns  [4] Householder fallbacks
//   [5] panels with >= 3 passes  [6] panels with >= 4 passes  [7] panel elements x passes applied by the six-launch chain (each such
//   pass reads and writes the panel: 16 bytes per element)  [8] the same for the single-launch form (the tile stays in LDS: flops only)
//   [9] panels handled by the single-launch form  [10] single-launch panels that gave up at an in-kernel barrier (time-outs)
constexpr int CQ_STAT_SLOTS = 64;
__device__ unsigned long long cq_stats[(CQ_STAT_SLOTS + 1) * 16];
// one thread, once per panel; the adds do not return a value, so the wave does not wait for them
__device__ __forceinline__ void cq_count(int slot, int passes, int ndefer, int nrefill, bool fallback, long long elems, bool single_launch = false) {
    unsigned long long* cs = cq_stats + slot * 16;
    atomicAdd(&cs[0], 1ull);
    atomicAdd(&cs[1], (unsigned long long)passes);
    atomicAdd(&cs[single_launch ? 8 : 7], (unsigned long long)passes * (unsigned long long)elems);
    if (single_launch) atomicAdd(&cs[9], 1ull);
    if (ndefer > 0) atomicAdd(&cs[2], (unsigned long long)ndefer);
    if (nrefill > 0) atomicAdd(&cs[3], (unsigned long long)nrefill);
    if (fallback) atomicAdd(&cs[4], 1ull);
    if (passes >= 3) atomicAdd(&cs[5], 1ull);
    if (passes >= 4) atomicAdd(&cs[6], 1ull);
}

// 256-thread sum through LDS (two barriers); red: >= 4 doubles
__device__ __forceinline__ double cq_block_sum(double v, double* red, int tid) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// Last resort, one workgroup: Householder QR of the whole panel in global memory (dgeqr2 + dorg2r, column by column).
// Y <- 2^-emax X, reflectors in place, then the explicit Q in place.  Slow by design (one CU streams the panel ~100 times).
__device__ void cq_fallback_householder(const double* X, int64_t xrs, int64_t xcs, double* Y, int64_t rs, int64_t cs, int64_t nrows, int b,
                                        int emax, double* lds, int tid) {
    double* red = lds;              // 4
    double* wv = lds + 8;           // 32 dot products
    double* taus = lds + 48;        // 32
    const double scl = (emax > -2000) ? ldexp(1.0, -emax) : 0.0;
    for (int64_t e = tid; e < nrows * b; e += 256) {
        const int64_t i = e / b, j = e % b;
        Y[i * rs + j * cs] = X[i * xrs + j * xcs] * scl;
    }
    __threadfence();
    __syncthreads();
    for (int j = 0; j < b; ++j) {
        double s = 0.0;
        for (int64_t r = j + 1 + tid; r < nrows; r += 256) { const double y = Y[r * rs + j * cs]; s += y * y; }
        s = cq_block_sum(s, red, tid);
        const double alpha = Y[j * rs + j * cs];
        double tau = 0.0;
        if (s > 1e-300) {                                   // dlarfg: nothing below the diagonal -> H = I
            const double beta = -copysign(sqrt(alpha * alpha + s), alpha);
            tau = (beta - alpha) / beta;
            const double inv = 1.0 / (alpha - beta);
            for (int64_t r = j + 1 + tid; r < nrows; r += 256) Y[r * rs + j * cs] *= inv;
            __threadfence();
            __syncthreads();
            for (int c = j + 1; c < b; ++c) {
                double w = 0.0;
                for (int64_t r = j + 1 + tid; r < nrows; r += 256) w += Y[r * rs + j * cs] * Y[r * rs + c * cs];
                w = cq_block_sum(w, red, tid);
                if (tid == 0) wv[c] = tau * (w + Y[j * rs + c * cs]);
                __syncthreads();
                const double tw = wv[c];
                for (int64_t r = j + 1 + tid; r < nrows; r += 256) Y[r * rs + c * cs] -= Y[r * rs + j * cs] * tw;
                if (tid == 0) Y[j * rs + c * cs] -= tw;
            }
        }
        if (tid == 0) taus[j] = tau;
        __threadfence();
        __syncthreads();
    }
    // dorg2r: Q = H_0 ... H_{b-1} [I; 0] in place
    for (int j = b - 1; j >= 0; --j) {
        const double tau = taus[j];
        for (int c = j + 1; c < b; ++c) {                  // apply H_j to the columns already formed
            double w = 0.0;
            for (int64_t r = j + 1 + tid; r < nrows; r += 256) w += Y[r * rs + j * cs] * Y[r * rs + c * cs];
            w = cq_block_sum(w, red, tid);
            // row j of the columns formed so far is zero (they live in rows > j), so v^T q = w
            const double tw = tau * w;
            for (int64_t r = j + 1 + tid; r < nrows; r += 256) Y[r * rs + c * cs] -= Y[r * rs + j * cs] * tw;
            if (tid == 0) Y[j * rs + c * cs] = -tw;
            __threadfence();
            __syncthreads();
        }
        for (int64_t r = j + 1 + tid; r < nrows; r += 256) Y[r * rs + j * cs] *= -tau;
        if (tid == 0) Y[j * rs + j * cs] = 1.0 - tau;
        for (int64_t r = tid; r < j; r += 256) Y[r * rs + j * cs] = 0.0;
        __threadfence();
        __syncthreads();
    }
}

// acc -= s * v with the first factor wave-uniform (an SGPR pair), pinned in program order like cq_fnma
__device__ __forceinline__ void cq_fnma_s(double& acc, double s_uniform, double v) {
    asm volatile("v_fma_f64 %0, -%1, %2, %0" : "+v"(acc) : "s"(s_uniform), "v"(v));
}

// Slots of the reconstruction buffer `lu` (7 x 1024 doubles, 32 x 32 row-major each): the top block of the orthonormal panel as
// published by workgroup 0, then what the Householder reconstruction leaves for the post-processing launch.
enum { CQ_LU_YTOP = 0, CQ_LU_Y1 = 1024, CQ_LU_UINV = 2048, CQ_LU_UT = 3072, CQ_LU_UTQ = 4096, CQ_LU_WTOP = 5120, CQ_LU_WQTOP = 6144, CQ_LU_DOUBLES = 7168 };

// Householder reconstruction of the panel (Ballard et al. 2014; same mathematics as lu_reconstruct_kernel in qr.hip) by one
// workgroup:  Q1_top - S = L U  with the sign choice s_i = -sign(u_ii) (|pivot| >= 1, no pivoting needed), then
//   Y1 = L,  T = -U S L^-T,  Uinv = U^-1,  UT = Uinv T^T,  UTq = Uinv T,  Wtop = L T^T,  Wqtop = L T
// so that rows below the top block follow as  Y = Q1 Uinv,  W = Y T^T = Q1 UT,  Wq = Y T = Q1 UTq.
// The elimination runs in wave 0 without barriers: lane r holds row r, the pivot row reaches the other lanes through
// cross-lane reads (SGPR operands of the updates).  scr: LDS, >= 4 * 32 * 33 + 32 doubles.  Tp: b x b, pitch b.
// Ss != nullptr (single-launch form): Uinv, UT (and UTq) go to that LDS array (3 x 1024, what cq_post_tile multiplies by) instead of the
// global buffer, and only a workgroup with write_top (the one that owns the top rows of the panel) stores the top blocks Y1 / Wtop
// (/ Wqtop) to `lu`: the reconstruction is redundant in every workgroup, its 57 KB need not travel to memory and back 64 times.
// want_q == false: Y T is not wanted (UTq, Wqtop are skipped).
__device__ __forceinline__ void cq_lu(const double* ytop, bool coherent_loads, int b, double* lu, double* Tp, double* scr, int tid,
                                      double* Ss = nullptr, bool write_top = true, bool want_q = true) {
    constexpr int P = 33;
    double* Um = scr;                  // U (zeros below the diagonal), then U S, then T
    double* Lm = scr + 32 * P;         // L (unit lower, zeros above)
    double* Ui = scr + 2 * 32 * P;     // U^-1  (before that: the matrix itself, consumed block by block)
    double* Li = scr + 3 * 32 * P;     // L^-1
    double* sg = scr + 4 * 32 * P;
    double* Bm = Ui;
    const int lane = tid & 63, wave = tid >> 6;
    const int li_ = lane & 15, lk = lane >> 4;
    // The factorisation is blocked 2 x 2 in 16 x 16 blocks, so that only the two diagonal blocks go through the one-wave elimination
    // with its sequential pivots (2 x 16 steps over at most 15 columns instead of 32 over at most 31) and the 16-step substitutions
    // for their inverses; everything off the diagonal -- U12 = L11^-1 B12, the Schur complement B22 - L21 U12, and the off-diagonal
    // blocks of both inverses -- is 16 x 16 x 16 products on the matrix cores.  A product's result tile (D layout) is the B operand
    // of the next product's K steps as it stands, so two chained products need no trip through LDS.
    {   // the matrix, padded with an identity block that is never eliminated: one coalesced round trip
        double v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { const int e = tid + 256 * t; v[t] = coherent_loads ? cq_ld(ytop + e) : ytop[e]; }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int e = tid + 256 * t, r = e >> 5, c = e & 31;
            Bm[r * P + c] = (r < b && c < b) ? v[t] : ((r == c) ? 1.0 : 0.0);
        }
    }
    __syncthreads();
    // one-wave elimination of a 16-column block: lane -> row `r`, the block's columns c0 .. c0 + 15 of that row in registers; the pivot
    // row of step i sits in lane i.  Leaves U (rows above / on the diagonal), the multipliers (rows below) and the signs.
    auto eliminate16 = [&](const int r, const int c0, const bool rows_on) {
        double u[16], l[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) { u[c] = Bm[r * P + c0 + c]; l[c] = (r == c0 + c) ? 1.0 : 0.0; }
        double mysg = 1.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int gi = c0 + i;
            const double bii = cq_readlane(u[i], i);
            const bool live = gi < b;                      // uniform
            const double sgn = (bii >= 0.0) ? -1.0 : 1.0, piv = bii - sgn, rp = fast_rcp(live ? piv : 1.0);
            const double lv = (live && r > gi) ? u[i] * rp : 0.0;
#pragma unroll
            for (int c = i + 1; c < 16; ++c) {
                const double uic = cq_readlane(u[c], i);
                cq_fnma_s(u[c], uic, lv);
            }
            l[i] = (r > gi) ? lv : l[i];
            u[i] = (r > gi) ? 0.0 : ((r == gi && live) ? piv : u[i]);
            if (r == gi) mysg = live ? sgn : 1.0;
            __builtin_amdgcn_sched_barrier(0);
        }
        if (rows_on) {
#pragma unroll
            for (int c = 0; c < 16; ++c) { Um[r * P + c0 + c] = u[c]; Lm[r * P + c0 + c] = l[c]; }
            if (r >= c0 && r < c0 + 16) sg[r] = mysg;
        }
    };
    // inverse of a 16 x 16 triangular diagonal block by substitution, one column per lane (upper: of Um, lower unit: of Lm)
    auto inv_upper16 = [&](const int o, const int j) {
        double x[16], rd[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) rd[i] = fast_rcp(Um[(o + i) * P + o + i]);
#pragma unroll
        for (int i = 15; i >= 0; --i) {
            double sa[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = i + 1; k < 16; ++k) sa[k & 3] += Um[(o + i) * P + o + k] * x[k];
            const double t = (sa[0] + sa[1]) + (sa[2] + sa[3]);
            x[i] = (i > j) ? 0.0 : ((i == j) ? rd[i] : -t * rd[i]);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) Ui[(o + i) * P + o + j] = x[i];
    };
    auto inv_lower16 = [&](const int o, const int j) {
        double x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double sa[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < i; ++k) sa[k & 3] += Lm[(o + i) * P + o + k] * x[k];
            const double t = (sa[0] + sa[1]) + (sa[2] + sa[3]);
            x[i] = (i < j) ? 0.0 : ((i == j) ? 1.0 : -t);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) Li[(o + i) * P + o + j] = x[i];
    };
    // 16 x 16 x 16 products by one wave: D = sa A B + C, A from LDS (block at Am, pitch P), B from LDS or from the result tile of
    // the product before (register r = K step r)
    auto mm16 = [&](const double* Am, const double sa, const double* Bm_, const d4c* Breg, const d4c c0v) -> d4c {
        d4c acc = c0v;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k = ks * 4 + lk;
            const double fa = sa * Am[li_ * P + k];
            const double fb = Breg ? (*Breg)[ks] : Bm_[k * P + li_];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb, acc, 0, 0, 0);
        }
        return acc;
    };
    auto put16 = [&](double* Dm, const d4c v) {
#pragma unroll
        for (int q = 0; q < 4; ++q) Dm[(lk + 4 * q) * P + li_] = v[q];
    };
    const d4c zero4 = d4c{0.0, 0.0, 0.0, 0.0};
    FQ_CLK(14);
    // (1) first block column: rows 0 .. 31 in lanes 0 .. 31
    if (wave == 0) eliminate16(lane & 31, 0, lane < 32);
    __syncthreads();
    FQ_CLK(15);
    // (2) inverses of the first diagonal blocks (two waves side by side)
    if (wave == 0 && lane < 16) inv_upper16(0, lane);
    else if (wave == 1 && lane < 16) inv_lower16(0, lane);
    __syncthreads();
    FQ_CLK(16);
    // (3) U12 = L11^-1 B12, then the Schur complement S22 = B22 - L21 U12 straight from the result tile
    if (wave == 0) {
        const d4c u12 = mm16(Li, 1.0, Bm + 16, nullptr, zero4);
        d4c b22;
#pragma unroll
        for (int q = 0; q < 4; ++q) b22[q] = Bm[(16 + lk + 4 * q) * P + 16 + li_];
        const d4c s22 = mm16(Lm + 16 * P, -1.0, nullptr, &u12, b22);
        put16(Um + 16, u12);
        put16(Bm + 16 * P + 16, s22);
    } else if (wave == 1) {                                // blocks nobody computes: zeros
        for (int e = lane; e < 256; e += 64) {
            const int r = e >> 4, c = e & 15;
            Um[(16 + r) * P + c] = 0.0;                     // (U's lower-left block; the elimination left zeros there already)
            Lm[r * P + 16 + c] = 0.0;
            Li[r * P + 16 + c] = 0.0;
        }
    }
    __syncthreads();
    FQ_CLK(17);
    // (4) second diagonal block: rows 16 .. 31 in lanes 0 .. 15 (the other lanes mirror them)
    if (wave == 0) eliminate16(16 + (lane & 15), 16, lane < 16);
    __syncthreads();
    FQ_CLK(18);
    if (wave == 0 && lane < 16) inv_upper16(16, lane);
    else if (wave == 1 && lane < 16) inv_lower16(16, lane);
    __syncthreads();
    FQ_CLK(19);
    // (5) off-diagonal blocks of the inverses:  U^-1_12 = -U11^-1 (U12 U22^-1),  L^-1_21 = -L22^-1 (L21 L11^-1)
    if (wave == 0) {
        const d4c p = mm16(Um + 16, 1.0, Ui + 16 * P + 16, nullptr, zero4);
        const d4c x = mm16(Ui, -1.0, nullptr, &p, zero4);
        put16(Ui + 16, x);
    } else if (wave == 1) {
        const d4c p = mm16(Lm + 16 * P, 1.0, Li, nullptr, zero4);
        const d4c x = mm16(Li + 16 * P + 16, -1.0, nullptr, &p, zero4);
        put16(Li + 16 * P, x);
    } else if (wave == 2) {
        for (int e = lane; e < 256; e += 64) Ui[(16 + (e >> 4)) * P + (e & 15)] = 0.0;
    }
    __syncthreads();
    FQ_CLK(12);
    for (int e = tid; e < 1024; e += 256) Um[(e >> 5) * P + (e & 31)] *= sg[e & 31];      // U S, in place
    __syncthreads();
    // 32 x 32 products on the matrix cores, one wave each: acc[ti][tj] = op(A) op(B)
    auto mm = [&](const double* Am, const double* Bm, bool bt, d4c (&acc)[2][2]) {
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = d4c{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int k = ks * 4 + lk;
            double fa[2], fb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int i = t * 16 + li_;
                fa[t] = Am[i * P + k];
                fb[t] = bt ? Bm[i * P + k] : Bm[k * P + i];
            }
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[ti], fb[tj], acc[ti][tj], 0, 0, 0);
        }
    };
    d4c acc[2][2];
    if (wave == 0) mm(Um, Li, true, acc);                 // (U S) L^-T
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = ti * 16 + lk + 4 * q, j = tj * 16 + li_;
                    const double t = (i <= j && j < b) ? -acc[ti][tj][q] : 0.0;
                    Um[i * P + j] = t;                     // T
                    if (Tp && i < b && j < b) Tp[i * b + j] = t;
                }
    }
    __syncthreads();
    FQ_CLK(13);
    // wave 0: UT = Uinv T^T, wave 1: UTq = Uinv T, wave 2: Wtop = L T^T, wave 3: Wqtop = L T
    const bool q_wave = (wave & 1) == 1, top_wave = wave >= 2;
    const bool skip = (q_wave && !want_q) || (top_wave && !write_top);       // (wave-uniform)
    if (!skip) {
        mm((wave < 2) ? Ui : Lm, Um, (wave & 1) == 0, acc);
        double* dst = (Ss && wave < 2) ? Ss + (wave == 0 ? 1024 : 2048)
                                       : lu + (wave == 0 ? CQ_LU_UT : wave == 1 ? CQ_LU_UTQ : wave == 2 ? CQ_LU_WTOP : CQ_LU_WQTOP);
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = ti * 16 + lk + 4 * q, j = tj * 16 + li_;
                    dst[i * 32 + j] = (i < b && j < b) ? acc[ti][tj][q] : 0.0;
                }
    }
    for (int e = tid; e < 1024; e += 256) {
        const int i = e >> 5, j = e & 31;
        const double ui = (i < b && j < b) ? Ui[i * P + j] : 0.0;
        if (Ss) Ss[e] = ui; else lu[CQ_LU_UINV + e] = ui;
        if (write_top) lu[CQ_LU_Y1 + e] = Lm[i * P + j];
    }
    __syncthreads();
}

// Tail of a launch, run by the last workgroup to arrive: sum the partial Gram matrices (block order, optional per-block
// power-of-two weights), measure the distance from the identity, decide, and factor G = R^T R with deferral (wave 0).
// Gs: LDS 32 x 33, Rs: LDS 32 x 32 (16-byte aligned).  pass = number of passes applied to the panel whose Gram matrix this is.  Returns 1 when the caller
// decision (0: factor again, 1: converged, 2: out of passes; all threads get the same value).
__device__ __forceinline__ int cq_tail(const double* part, const int* bexp, int nblk, int b, int pass, CqState* stt, double* Rg, double* Gs,
                                       double* Rs, int maxpass, int tid, long long elems, int slot) {
    __shared__ int s_dec;
    const int lane = tid & 63;
    // exponents of pass 0: block weights 4^(e_blk - emax); every wave finds emax itself (no barrier)
    int emax = 0;
    if (bexp) {
        int e = -100000;
        for (int i = lane; i < nblk; i += 64) { const int x = cq_ldi(bexp + i); e = x > e ? x : e; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const int y = __shfl_xor(e, o, 64); e = y > e ? y : e; }
        emax = e;
    }
    {   // thread tid sums entries tid, tid + 256, tid + 512 over the blocks, in block order; 96 loads in flight per thread
        double acc[3] = {0.0, 0.0, 0.0};
        for (int blk0 = 0; blk0 < nblk; blk0 += 32) {
            double v[3][32];
            int ex[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const int blk = blk0 + u;
                const bool in = blk < nblk;
                ex[u] = (bexp && in) ? cq_ldi(bexp + blk) : emax;
#pragma unroll
                for (int q = 0; q < 3; ++q) v[q][u] = in ? cq_ld(part + (int64_t)blk * CQ_PART + tid + 256 * q) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const double w = bexp ? ldexp(1.0, 2 * (ex[u] - emax)) : 1.0;
#pragma unroll
                for (int q = 0; q < 3; ++q) acc[q] = fma(w, v[q][u], acc[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int e = tid + 256 * q, i = (e >> 4) & 15, j = e & 15;
            const int gi = (q == 2 ? 16 : 0) + i, gj = (q == 0 ? 0 : 16) + j;
            Gs[gi * CQ_P + gj] = acc[q];
            if (q == 1) Gs[gj * CQ_P + gi] = acc[q];       // the diagonal tiles carry both triangles already
        }
    }
    __syncthreads();
    if (tid < 64) {
        // ---- wave 0: lane k (mod 32) holds column k of G; the padding block (rows / columns >= b) becomes the identity
        const int k = lane & 31;
        double g[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const double x = Gs[i * CQ_P + k];
            g[i] = (i < b && k < b) ? x : ((i == k) ? 1.0 : 0.0);
        }
        double dev = 0.0;                                 // distance from the identity
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const double x = fabs(g[i] - (i == k ? 1.0 : 0.0));
            dev = (x == x) ? fmax(dev, x) : 1e300;        // NaN counts as "far"
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dev = fmax(dev, __shfl_xor(dev, o, 64));
        int dec = 0;                                      // 0: factor again, 1: converged, 2: out of passes -> fallback
        if (pass > 0 && dev <= CQ_DONE) dec = 1;
        else if (pass >= maxpass) dec = 2;
        if (lane == 0) {
            stt->dev_hist[pass <= CQ_MAXPASS ? pass : CQ_MAXPASS] = dev;
            if (bexp) stt->emax = emax;
            s_dec = dec;
        }
        if (dec != 0) {
            if (lane == 0) {
                stt->done = 1;
                stt->final_next = 0;
                stt->dead = 0u;
                if (dec == 2) stt->fallback = 1;
                cq_count(slot, pass, stt->ndefer_total, stt->nrefill_total, dec == 2, elems);
            }
        } else {
            // ---- Cholesky, right-looking.  Lane k holds column k of the trailing matrix; row j of R (lane k: R[j][k]) goes
            // to LDS, from where every lane reads the multipliers R[j][i] as broadcasts (a cross-lane read per multiplier
            // would cost three VALU instructions per update instead of one).  Only the multiplier of row j+1, which the next
            // pivot waits for, is read across lanes, so the LDS round trip stays off the critical path.
            double gd = 0.0;                              // squared norm of column k before any reduction
#pragma unroll
            for (int i = 0; i < 32; ++i) gd = (i == k) ? g[i] : gd;
            // a zero (underflowing, non-finite) column is refilled with noise in the next pass; a pivot below CQ_THETA of the
            // column's squared norm is deferred; either way row j of R = e_j
            const bool zero_k = !(gd > 1e-290) || !(gd < 1e300);
            const unsigned deadmask = (unsigned)(__ballot(zero_k) & 0xffffffffull);
            const double thr_k = zero_k ? 1e308 : CQ_THETA * gd;
            unsigned badmask = 0u;
            double dkk = 1.0;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const double d = cq_readlane(g[j], j), thr = cq_readlane(thr_k, j);
                const bool ok = d > thr;                  // uniform
                badmask |= ok ? 0u : (1u << j);
                const double rinv = cq_rsqrt2(ok ? d : 1.0);
                double r = (k >= j) ? g[j] * (ok ? rinv : 0.0) : 0.0;
                r = (!ok && k == j) ? 1.0 : r;
                if (j == k) dkk = r;
                if (lane < 32) Rs[j * 32 + k] = r;
                if (j < 31) {
                    const double m1 = cq_readlane(r, j + 1);
                    const int i0 = (j + 3) & ~1;          // 16-byte pairs (i, i+1) with i even; an odd row j+2 goes alone
                    double2 m[16];
                    double m2 = 0.0;
                    if (j + 2 < 32 && ((j + 2) & 1)) m2 = Rs[j * 32 + j + 2];
#pragma unroll
                    for (int i = i0; i < 32; i += 2) m[i >> 1] = *reinterpret_cast<const double2*>(&Rs[j * 32 + i]);   // all reads in flight
                    cq_fnma(g[j + 1], m1, r);
                    if (j + 2 < 32 && ((j + 2) & 1)) cq_fnma(g[j + 2], m2, r);
#pragma unroll
                    for (int i = i0; i < 32; i += 2) {
                        cq_fnma(g[i], m[i >> 1].x, r);
                        cq_fnma(g[i + 1], m[i >> 1].y, r);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);        // keep the updates of step j in step j (the scheduler otherwise
            }                                             // sinks them into 31-long dependent chains and spills the multipliers)
            __builtin_amdgcn_wave_barrier();
            for (int e = lane; e < 1024; e += 64) Rg[e] = Rs[e];
            if (lane < 32) Rg[1024 + k] = fast_rcp(dkk);  // reciprocal diagonal for the substitution
            if (lane == 0) {
                const int ndefer = __popc(badmask & ~deadmask);
                stt->final_next = (pass > 0 && dev <= CQ_LAST && badmask == 0u) ? 1 : 0;
                stt->dead = deadmask;
                stt->ndefer_total += ndefer;
                stt->nrefill_total += __popc(deadmask);
            }
        }
    }
    __syncthreads();
    return s_dec;                                          // 0: another pass, 1: converged, 2: fallback flagged
}

// Pass 0: rows r0 .. r0+nr-1 of the panel into the LDS tile T (zero padded), scaled by the power of two that brings the tile's
// largest entry into [0.5, 1) so that squares neither overflow nor underflow whatever the input scale; returns that exponent
// (-2000 for an all-zero tile, which then weighs nothing).  red: LDS, 4 doubles.  Ends with a barrier.
__device__ __forceinline__ int cq_load_scaled_tile(const double* __restrict__ X, int64_t rs, int64_t cs, int64_t r0, int nr, int b, double* T,
                                                   double* red, int tid) {
    const bool colfast = (cs == 1);
    double amax = 0.0;
    {
        double xv[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int e = tid + 256 * u;
            const int i = colfast ? e >> 5 : e & 255, j = colfast ? e & 31 : e >> 8;
            xv[u] = (i < nr && j < b) ? X[(r0 + i) * rs + j * cs] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int e = tid + 256 * u;
            const int i = colfast ? e >> 5 : e & 255, j = colfast ? e & 31 : e >> 8;
            T[i * CQ_P + j] = xv[u];
            const double a = fabs(xv[u]);
            amax = (a == a) ? fmax(amax, a) : 1.7e308;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmax(amax, __shfl_xor(amax, o, 64));
    if ((tid & 63) == 0) red[tid >> 6] = amax;
    __syncthreads();
    amax = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    int ex = -2000;                                       // an all-zero block weighs nothing
    if (amax > 0.0 && amax < 1.7e308) frexp(amax, &ex);
    const double scl = (ex > -2000) ? ldexp(1.0, -ex) : 0.0;
#pragma unroll
    for (int u = 0; u < 32; ++u) {
        const int e = tid + 256 * u;
        T[(e >> 5) * CQ_P + (e & 31)] *= scl;
    }
    __syncthreads();
    return ex;
}

// ---- pass 0: Gram matrix of the input panel -------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cq_gram_kernel(const double* __restrict__ X, int64_t rs, int64_t cs, int64_t nrows, int b, int nblk,
                                                      double* part, int* bexp, CqState* stt, double* Rg, int slot, const int* active) {
    __shared__ double T[CQ_RB * CQ_P];
    __shared__ double Gs[32 * CQ_P];
    __shared__ __attribute__((aligned(16))) double Rs[32 * 32];
    __shared__ double red[4];
    __shared__ int s_ticket;
    const int tid = threadIdx.x, blk = blockIdx.x;
    if (active && *active == 0) return;                      // the factorisation this panel belongs to has stopped (device-side exit test, qr.hip)
    int64_t r0;
    int nr;
    cq_block_rows(nrows, nblk, blk, r0, nr);
    const int ex = cq_load_scaled_tile(X, rs, cs, r0, nr, b, T, red, tid);
    cq_block_gram(T, T, part + (int64_t)blk * CQ_PART, tid);
    if (tid == 0) cq_sti(bexp + blk, ex);
    cq_publish_wait();
    __syncthreads();
    if (tid == 0) s_ticket = atomicAdd(&stt->counter, 1);
    __syncthreads();
    if (s_ticket != nblk - 1) return;
    __threadfence();
    if (tid == 0) { stt->counter = 0; stt->done = 0; stt->pass = 0; stt->ndefer_total = 0; stt->nrefill_total = 0; stt->fallback = 0; }
    __syncthreads();
    cq_tail(part, bexp, nblk, b, 0, stt, Rg, Gs, Rs, CQ_MAXPASS, tid, (long long)nrows * b, slot);
}

// X <- X R^-1 on the 256-row tile T (LDS), row tid; Rs (LDS, 16-byte aligned): R row-major (1024) + reciprocal diagonal (32).
// Columns flagged in deadmask (exactly zero before this pass) are refilled with hash noise.
__device__ __forceinline__ void cq_substitute(double* T, const double* Rs, int tid, unsigned deadmask, uint64_t seed, int64_t r0, int nr,
                                              double scl = 1.0) {
    {   // substitution on row tid (right-looking: after step j all later columns are independent updates).  The multipliers
        // of step j+1 are fetched from LDS (broadcast reads) while step j computes; the scheduling barriers keep the compiler
        // from hoisting all 250 reads to the top (512 VGPRs and spills otherwise).
        double x[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) x[j] = T[tid * CQ_P + j] * scl;
        double2 mc[16], mn[16];
        double dc = Rs[1024], dn = 0.0, sc = Rs[1], sn = 0.0;      // reciprocal diagonal, the odd first multiplier
#pragma unroll
        for (int q = 1; q < 16; ++q) mc[q] = *reinterpret_cast<const double2*>(&Rs[2 * q]);
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            if (j < 31) {                                 // prefetch step j+1
                dn = Rs[1024 + j + 1];
                if (j + 2 < 32 && ((j + 2) & 1)) sn = Rs[(j + 1) * 32 + j + 2];
#pragma unroll
                for (int q = (j + 3) >> 1; q < 16; ++q) mn[q] = *reinterpret_cast<const double2*>(&Rs[(j + 1) * 32 + 2 * q]);
            }
            const double xj = x[j] * dc;
            x[j] = xj;
            if (j < 31) {
                if ((j + 1) & 1) cq_fnma(x[j + 1], xj, sc);
#pragma unroll
                for (int q = (j + 2) >> 1; q < 16; ++q) {
                    cq_fnma(x[2 * q], xj, mc[q].x);
                    cq_fnma(x[2 * q + 1], xj, mc[q].y);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            dc = dn; sc = sn;
#pragma unroll
            for (int q = 0; q < 16; ++q) mc[q] = mn[q];
        }
        if (deadmask) {
#pragma unroll
            for (int j = 0; j < 32; ++j)
                if ((deadmask >> j) & 1u) x[j] = (tid < nr) ? cq_hash_unit(seed + (uint64_t)(r0 + tid) * 64 + j) : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) T[tid * CQ_P + j] = x[j];
    }
}

// the tile back to global memory, coalesced whatever the layout
__device__ __forceinline__ void cq_store_tile(const double* T, double* Y, int64_t rs, int64_t cs, int64_t r0, int nr, int b, int tid) {
    {   // coalesced store of the tile
        const bool ofast = (cs == 1);
#pragma unroll
        for (int u0 = 0; u0 < 32; u0 += 8) {
            double ov[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = tid + 256 * (u0 + u);
                const int i = ofast ? e >> 5 : e & 255, j = ofast ? e & 31 : e >> 8;
                ov[u] = T[i * CQ_P + j];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = tid + 256 * (u0 + u);
                const int i = ofast ? e >> 5 : e & 255, j = ofast ? e & 31 : e >> 8;
                if (i < nr && j < b) Y[(r0 + i) * rs + j * cs] = ov[u];
            }
        }
    }
}

// ---- pass t >= 1: X <- X R^-1, then the Gram matrix of the new panel -----------------------------------------------
// first: the source is the caller's panel (scaled by 2^-emax on the way in), later passes work in place on Y.
__global__ __launch_bounds__(256) void cq_pass_kernel(const double* Xsrc, int64_t srs, int64_t scs, double* Y, int64_t rs, int64_t cs,
                                                      int64_t nrows, int b, int nblk, int first, int launch_no, double* part, CqState* stt,
                                                      double* Rg, uint64_t seed, double* lu, double* Tp, int maxpass, int slot, const int* active) {
    __shared__ double T[CQ_RB * CQ_P];
    __shared__ double Gs[32 * CQ_P];
    __shared__ __attribute__((aligned(16))) double Rs[32 * 32 + 32 + 32 * 32];      // factor + reciprocals | Cholesky rows of the tail
    __shared__ int s_ticket;
    __shared__ int s_st[4];
    const int tid = threadIdx.x, blk = blockIdx.x;
    if (active && *active == 0) return;
    CQ_CLK_DECL;
    CQ_CLK(0);
    int64_t r0;
    int nr;
    cq_block_rows(nrows, nblk, blk, r0, nr);
    const double* src = first ? Xsrc : Y;
    const int64_t xrs = first ? srs : rs, xcs = first ? scs : cs;
    const bool colfast = (xcs == 1);
    if (launch_no >= 3) {                                 // 94 % of the third and 99.8 % of the fourth passes find the panel converged:
        if (tid == 0) s_st[0] = cq_ldi(&stt->done);       // look before fetching the tile
        __syncthreads();
        if (s_st[0]) return;
        __syncthreads();
    }
    {   // the state, the tile and the triangular factor in one memory round trip (a launch that finds the panel converged
        // throws the tile away)
        double xv[32], rv[5];
        int sv = 0;
        if (tid < 4) sv = cq_ldi(tid == 0 ? &stt->done : tid == 1 ? &stt->final_next : tid == 2 ? &stt->emax : (const int*)&stt->dead);
#pragma unroll
        for (int u = 0; u < 5; ++u) rv[u] = (tid + 256 * u < 1056) ? Rg[tid + 256 * u] : 0.0;
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int e = tid + 256 * u;
            const int i = colfast ? e >> 5 : e & 255, j = colfast ? e & 31 : e >> 8;
            xv[u] = (i < nr && j < b) ? src[(r0 + i) * xrs + j * xcs] : 0.0;
        }
        if (tid < 4) s_st[tid] = sv;
        __syncthreads();
        if (s_st[0]) return;
        const int emax0 = s_st[2];
        const double scl = first ? ((emax0 > -2000) ? ldexp(1.0, -emax0) : 0.0) : 1.0;
#pragma unroll
        for (int u = 0; u < 5; ++u)
            if (tid + 256 * u < 1056) Rs[tid + 256 * u] = rv[u];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int e = tid + 256 * u;
            const int i = colfast ? e >> 5 : e & 255, j = colfast ? e & 31 : e >> 8;
            T[i * CQ_P + j] = xv[u] * scl;
        }
    }
    const int fin = s_st[1], emax = s_st[2];
    const unsigned deadmask = (unsigned)s_st[3];
    __syncthreads();
    CQ_CLK(1);
    cq_substitute(T, Rs, tid, deadmask, seed, r0, nr);
    __syncthreads();
    CQ_CLK(2);
    cq_store_tile(T, Y, rs, cs, r0, nr, b, tid);
    // The Householder reconstruction only needs the top block of the panel, which workgroup 0 owns: it keeps a copy (Rs is free
    // after the substitution) and reconstructs after it has taken its ticket -- speculatively, while the last workgroup is still
    // reducing and deciding; if the panel turns out not to be converged the next pass simply overwrites what it wrote.
    const bool recon = (lu != nullptr) && (blk == 0);
    if (recon) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + 256 * u;
            Rs[e] = T[(e >> 5) * CQ_P + (e & 31)];
        }
    }
    if (!fin) cq_block_gram(T, T, part + (int64_t)blk * CQ_PART, tid);      // (its first barrier follows its reads of T)
    CQ_CLK(3);
    cq_publish_wait();
    __syncthreads();
    CQ_CLK(4);
    if (tid == 0) s_ticket = atomicAdd(&stt->counter, 1);
    __syncthreads();
    CQ_CLK(5);
    if (s_ticket != nblk - 1) {
        if (recon) cq_lu(Rs, false, b, lu, Tp, T, tid);
        return;
    }
    __threadfence();
    CQ_CLK(6);
    if (tid == 0) { stt->counter = 0; stt->pass = launch_no; }
    int dec = 1;
    if (fin) {
        if (tid == 0) {
            stt->done = 1; stt->final_next = 0; stt->dead = 0u;
            cq_count(slot, launch_no, stt->ndefer_total, stt->nrefill_total, false, (long long)nrows * b);
        }
        __syncthreads();
    } else {
        __syncthreads();
        dec = cq_tail(part, nullptr, nblk, b, launch_no, stt, Rg, Gs, Rs + 1056, maxpass, tid, (long long)nrows * b, slot);
    }
    CQ_CLK(7);
    if (launch_no <= 2) CQ_CLK_DUMP(12 * (launch_no - 1));
    if (recon && dec == 1) cq_lu(Rs, false, b, lu, Tp, T, tid);          // workgroup 0 arrived last: reconstruct now
}

// ---- post-processing: the reflector panels from the orthonormal one ---------------------------------------------------
// rows >= b:  Y <- Q1 Uinv,  W <- Q1 UT (= Y T^T),  Wq <- Q1 UTq (= Y T, optional);  rows < b come from the reconstruction.
// One tile of <= 256 rows on the matrix cores (wave w owns rows 64w .. 64w+63); tile: LDS 256 x CQ_P, Ss: LDS 3 x 1024.
// tile_ready: the orthonormal tile is in `tile` already (single-launch form), otherwise it is fetched from Y.
__device__ __forceinline__ void cq_post_tile(int blk, int nblk, int64_t nrows, int b, double* Y, int64_t rs, int64_t cs, double* W, int64_t wrs,
                                             int64_t wcs, double* Wq, const double* lu, double* tile, double* Ss, int tid, bool tile_ready = false,
                                             bool ss_ready = false) {
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    int64_t r0;
    int nr;
    cq_block_rows(nrows, nblk, blk, r0, nr);
    const bool xrow = (cs == 1);
    {
        double sv[3][4], xv[32];
        if (!ss_ready) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int e = tid + 256 * t;
                sv[0][t] = lu[CQ_LU_UINV + e];
                sv[1][t] = lu[CQ_LU_UT + e];
                sv[2][t] = lu[CQ_LU_UTQ + e];
            }
        }
        if (!tile_ready) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const int e = tid + 256 * u;
                const int i = xrow ? e >> 5 : e & 255, j = xrow ? e & 31 : e >> 8;
                xv[u] = (i < nr && j < b) ? Y[(r0 + i) * rs + j * cs] : 0.0;
            }
        }
        if (!ss_ready) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int e = tid + 256 * t;
                Ss[e] = sv[0][t]; Ss[1024 + e] = sv[1][t]; Ss[2048 + e] = sv[2][t];
            }
        }
        if (!tile_ready) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const int e = tid + 256 * u;
                const int i = xrow ? e >> 5 : e & 255, j = xrow ? e & 31 : e >> 8;
                tile[i * CQ_P + j] = xv[u];
            }
        }
    }
    __syncthreads();
    double fa[4][8];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) fa[mt][ks] = tile[(wave * 64 + mt * 16 + li) * CQ_P + ks * 4 + lk];
    __syncthreads();
#pragma unroll 1
    for (int o = 0; o < 3; ++o) {
        double* dst = (o == 0) ? Y : (o == 1) ? W : Wq;
        if (dst == nullptr) continue;                      // uniform
        const int64_t drs = (o == 1) ? wrs : rs, dcs = (o == 1) ? wcs : cs;
        const double* Sm = Ss + 1024 * o;
        const double* top = lu + ((o == 0) ? CQ_LU_Y1 : (o == 1) ? CQ_LU_WTOP : CQ_LU_WQTOP);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            double fb[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) fb[ks] = Sm[(ks * 4 + lk) * 32 + nt * 16 + li];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                d4c acc = d4c{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[mt][ks], fb[ks], acc, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) tile[(wave * 64 + mt * 16 + lk + 4 * q) * CQ_P + nt * 16 + li] = acc[q];
            }
        }
        __syncthreads();
        const bool drow = (dcs == 1);
        // 16-byte stores along the unit-stride direction where the layout allows (full panels): the epilogue of three 64 KB outputs is
        // bound by store issue (~7 B / cycle / CU with 8-byte stores)
        const bool wide = b == 32 && (((uintptr_t)dst & 15) == 0) &&
                          (drow ? (drs & 1) == 0 : (drs == 1 && (dcs & 1) == 0 && (r0 & 1) == 0 && (nr & 1) == 0));
        if (wide) {
#pragma unroll
            for (int u0 = 0; u0 < 16; u0 += 8) {
                double2 ov[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = tid + 256 * (u0 + u);
                    const int i = drow ? e >> 4 : 2 * (e & 127), j = drow ? 2 * (e & 15) : e >> 7;
                    ov[u] = make_double2(tile[i * CQ_P + j], tile[(drow ? i : i + 1) * CQ_P + (drow ? j + 1 : j)]);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = tid + 256 * (u0 + u);
                    const int i = drow ? e >> 4 : 2 * (e & 127), j = drow ? 2 * (e & 15) : e >> 7;
                    if (i < nr && r0 + i >= b) *reinterpret_cast<double2*>(dst + (r0 + i) * drs + j * dcs) = ov[u];
                }
            }
        } else {
#pragma unroll
        for (int u0 = 0; u0 < 32; u0 += 8) {
            double ov[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = tid + 256 * (u0 + u);
                const int i = drow ? e >> 5 : e & 255, j = drow ? e & 31 : e >> 8;
                ov[u] = tile[i * CQ_P + j];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = tid + 256 * (u0 + u);
                const int i = drow ? e >> 5 : e & 255, j = drow ? e & 31 : e >> 8;
                if (i < nr && j < b && r0 + i >= b) dst[(r0 + i) * drs + j * dcs] = ov[u];
            }
        }
        }
        if (r0 < b) {                                     // (the first tile only) the top block comes from the reconstruction
            for (int e = tid; e < 1024; e += 256) {
                const int i = e >> 5, j = e & 31;
                if (i >= r0 && i - r0 < nr && i < b && j < b) dst[(int64_t)i * drs + j * dcs] = top[i * 32 + j];
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void cq_post_kernel(const double* X, int64_t xrs, int64_t xcs, double* Y, int64_t rs, int64_t cs, int64_t nrows,
                                                      int b, int nblk, CqState* stt, double* lu, double* Tp, double* W, int64_t wrs, int64_t wcs,
                                                      double* Wq, const int* active) {
    __shared__ double tile[CQ_RB * CQ_P];
    __shared__ double Ss[3 * 1024];
    __shared__ double scr[4 * 32 * 33 + 64];
    __shared__ int s_st[2];
    const int tid = threadIdx.x, blk = blockIdx.x;
    if (active && *active == 0) return;
    if (tid == 0) { s_st[0] = cq_ldi(&stt->fallback); s_st[1] = cq_ldi(&stt->emax); }
    __syncthreads();
    if (!s_st[0]) {
        if (lu) cq_post_tile(blk, nblk, nrows, b, Y, rs, cs, W, wrs, wcs, Wq, lu, tile, Ss, tid);
        return;
    }
    // The passes did not converge (never seen on the contraction path): workgroup 0 redoes the panel with Householder
    // reflections and post-processes every tile itself.
    if (blk != 0) return;
    cq_fallback_householder(X, xrs, xcs, Y, rs, cs, nrows, b, s_st[1], scr, tid);
    if (!lu) return;
    __threadfence();
    __syncthreads();
    for (int e = tid; e < 1024; e += 256) {
        const int i = e >> 5, j = e & 31;
        lu[CQ_LU_YTOP + e] = (i < b && j < b && i < nrows) ? cq_ld(Y + (int64_t)i * rs + j * cs) : 0.0;
    }
    __threadfence();
    __syncthreads();
    cq_lu(lu + CQ_LU_YTOP, true, b, lu, Tp, scr, tid);
    __threadfence();
    __syncthreads();
    for (int t = 0; t < nblk; ++t) {
        cq_post_tile(t, nblk, nrows, b, Y, rs, cs, W, wrs, wcs, Wq, lu, tile, Ss, tid);
        __threadfence();
        __syncthreads();
    }
}

// ---- the whole panel step in ONE launch ------------------------------------------------------------------------------
// For panels of up to CQ_FUSED_MAXBLK x 256 = 8192 rows -- and for taller ones, up to 16384 rows, when cq_big_admit lets them in -- the chain
// gram -> pass ... pass -> post  runs inside one kernel: every workgroup keeps its 256-row tile in
// LDS from the first load to the last store (the six-launch form reloads and stores it in every launch), the workgroups meet at
// in-kernel barriers (a monotone arrival counter polled by one lane, MI355X guide "Guideline 16": partials written with
// agent-scope stores, drained, one atomic add per workgroup, relaxed agent-scope poll, agent-scope loads of the partials),
// and after each barrier EVERY workgroup reduces the published partial Gram matrices, takes the decision and factors G itself
// -- the same arithmetic in the same order as cq_tail, hence the same bits in every workgroup and the same result as the
// six-launch form -- so no second hand-off is needed to distribute R.  The Householder reconstruction is redundant in the same
// way: workgroup 0 publishes the top 32 x 32 block of the panel next to its partial Gram matrix.
// Co-residency: a workgroup needs a whole CU (127 KB of LDS) and only ever waits for workgroups of its OWN launch, so the launches in
// flight can deadlock only if together they ask for more CUs than the chip has.  At most CQ_FUSED_MAXBLK = 32 workgroups per launch and
// at most 8 launches in flight (the 8 hardware queues the package asks for, one chain per queue) make 256 = the CUs of an MI355X; every
// other kernel on the device finishes without waiting for anyone.  Should a process be configured with more queues than that, the
// spins are bounded (CQ_SPIN_LIMIT polls, seconds): a launch that gives up poisons its output with NaN and raises the sticky
// stt->timeout, which later launches of the call honour.
constexpr int CQ_FUSED_MAXBLK = 32;
constexpr unsigned CQ_SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ bool cq_grid_barrier(int* counter, int target, int* s_flag, int tid, unsigned spin_limit = CQ_SPIN_LIMIT) {
    cq_publish_wait();                                     // every wave: its agent-scope stores have completed
    __syncthreads();
    if (tid == 0) {
        atomicAdd(counter, 1);
        int ok = 1;
        unsigned spins = 0;
        while (cq_ldi(counter) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > spin_limit) { ok = 0; break; }
        }
        *s_flag = ok;
    }
    __syncthreads();
    return *s_flag != 0;
}

// cq_tail for the single-launch form, run by EVERY workgroup on the same published partials: sum (block order, optional
// per-block power-of-two weights), distance from the identity, decision, Cholesky with deferral.  The factor goes to Rf (LDS:
// 32 x 32 row-major + the 32 reciprocals of its diagonal) instead of global memory.  s_out (LDS, 4 ints): [0] decision
// (0 factor again, 1 converged, 2 out of passes), [1] final_next, [2] dead-column mask, [3] emax (pass 0).  Only `writer`
// (workgroup 0) keeps the panel's state block and the statistics.  Ends with a barrier.
// emax_known > -100000: `part` is ONE pre-summed matrix (sliced reduction, see cq_slice_reduce) of a pass-0 Gram whose weights were
// applied with that exponent.
__device__ __forceinline__ void cq_tail_fused(const double* part, const int* bexp, int nblk, int b, int pass, CqState* stt, bool writer, double* Gs,
                                              double* Rf, int* s_out, int maxpass, int tid, long long elems, int slot, int emax_known = -100001) {
    const int lane = tid & 63;
    int emax = emax_known > -100000 ? emax_known : 0;
    if (bexp) {
        int e = -100000;
        for (int i = lane; i < nblk; i += 64) { const int x = cq_ldi(bexp + i); e = x > e ? x : e; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const int y = __shfl_xor(e, o, 64); e = y > e ? y : e; }
        emax = e;
    }
    {
        double acc[3] = {0.0, 0.0, 0.0};
        for (int blk0 = 0; blk0 < nblk; blk0 += 32) {
            double v[3][32];
            int ex[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const int blk = blk0 + u;
                const bool in = blk < nblk;
                ex[u] = (bexp && in) ? cq_ldi(bexp + blk) : emax;
#pragma unroll
                for (int q = 0; q < 3; ++q) v[q][u] = in ? cq_ld(part + (int64_t)blk * CQ_PART + tid + 256 * q) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const double w = bexp ? ldexp(1.0, 2 * (ex[u] - emax)) : 1.0;
#pragma unroll
                for (int q = 0; q < 3; ++q) acc[q] = fma(w, v[q][u], acc[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int e = tid + 256 * q, i = (e >> 4) & 15, j = e & 15;
            const int gi = (q == 2 ? 16 : 0) + i, gj = (q == 0 ? 0 : 16) + j;
            Gs[gi * CQ_P + gj] = acc[q];
            if (q == 1) Gs[gj * CQ_P + gi] = acc[q];
        }
    }
    __syncthreads();
    if (tid < 64) {
        const int k = lane & 31;
        double g[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const double x = Gs[i * CQ_P + k];
            g[i] = (i < b && k < b) ? x : ((i == k) ? 1.0 : 0.0);
        }
        double dev = 0.0;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const double x = fabs(g[i] - (i == k ? 1.0 : 0.0));
            dev = (x == x) ? fmax(dev, x) : 1e300;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dev = fmax(dev, __shfl_xor(dev, o, 64));
        int dec = 0;
        if (pass > 0 && dev <= CQ_DONE) dec = 1;
        else if (pass >= maxpass) dec = 2;
        if (lane == 0) {
            s_out[0] = dec;
            s_out[3] = emax;
            if (dec != 0) { s_out[1] = 0; s_out[2] = 0; }
            if (writer) {
                stt->dev_hist[pass <= CQ_MAXPASS ? pass : CQ_MAXPASS] = dev;
                if (bexp || emax_known > -100000) stt->emax = emax;
                if (dec != 0) {
                    stt->done = 1;
                    stt->final_next = 0;
                    stt->dead = 0u;
                    if (dec == 2) stt->fallback = 1;
                    cq_count(slot, pass, stt->ndefer_total, stt->nrefill_total, dec == 2, elems, true);
                }
            }
        }
        if (dec == 0) {
            double gd = 0.0;
#pragma unroll
            for (int i = 0; i < 32; ++i) gd = (i == k) ? g[i] : gd;
            const bool zero_k = !(gd > 1e-290) || !(gd < 1e300);
            const unsigned deadmask = (unsigned)(__ballot(zero_k) & 0xffffffffull);
            const double thr_k = zero_k ? 1e308 : CQ_THETA * gd;
            unsigned badmask = 0u;
            double dkk = 1.0;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const double d = cq_readlane(g[j], j), thr = cq_readlane(thr_k, j);
                const bool ok = d > thr;
                badmask |= ok ? 0u : (1u << j);
                const double rinv = cq_rsqrt2(ok ? d : 1.0);
                double r = (k >= j) ? g[j] * (ok ? rinv : 0.0) : 0.0;
                r = (!ok && k == j) ? 1.0 : r;
                if (j == k) dkk = r;
                if (lane < 32) Rf[j * 32 + k] = r;
                if (j < 31) {
                    const double m1 = cq_readlane(r, j + 1);
                    const int i0 = (j + 3) & ~1;
                    double2 m[16];
                    double m2 = 0.0;
                    if (j + 2 < 32 && ((j + 2) & 1)) m2 = Rf[j * 32 + j + 2];
#pragma unroll
                    for (int i = i0; i < 32; i += 2) m[i >> 1] = *reinterpret_cast<const double2*>(&Rf[j * 32 + i]);
                    cq_fnma(g[j + 1], m1, r);
                    if (j + 2 < 32 && ((j + 2) & 1)) cq_fnma(g[j + 2], m2, r);
#pragma unroll
                    for (int i = i0; i < 32; i += 2) {
                        cq_fnma(g[i], m[i >> 1].x, r);
                        cq_fnma(g[i + 1], m[i >> 1].y, r);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_wave_barrier();
            if (lane < 32) Rf[1024 + k] = fast_rcp(dkk);
            if (lane == 0) {
                const int fin = (pass > 0 && dev <= CQ_LAST && badmask == 0u) ? 1 : 0;
                s_out[1] = fin;
                s_out[2] = (int)deadmask;
                if (writer) {
                    stt->final_next = fin;
                    stt->dead = deadmask;
                    stt->ndefer_total += __popc(badmask & ~deadmask);
                    stt->nrefill_total += __popc(deadmask);
                }
            }
        }
    }
    __syncthreads();
}

// Sliced reduction of the published partial Gram matrices (launches of more than CQ_SLICE_FROM workgroups; an in-kernel barrier costs 1-2 us): workgroup w adds slice w of
// all nblk partials -- block order, the same fused multiply-adds with the same power-of-two weights as the full reduction, hence the
// same bits -- and publishes its slice of the sum; after one more barrier everybody fetches the 768 sums.  nblk^2 x 6 KB of agent-scope
// loads per round become 2 x nblk x 6 KB.  Returns the exponent the weights refer to (pass 0).
constexpr int CQ_SLICE_FROM = 4;
__device__ __forceinline__ int cq_slice_reduce(const double* part, const int* bexp, int nblk, int blk, double* gsum, int tid) {
    const int lane = tid & 63;
    int emax = 0;
    if (bexp) {
        int e = -100000;
        for (int i = lane; i < nblk; i += 64) { const int x = cq_ldi(bexp + i); e = x > e ? x : e; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const int y = __shfl_xor(e, o, 64); e = y > e ? y : e; }
        emax = e;
    }
    const int sl = (CQ_PART + nblk - 1) / nblk;            // <= 154 entries per workgroup
    const int e0 = blk * sl + tid;
    if (tid < sl && e0 < CQ_PART) {
        double acc = 0.0;
        for (int b0 = 0; b0 < nblk; b0 += 16) {
            double v[16];
            int ex[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const bool in = b0 + u < nblk;
                v[u] = in ? cq_ld(part + (int64_t)(b0 + u) * CQ_PART + e0) : 0.0;
                ex[u] = (bexp && in) ? cq_ldi(bexp + b0 + u) : emax;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = fma(bexp ? ldexp(1.0, 2 * (ex[u] - emax)) : 1.0, v[u], acc);
        }
        cq_st(gsum + e0, acc);
    }
    return emax;
}

// part: 2 x nblk x CQ_PART (pass parity), topblk: 2 x 1024 (pass parity), lu_all: nblk x CQ_LU_DOUBLES (private to each workgroup)
// or NULL (orthonormalisation only).  base: arrivals booked on stt->fcounter by the earlier launches of the call; every launch books
// exactly (maxpass + 1) * nblk (twice that from CQ_SLICE_FROM workgroups on: one more barrier per pass).
__global__ __launch_bounds__(256) void cq_fused_kernel(const double* X, int64_t xrs, int64_t xcs, double* Y, int64_t rs, int64_t cs, int64_t nrows,
                                                       int b, int nblk, double* part, int* bexp, double* topblk, CqState* stt, int base,
                                                       uint64_t seed, double* lu_all, double* Tp, double* W, int64_t wrs, int64_t wcs, double* Wq,
                                                       int maxpass, int slot, unsigned spin_limit, double* gsum, const int* active) {
    if (active && *active == 0) {                            // the factorisation has stopped (device-side exit test, qr.hip): every workgroup leaves,
        if (threadIdx.x == 0) atomicAdd(&stt->fcounter, (maxpass + 1) * (nblk > CQ_SLICE_FROM ? 2 : 1));      // its arrivals booked as the host counted them
        return;
    }
    __shared__ double T[CQ_RB * CQ_P];
    __shared__ __attribute__((aligned(16))) double GR[3 * 1024];      // Gram 32 x 33 | factor 1024 + 32; later the three S matrices of the post step
    __shared__ __attribute__((aligned(16))) double scr[4 * 32 * 33 + 64];
    __shared__ double red[4];
    __shared__ int s_out[4];
    __shared__ int s_flag;
    double* Gs = GR;
    double* Rf = GR + 32 * CQ_P;                            // 1056 doubles in: 16-byte aligned
    const int tid = threadIdx.x, blk = blockIdx.x;
    const bool writer = (blk == 0);
    int64_t r0;
    int nr;
    cq_block_rows(nrows, nblk, blk, r0, nr);
    int nbar = 0;
    bool alive = true;
    if (tid == 0) s_flag = cq_ldi(&stt->timeout) ? 0 : 1;
    __syncthreads();
    alive = s_flag != 0;
    __syncthreads();
    if (alive) {
        FQ_CLK(0);
        if (writer && tid == 0) { stt->done = 0; stt->pass = 0; stt->ndefer_total = 0; stt->nrefill_total = 0; stt->fallback = 0; }
        const int ex = cq_load_scaled_tile(X, xrs, xcs, r0, nr, b, T, red, tid);
        FQ_CLK(1);
        cq_block_gram(T, scr, part + (int64_t)blk * CQ_PART, tid);
        if (tid == 0) cq_sti(bexp + blk, ex);
        FQ_CLK(2);
        alive = cq_grid_barrier(&stt->fcounter, base + (++nbar) * nblk, &s_flag, tid, spin_limit);
        FQ_CLK(3);
        int dec = 0, tlast = 0;
        const bool sliced = nblk > CQ_SLICE_FROM;
        int emax0 = 0;
        if (alive && sliced) {
            emax0 = cq_slice_reduce(part, bexp, nblk, blk, gsum, tid);
            alive = cq_grid_barrier(&stt->fcounter, base + (++nbar) * nblk, &s_flag, tid, spin_limit);
        }
        if (alive) {
            if (sliced) cq_tail_fused(gsum, nullptr, 1, b, 0, stt, writer, Gs, Rf, s_out, maxpass, tid, (long long)nrows * b, slot, emax0);
            else cq_tail_fused(part, bexp, nblk, b, 0, stt, writer, Gs, Rf, s_out, maxpass, tid, (long long)nrows * b, slot);
            FQ_CLK(4);
            dec = s_out[0];
            const int emax = s_out[3];
            const double scl0 = (ex > -2000 && emax > -2000) ? ldexp(1.0, ex - emax) : 0.0;     // tile is 2^-ex X; the passes work on 2^-emax X
            for (int t = 1; dec == 0; ++t) {
                const int fin = s_out[1];
                const unsigned deadmask = (unsigned)s_out[2];
                __syncthreads();                             // s_out is rewritten by the next tail
                cq_substitute(T, Rf, tid, deadmask, seed + 0x9E3779B97F4A7C15ULL * (uint64_t)t, r0, nr, t == 1 ? scl0 : 1.0);
                __syncthreads();
                if (t == 1) FQ_CLK(5);
                double* pt = part + (int64_t)(t & 1) * nblk * CQ_PART;
                if (!fin) cq_block_gram(T, scr, pt + (int64_t)blk * CQ_PART, tid);
                if (writer && lu_all) {
                    double* tb = topblk + (t & 1) * 1024;
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const int e = tid + 256 * u; cq_st(tb + e, T[(e >> 5) * CQ_P + (e & 31)]); }
                }
                if (writer && tid == 0) stt->pass = t;
                if (t == 1) FQ_CLK(6);
                alive = cq_grid_barrier(&stt->fcounter, base + (++nbar) * nblk, &s_flag, tid, spin_limit);
                if (!alive) break;
                if (t == 1) FQ_CLK(7);
                tlast = t;
                if (fin) {
                    dec = 1;
                    if (writer && tid == 0) {
                        stt->done = 1; stt->final_next = 0; stt->dead = 0u;
                        cq_count(slot, t, stt->ndefer_total, stt->nrefill_total, false, (long long)nrows * b, true);
                    }
                    break;
                }
                if (sliced) {
                    double* gs = gsum + (t & 1) * CQ_PART;
                    (void)cq_slice_reduce(pt, nullptr, nblk, blk, gs, tid);
                    alive = cq_grid_barrier(&stt->fcounter, base + (++nbar) * nblk, &s_flag, tid, spin_limit);
                    if (!alive) break;
                    cq_tail_fused(gs, nullptr, 1, b, t, stt, writer, Gs, Rf, s_out, maxpass, tid, (long long)nrows * b, slot);
                } else cq_tail_fused(pt, nullptr, nblk, b, t, stt, writer, Gs, Rf, s_out, maxpass, tid, (long long)nrows * b, slot);
                dec = s_out[0];
            }
        }
        FQ_CLK(8);
        if (alive && dec == 1) {
            if (lu_all == nullptr) {
                cq_store_tile(T, Y, rs, cs, r0, nr, b, tid);
            } else {
                double* lu = lu_all + (int64_t)blk * CQ_LU_DOUBLES;
                // (the Gram matrix / factor in GR are done with: the three S matrices of the post step go there directly; the top blocks,
                //  which only the workgroup with the first rows stores, travel through its own slice of lu_all)
                cq_lu(topblk + (tlast & 1) * 1024, true, b, lu, writer ? Tp : nullptr, scr, tid, GR, r0 < b, Wq != nullptr);
                __syncthreads();                             // (waits for this workgroup's stores to lu as well)
                FQ_CLK(9);
                cq_post_tile(blk, nblk, nrows, b, Y, rs, cs, W, wrs, wcs, Wq, lu, T, GR, tid, true, true);
                FQ_CLK(10);
            }
        } else if (alive && dec == 2 && writer) {
            // out of passes (never seen on the contraction path): workgroup 0 redoes the panel with Householder reflections from
            // the untouched input and post-processes every tile itself, exactly as cq_post_kernel does
            const int emax = s_out[3];
            __syncthreads();
            cq_fallback_householder(X, xrs, xcs, Y, rs, cs, nrows, b, emax, scr, tid);
            if (lu_all) {
                double* lu = lu_all;
                __threadfence();
                __syncthreads();
                for (int e = tid; e < 1024; e += 256) {
                    const int i = e >> 5, j = e & 31;
                    lu[CQ_LU_YTOP + e] = (i < b && j < b && i < nrows) ? cq_ld(Y + (int64_t)i * rs + j * cs) : 0.0;
                }
                __threadfence();
                __syncthreads();
                cq_lu(lu + CQ_LU_YTOP, true, b, lu, Tp, scr, tid);
                __threadfence();
                __syncthreads();
                for (int t = 0; t < nblk; ++t) {
                    cq_post_tile(t, nblk, nrows, b, Y, rs, cs, W, wrs, wcs, Wq, lu, T, GR, tid);
                    __threadfence();
                    __syncthreads();
                }
            }
        }
    }
    if (!alive) {                                            // a barrier gave up (or an earlier launch of this call did): poison the output
        if (tid == 0) {
            cq_sti(&stt->timeout, 1);
            // sticky per-stream count (survives the clearing of the state block at the end of the call): the host compares it
            // with the value it saw last (fused_timeouts) and redoes the work with the six-launch chain
            atomicAdd(&cq_stats[slot * 16 + 10], 1ull);
        }
        const double bad = __longlong_as_double(0x7ff8000000000000LL);
        for (int e = tid; e < nr * b; e += 256) {
            const int i = e / b, j = e % b;
            Y[(r0 + i) * rs + j * cs] = bad;
            if (W) W[(r0 + i) * wrs + j * wcs] = bad;
        }
    }
    // every launch books (maxpass + 1) * nblk arrivals (twice that with the sliced reduction), however many barriers it took
    const int nbar_booked = (maxpass + 1) * (nblk > CQ_SLICE_FROM ? 2 : 1);
    if (tid == 0 && nbar < nbar_booked) atomicAdd(&stt->fcounter, nbar_booked - nbar);
}

// ---- host driver ---------------------------------------------------------------------------------------------------
// layout: state | R (1024 + 32) | partial Gram matrices (x 2 for the single-launch form) | block exponents |
//         reconstruction buffers (one per workgroup for the single-launch form) | top block of the panel (x 2)
// workspace layout: as for the single-launch form up to 2 x CQ_FUSED_MAXBLK workgroups (whether a launch of more than CQ_FUSED_MAXBLK
// takes that form is decided per call, see cq_big_admit)
static inline bool cq_fused_fits(int64_t nblk) { return nblk <= 2 * CQ_FUSED_MAXBLK; }
int64_t cholqr_ws_bytes(int64_t nrows, int b) {
    (void)b;
    const int64_t nblk = cdiv(nrows, CQ_RB);
    const int64_t nlu = cq_fused_fits(nblk) ? nblk : 1, npart = cq_fused_fits(nblk) ? 2 * nblk : nblk;
    return CQ_STATE_BYTES + align_up((1024 + 32) * 8, 256) + align_up(npart * CQ_PART * 8, 256) + align_up(nblk * 4, 256) +
           align_up(nlu * CQ_LU_DOUBLES * 8, 256) + align_up(2 * 1024 * 8, 256) + align_up(2 * CQ_PART * 8, 256) + 256;
}

// TN_PANEL_FUSED=0 keeps the six-launch chain for every panel (A/B measurements, cross-checks); read per call: the tests switch it
static bool cq_fused_enabled() {
    const char* e = getenv("TN_PANEL_FUSED");
    return !(e && e[0] == '0');
}

// statistics slot of a stream (first come, first served; the streams beyond CQ_STAT_SLOTS share the last slot)
static std::mutex cq_slot_mu;
static std::map<hipStream_t, int> cq_slot_of;
static std::vector<int> cq_slot_free;                               // slots of destroyed streams (tn_stream_destroy), handed out again first
static int cq_slot_next = 0;
static int cq_stat_slot(hipStream_t st) {
    std::lock_guard<std::mutex> lk(cq_slot_mu);
    auto it = cq_slot_of.find(st);
    if (it != cq_slot_of.end()) return it->second;
    int s = CQ_STAT_SLOTS;
    if (!cq_slot_free.empty()) { s = cq_slot_free.back(); cq_slot_free.pop_back(); }
    else if (cq_slot_next < CQ_STAT_SLOTS) s = cq_slot_next++;
    cq_slot_of.emplace(st, s);
    return s;
}
int cholqr_stream_slot(hipStream_t st) { return cq_stat_slot(st); }      // (smallqr.hip shares the numbering)

// Co-residency budget of the launches with in-kernel barriers (cq_fused_kernel here, sq_kernel in smallqr.hip): a workgroup of them
// needs a whole CU and waits only for workgroups of its own launch, so launches in flight cannot deadlock while together they ask
// for no more CUs than this process may count on.  Nothing about that is assumed: the budget is derived at first use from
//   * the device (hipDeviceAttributeMultiprocessorCount), or TN_PANEL_CU_BUDGET when several processes share the card (the CUs this
//     process may count on: half the chip for two tenants ...; 0 keeps every panel on the six-launch chain),
//   * the number of hardware queues the runtime multiplexes the streams onto (GPU_MAX_HW_QUEUES as the runtime itself reads it at
//     initialisation; 4 when unset): at most that many kernels of the process are in flight,
// which gives  maxblk = min(32, budget / queues)  workgroups for an ordinary launch (32 with the package's 8 queues on an MI355X).  A
// taller panel (up to 2 maxblk workgroups) is admitted only while the budget still holds with it:
//     2 maxblk B + maxblk (S - B) <= budget,   S = min(streams of this process that have run panels, queues),  B = tall launches in
// flight (this one included): all four chains of a solve when nothing else runs panels, three with a fifth stream around, none with
// eight.  In-flight tall launches are tracked with one event per stream (recorded behind the launch, queried before the next
// admission, all under one mutex that also covers the launch itself).  Streams created with a CU mask (tn_stream_create_masked) and
// streams on which a launch has ever given up at a barrier (fused_timeouts) are taken off these forms for good.  A panel that is not
// admitted takes the six-launch chain: the result is the same bit for bit.  TN_PANEL_FUSED_BIG=0: never admit tall panels.
struct FusedBudget { int cus = 0, queues = 4, maxblk = 0; bool ready = false; };
static FusedBudget cq_budget_of[16];
static std::mutex cq_budget_mu;
static FusedBudget fused_budget() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { (void)hipGetLastError(); return FusedBudget(); }
    std::lock_guard<std::mutex> lk(cq_budget_mu);
    FusedBudget& b = cq_budget_of[dev];
    if (!b.ready) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); cus = 0; }
        if (const char* e = getenv("TN_PANEL_CU_BUDGET")) { const int v = atoi(e); if (v >= 0 && v < cus) cus = v; }
        int q = 4;
        if (const char* e = getenv("GPU_MAX_HW_QUEUES")) { const int v = atoi(e); if (v >= 1) q = v; }
        b.cus = cus; b.queues = q;
        b.maxblk = cus / q < CQ_FUSED_MAXBLK ? cus / q : CQ_FUSED_MAXBLK;
        b.ready = true;
    }
    return b;
}
static std::map<hipStream_t, bool> cq_stream_off;                  // (guarded by cq_slot_mu)
void fused_forms_disable(hipStream_t st) {
    std::lock_guard<std::mutex> lk(cq_slot_mu);
    cq_stream_off[st] = true;
}
static void cq_dirty_mark(int slot);
// tn_stream_destroy: the stream's slot (statistics, panel state, admission bookkeeping) goes back to the pool, so that the count of
// live streams the admission of tall panels works with stays a count of LIVE streams
void fused_stream_released(hipStream_t st) {
    std::lock_guard<std::mutex> lk(cq_slot_mu);
    cq_stream_off.erase(st);
    auto it = cq_slot_of.find(st);
    if (it == cq_slot_of.end()) return;
    const int slot = it->second;
    cq_slot_of.erase(it);
    if (slot < CQ_STAT_SLOTS) { cq_slot_free.push_back(slot); cq_dirty_mark(slot); }
}
static bool cq_stream_is_off(hipStream_t st) {
    std::lock_guard<std::mutex> lk(cq_slot_mu);
    return cq_stream_off.count(st) != 0;
}
// may a launch of nwg workgroups with in-kernel barriers go out on this stream?
bool fused_forms_allowed(hipStream_t st, int nwg) {
    if (!cq_fused_enabled() || cq_stream_is_off(st)) return false;
    if (cq_stat_slot(st) >= CQ_STAT_SLOTS) return false;
    return nwg <= fused_budget().maxblk;
}

struct CqBigTrack {
    hipEvent_t ev[CQ_STAT_SLOTS + 1] = {};
    bool pending[CQ_STAT_SLOTS + 1] = {};
};
static CqBigTrack cq_big;
static std::mutex cq_big_mu;
static bool cq_big_enabled() {                                      // read per call: the tests switch it
    const char* e = getenv("TN_PANEL_FUSED_BIG");
    return !(e && e[0] == '0');
}
// call with cq_big_mu held
static bool cq_big_admit(int slot, int nslots) {
    if (slot >= CQ_STAT_SLOTS) return false;                       // streams without a slot of their own are not tracked
    const FusedBudget b = fused_budget();
    if (b.maxblk < 1) return false;
    int inflight = 0;
    for (int s = 0; s < CQ_STAT_SLOTS; ++s) {
        if (s == slot || !cq_big.pending[s]) continue;            // (an earlier tall launch of THIS stream is not concurrent with the new one)
        if (hipEventQuery(cq_big.ev[s]) == hipSuccess) cq_big.pending[s] = false;
        else ++inflight;
    }
    const int S = nslots < b.queues ? nslots : b.queues;
    return inflight + 1 <= b.cus / b.maxblk - S;
}
static void cq_big_launched(hipStream_t st, int slot) {
    if (!cq_big.ev[slot] && hipEventCreateWithFlags(&cq_big.ev[slot], hipEventDisableTiming) != hipSuccess) { cq_big.ev[slot] = nullptr; return; }
    if (hipEventRecord(cq_big.ev[slot], st) == hipSuccess) cq_big.pending[slot] = true;
}

// ---- time-outs of the launches with in-kernel barriers ----------------------------------------------------------------------
// A launch that gives up at a barrier poisons its outputs with NaN and adds to a sticky per-stream device counter (cq_stats[10]
// here, sq_stats[3] in smallqr.hip).  Every caller that has enqueued such launches asks fused_timeouts before it hands results
// back: one 16-byte read-back and a synchronisation.  A positive answer means: the results of the stream since the previous
// check are invalid, the stream has been taken off the single-launch forms (the co-residency the spins rely on evidently does
// not hold: another tenant on the card, a debugger, ...), its barrier state is cleared, and the caller must redo the work -- which
// now takes the six-launch chain / the blocked path, bit-identical results.  Callers that own many factorisations (tn_compress_mps)
// defer the check to the end of their call (FusedDeferCheck).
int smallqr_stats(hipStream_t st, unsigned long long* out4, int reset);
int smallqr_reset_state(hipStream_t st);
static thread_local long cq_fused_launches = 0;                    // launches with in-kernel barriers enqueued by this thread since its last check
static thread_local int cq_defer_depth = 0;
void fused_note_launch() { ++cq_fused_launches; }
void fused_defer_push() { ++cq_defer_depth; }
void fused_defer_pop() { --cq_defer_depth; }
bool fused_check_deferred() { return cq_defer_depth > 0; }
bool fused_check_needed() { return cq_fused_launches > 0; }
static unsigned long long cq_timeouts_seen[CQ_STAT_SLOTS + 1][2];  // last values of the two counters per slot (guarded by cq_slot_mu)
int fused_timeouts(hipStream_t st, int* count_out) {
    *count_out = 0;
    cq_fused_launches = 0;
    const int slot = cq_stat_slot(st);
    if (slot >= CQ_STAT_SLOTS) return 0;                           // such streams never take these forms
    unsigned long long* h = (unsigned long long*)pinned_host(64, 7);
    unsigned long long tmp[8];
    if (!h) h = tmp;
    hipError_t e = hipMemcpyFromSymbolAsync(h, HIP_SYMBOL(cq_stats), 8, ((size_t)slot * 16 + 10) * 8, hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return hip_fail(e, "read panel time-outs");
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return hip_fail(e, "sync panel time-outs");
    const unsigned long long a = h[0];
    unsigned long long sq[4];
    int rc = smallqr_stats(st, sq, 0);
    if (rc) return rc;
    unsigned long long da, db;
    {
        std::lock_guard<std::mutex> lk(cq_slot_mu);
        da = a - cq_timeouts_seen[slot][0];
        db = sq[3] - cq_timeouts_seen[slot][1];
        cq_timeouts_seen[slot][0] = a;
        cq_timeouts_seen[slot][1] = sq[3];
        if (da + db > 0) cq_stream_off[st] = true;
    }
    if (da + db == 0) return 0;
    *count_out = (int)(da + db > 2147483647ull ? 2147483647ull : da + db);
    fprintf(stderr, "[libtnpeps] %llu launch(es) with in-kernel barriers gave up on stream %p (workgroups not co-resident: is the device shared? "
            "see TN_PANEL_CU_BUDGET); the work is redone through the multi-launch forms, which this stream uses from now on\n",
            (unsigned long long)(da + db), (void*)st);
    // leave a clean slate: the stream's panel state (sticky flag, barrier counter) and the small-QR barrier state
    {
        std::lock_guard<std::mutex> lk(cq_slot_mu);
        cq_dirty_mark(slot);
    }
    if ((rc = smallqr_reset_state(st))) return rc;
    return 0;
}

// The panel state of a factorisation lives in a block of its own per stream (a stream's calls do not overlap), NOT in the shared
// scratch: it is zero when a call starts because the call before it ended by clearing it (the first launch after the last panel --
// diag_qr_kernel of tn_qr -- zeroes the block), so no memset launch per factorisation.  A call that fails half-way leaves its
// stream's block marked dirty on the host and the next call clears it with a memset.  Streams beyond CQ_STAT_SLOTS fall back to the
// state block at the head of the workspace and a memset per call.
__device__ char cq_state_pool[CQ_STAT_SLOTS * CQ_STATE_BYTES];
static bool cq_dirty[CQ_STAT_SLOTS];
static void cq_dirty_mark(int slot) { cq_dirty[slot] = true; }      // the next cholqr_begin on this stream clears the block with a memset
int cholqr_begin(hipStream_t st, void* ws, void** state_out) {
    const int slot = cq_stat_slot(st);
    if (slot >= CQ_STAT_SLOTS) {
        *state_out = ws;
        const hipError_t e = hipMemsetAsync(ws, 0, CQ_STATE_BYTES, st);
        return e == hipSuccess ? 0 : hip_fail(e, "memset panel state");
    }
    static char* bases[16] = {};                                   // a __device__ symbol has one address per device
    static std::mutex mu;
    bool dirty;
    char* base = nullptr;
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { set_error("cholqr_begin: no current device"); return 1; }
        std::lock_guard<std::mutex> lk(mu);
        if (!bases[dev]) {
            void* p = nullptr;
            const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(cq_state_pool));
            if (e != hipSuccess) return hip_fail(e, "panel state pool");
            bases[dev] = (char*)p;
        }
        base = bases[dev];
        dirty = cq_dirty[slot];
        cq_dirty[slot] = true;
    }
    *state_out = base + (size_t)slot * CQ_STATE_BYTES;
    if (dirty) {
        const hipError_t e = hipMemsetAsync(*state_out, 0, CQ_STATE_BYTES, st);
        if (e != hipSuccess) return hip_fail(e, "memset panel state");
    }
    return 0;
}
// the launch that clears the state block has been enqueued: the next call on this stream finds it clean
void cholqr_end_ok(hipStream_t st) {
    const int slot = cq_stat_slot(st);
    if (slot < CQ_STAT_SLOTS) cq_dirty[slot] = false;
}

// The state block at the head of the workspace must be zero before the first panel of a call (the kernels leave it clean).
int cholqr_reset(hipStream_t st, void* ws) {
    const hipError_t e = hipMemsetAsync(ws, 0, CQ_STATE_BYTES, st);
    return e == hipSuccess ? 0 : hip_fail(e, "memset panel state");
}

// TN_PANEL_CAPTURE=<directory> (diagnostics; tools/capture_panels.py): after every panel the stream is synchronised, and the input
// of a panel that needed at least TN_PANEL_CAPTURE_MIN (4) substitution passes and has at most TN_PANEL_CAPTURE_MAXROWS (4096) rows is
// written to <directory>/panel_<seq>_<rows>x<b>_p<passes>.f64 (row-major doubles) -- the hard panels of a real sweep, kept as
// regression inputs under tests/golden/.
__global__ __launch_bounds__(256) void cq_capture_kernel(const double* X, int64_t rs, int64_t cs, int64_t nrows, int b, double* out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < nrows * b) out[e] = X[(e / b) * rs + (e % b) * cs];
}
static void cq_capture(hipStream_t st, const double* X, int64_t irs, int64_t ics, int64_t nrows, int b, const void* ws) {
    static const char* dir = getenv("TN_PANEL_CAPTURE");
    if (!dir || !dir[0]) return;
    static const int min_pass = [] { const char* e = getenv("TN_PANEL_CAPTURE_MIN"); return e ? atoi(e) : 4; }();
    static const int64_t max_rows = [] { const char* e = getenv("TN_PANEL_CAPTURE_MAXROWS"); return e ? atoll(e) : 4096ll; }();
    static std::mutex mu;
    static int seq = 0;
    CqState h;
    if (hipStreamSynchronize(st) != hipSuccess || hipMemcpy(&h, ws, sizeof(CqState), hipMemcpyDeviceToHost) != hipSuccess) return;
    if (h.pass < min_pass || nrows > max_rows) return;
    double* d = nullptr;
    if (hipMalloc(&d, (size_t)nrows * b * 8) != hipSuccess) return;
    hipLaunchKernelGGL(cq_capture_kernel, dim3((unsigned)cdiv(nrows * b, 256)), dim3(256), 0, st, X, irs, ics, nrows, b, d);
    std::vector<double> host((size_t)nrows * b);
    if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(host.data(), d, host.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
        std::lock_guard<std::mutex> lk(mu);
        char path[1024];
        snprintf(path, sizeof(path), "%s/panel_%05d_%lldx%d_p%d.f64", dir, seq++, (long long)nrows, b, h.pass);
        if (FILE* f = fopen(path, "wb")) { fwrite(host.data(), 8, host.size(), f); fclose(f); }
    }
    (void)hipFree(d);
}

// One panel step.  Y (nrows x b, strides rs/cs) receives an orthonormal basis Q1 of the column space of the panel X (read
// only; must not overlap Y).  With reconstruct != 0 the launches go on to the Householder reconstruction (Ballard et al.):
// on return Y holds the unit lower trapezoidal reflectors, Tp (b x b, pitch b) their T factor, W = Y T^T (nrows x b, strides
// wrs/wcs) and, when Wq != NULL, Wq = Y T (strides of Y), i.e. what lu_reconstruct_kernel + rows_times_small3 of qr.hip produce,
// in the launch slots that would otherwise return at once.
int cholqr_panel(hipStream_t st, const double* X, int64_t irs, int64_t ics, double* Y, int64_t rs, int64_t cs, int64_t nrows, int b, void* ws,
                 int64_t ws_bytes, uint64_t seed, int reconstruct, double* Tp, double* W, int64_t wrs, int64_t wcs, double* Wq, int* fused_base,
                 void* state, const int* active) {
    TN_CHECK_ARG(b >= 1 && b <= 32, "panel width must be <= 32");
    TN_CHECK_ARG(nrows >= b, "panel must have at least b rows");
    TN_CHECK_ARG(ws_bytes >= cholqr_ws_bytes(nrows, b), "workspace too small");
    TN_CHECK_ARG(X != Y, "panel and basis must not alias");
    TN_CHECK_ARG(!reconstruct || (Tp && W), "reconstruction needs T and W");
    const int nblk = (int)cdiv(nrows, CQ_RB);
    char* p = (char*)ws;
    CqState* stt = (CqState*)(state ? state : ws); p += CQ_STATE_BYTES;       // (state: the stream's own block, see cholqr_begin)
    double* Rg = (double*)p; p += align_up((1024 + 32) * 8, 256);
    const bool fits = cq_fused_fits(nblk);
    double* part = (double*)p; p += align_up((int64_t)(fits ? 2 * nblk : nblk) * CQ_PART * 8, 256);
    int* bexp = (int*)p; p += align_up((int64_t)nblk * 4, 256);
    double* lu = reconstruct ? (double*)p : nullptr; p += align_up((int64_t)(fits ? nblk : 1) * CQ_LU_DOUBLES * 8, 256);
    double* topblk = (double*)p; p += align_up(2 * 1024 * 8, 256);
    double* gsum = (double*)p;                                      // 2 x CQ_PART: the summed Gram matrix of the sliced reduction (pass parity)
    // TN_PANEL_MAXPASS (1 .. CQ_MAXPASS): fewer substitution passes, to drive the Householder fallback in tests
    static const int maxpass = [] { const char* e = getenv("TN_PANEL_MAXPASS"); const int v = e ? atoi(e) : CQ_MAXPASS; return v >= 1 && v <= CQ_MAXPASS ? v : CQ_MAXPASS; }();
    const int slot = cq_stat_slot(st);
    const int maxblk = fused_budget().maxblk;                      // ordinary single-launch panels: at most this many workgroups
    const bool tall = nblk > maxblk;
    const bool usable = fits && fused_base && nblk <= 2 * maxblk && fused_forms_allowed(st, 1);
    std::unique_lock<std::mutex> big_lock(cq_big_mu, std::defer_lock);
    bool admitted = usable;
    if (usable && tall) {
        admitted = false;
        if (cq_big_enabled()) {
            int nslots;
            { std::lock_guard<std::mutex> lk(cq_slot_mu); nslots = (int)cq_slot_of.size(); }
            big_lock.lock();
            admitted = cq_big_admit(slot, nslots);
            if (!admitted) big_lock.unlock();
        }
    }
    if (admitted) {
        static const unsigned spin_limit = [] { const char* e = getenv("TN_PANEL_SPIN_LIMIT"); return e ? (unsigned)strtoul(e, nullptr, 10) : CQ_SPIN_LIMIT; }();
        fused_note_launch();
        // one launch for the whole chain.  Algorithmic bytes: the panel in, the reflectors (and W, Wq) out -- the tile never
        // leaves LDS in between; flops: Gram + post at launch time, the passes are booked from the device counter (cq_stats[3])
        prof_begin(st, PROF_TSQR);
        hipLaunchKernelGGL(cq_fused_kernel, dim3(nblk), dim3(256), 0, st, X, irs, ics, Y, rs, cs, nrows, b, nblk, part, bexp, topblk, stt,
                           *fused_base, seed, lu, Tp, W, wrs, wcs, Wq, maxpass, slot, spin_limit, gsum, active);
        TN_CHECK_LAUNCH("cq_fused_kernel");
        if (tall) { cq_big_launched(st, slot); big_lock.unlock(); }
        *fused_base += (maxpass + 1) * nblk * (nblk > CQ_SLICE_FROM ? 2 : 1);
        const double e = (double)nrows * b;
        prof_end(st, PROF_TSQR, (2.0 + (reconstruct ? (Wq ? 6.0 : 4.0) : 0.0)) * e * b, (reconstruct ? (Wq ? 32.0 : 24.0) : 16.0) * e);
        cq_capture(st, X, irs, ics, nrows, b, stt);
        return 0;
    }
    prof_begin(st, PROF_TSQR);
    hipLaunchKernelGGL(cq_gram_kernel, dim3(nblk), dim3(256), 0, st, X, irs, ics, nrows, b, nblk, part, bexp, stt, Rg, slot, active);
    TN_CHECK_LAUNCH("cq_gram_kernel");
    prof_end(st, PROF_TSQR, 2.0 * nrows * b * b, 8.0 * nrows * b);
    for (int t = 1; t <= maxpass; ++t) {
        prof_begin(st, PROF_TSQR);
        hipLaunchKernelGGL(cq_pass_kernel, dim3(nblk), dim3(256), 0, st, X, irs, ics, Y, rs, cs, nrows, b, nblk, t == 1 ? 1 : 0, t, part, stt,
                           Rg, seed + 0x9E3779B97F4A7C15ULL * (uint64_t)t, lu, Tp, maxpass, slot, active);
        TN_CHECK_LAUNCH("cq_pass_kernel");
        // (the passes that really run are booked from the device counter cq_stats[7] by the caller of tn_panel_stats:
        //  3 n b^2 flops and 16 n b bytes per applied pass; a launch that finds the panel converged moves nothing)
        prof_end(st, PROF_TSQR, 0.0, 0.0);
    }
    prof_begin(st, PROF_TSQR);
    hipLaunchKernelGGL(cq_post_kernel, dim3(nblk), dim3(256), 0, st, X, irs, ics, Y, rs, cs, nrows, b, nblk, stt, lu, Tp, W, wrs, wcs, Wq, active);
    TN_CHECK_LAUNCH("cq_post_kernel");
    prof_end(st, PROF_TSQR, reconstruct ? (Wq ? 6.0 : 4.0) * nrows * b * b : 0.0, reconstruct ? (Wq ? 32.0 : 24.0) * nrows * b : 0.0);
    cq_capture(st, X, irs, ics, nrows, b, stt);
    return 0;
}

int cholqr_orthonormalize(hipStream_t st, const double* X, int64_t irs, int64_t ics, double* Y, int64_t rs, int64_t cs, int64_t nrows,
                          int b, void* ws, int64_t ws_bytes, uint64_t seed, int* fused_base, void* state) {
    return cholqr_panel(st, X, irs, ics, Y, rs, cs, nrows, b, ws, ws_bytes, seed, 0, nullptr, nullptr, 0, 0, nullptr, fused_base, state, nullptr);
}

// diagnostics: state block of the last panel (synchronises the stream) and the process-wide counters
int cholqr_debug_state(hipStream_t st, const void* ws, int* ints9, double* dev_hist) {
    CqState h;
    hipError_t e = hipMemcpyAsync(&h, ws, sizeof(CqState), hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return hip_fail(e, "memcpy state");
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return hip_fail(e, "sync state");
    ints9[0] = h.counter; ints9[1] = h.done; ints9[2] = h.final_next; ints9[3] = h.pass; ints9[4] = h.emax; ints9[5] = (int)h.dead;
    ints9[6] = h.ndefer_total; ints9[7] = h.nrefill_total; ints9[8] = h.fallback;
    for (int i = 0; i <= CQ_MAXPASS + 1; ++i) dev_hist[i] = h.dev_hist[i];
    return 0;
}
// st_or_null: the counters of that stream only (as far as it has a slot of its own); NULL with all_streams: the sum over all slots
int cholqr_stats(unsigned long long* out16, int reset, hipStream_t st_or_null, int all_streams) {
    static unsigned long long raw[(CQ_STAT_SLOTS + 1) * 16];
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    hipError_t e = hipMemcpyFromSymbol(raw, HIP_SYMBOL(cq_stats), sizeof(raw));
    if (e != hipSuccess) return hip_fail(e, "read panel statistics");
    for (int i = 0; i < 16; ++i) out16[i] = 0;
    if (all_streams) {
        for (int s = 0; s <= CQ_STAT_SLOTS; ++s)
            for (int i = 0; i < 16; ++i) out16[i] += raw[s * 16 + i];
        // (word [10], the sticky count of launches that gave up at a barrier, is NOT a statistic: fused_timeouts compares it with the value
        //  it saw last, so a reset keeps it -- zeroing it would turn the next comparison into a huge unsigned difference, or lose a time-out)
        if (reset) {
            for (int s = 0; s <= CQ_STAT_SLOTS; ++s)
                for (int i = 0; i < 16; ++i) if (i != 10) raw[s * 16 + i] = 0;
            if ((e = hipMemcpyToSymbol(HIP_SYMBOL(cq_stats), raw, sizeof(raw))) != hipSuccess) return hip_fail(e, "reset panel statistics");
        }
        return 0;
    }
    const int slot = cq_stat_slot(st_or_null);
    for (int i = 0; i < 16; ++i) out16[i] = raw[slot * 16 + i];
    if (reset) {
        unsigned long long z[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        z[10] = raw[slot * 16 + 10];
        if ((e = hipMemcpyToSymbol(HIP_SYMBOL(cq_stats), z, sizeof(z), (size_t)slot * sizeof(z))) != hipSuccess) return hip_fail(e, "reset panel statistics");
    }
    return 0;
}

}  // namespace tn
