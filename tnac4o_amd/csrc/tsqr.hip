// Panel factorisation for the blocked QR (K3): communication-avoiding TSQR with Householder reflectors.
//
// A tall panel (nrows x b, b <= 32) is cut into row blocks of <= 256 rows, one workgroup each; every block is
// reduced to its b x b triangle by Householder reflections held in LDS (tsqr_factor_kernel); the stacked triangles
// form the next, 8x shorter level, until a single block remains (3 levels for 16384 rows).  Walking the levels back
// down (tsqr_apply_kernel) applies the stored reflectors to [I; 0] and yields the explicit orthonormal panel basis
// Q1, exactly orthonormal to rounding for ANY input (zero or dependent columns just give tau = 0 reflectors).
// Compared with the Gram/Jacobi panel step this needs 2 launches per level instead of ~19 per panel and no iteration.
#include "common.h"

namespace tn {

constexpr int TS_RB = 256;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
// 256-thread block reductions through a 4-entry LDS scratch (two barriers)
__device__ __forceinline__ double block_sum4(double v, double* s4) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
    __syncthreads();
    const double r = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    __syncthreads();
    return r;
}
__device__ __forceinline__ double block_max4(double v, double* s4) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
    __syncthreads();
    const double r = fmax(fmax(s4[0], s4[1]), fmax(s4[2], s4[3]));
    __syncthreads();
    return r;
}

__device__ __forceinline__ void block_rows(int64_t nrows, int nblk, int blk, int64_t& r0, int& nr) {
    const int64_t base = nrows / nblk, rem = nrows % nblk;
    r0 = blk * base + (blk < rem ? blk : rem);
    nr = (int)(base + (blk < rem ? 1 : 0));
}

// Block layout of tsqr_factor_kernel: the first 32 rows of a block (the "head", where the triangle R and the pivots
// live) stay in an LDS tile H; the remaining <= 224 rows (the "body") live in REGISTERS: thread (c, g) =
// (tid & 31, tid >> 5) owns body rows 32 + 28 g ... 32 + 28 g + 27 of column c.  LDS carries only what crosses threads:
// the head, the body of the current column (vbuf) and 8 partial sums per column (part).

// Householder QR of one row block.  Per column j a single pass forms  s_c = sum_{r>j} a_rj a_rc  for every c; the
// reflector then follows without any further reduction:  beta = -sign(alpha) sqrt(alpha^2 + s_j),
// v = [1; a_j / (alpha - beta)],  tau = 1 + |alpha| / sqrt(alpha^2 + s_j),  v^T a_c = a_jc + s_c / (alpha - beta)
// (no cancellation: column-wise backward stable).  Two barriers per column, no cross-lane shuffles, no dynamic register
// indexing.  The block is pre-scaled by a power of two so that squares neither overflow nor lose entries above 1e-145 of
// the block maximum.  X (input) and Vout (reflectors + triangle, LAPACK layout) may be the same array.
constexpr int TS_BODY = 28;

__global__ __launch_bounds__(256) void tsqr_factor_kernel(const double* X, int64_t rs, int64_t cs, double* Vout, int64_t ors,
                                                          int64_t ocs, int64_t nrows, int b, int nblk,
                                                          double* __restrict__ taus, double* __restrict__ Rout) {
    constexpr int P = 33;
    __shared__ double T[TS_RB * P];          // staging for coalesced global loads / stores; rows 0..31 double as the head H
    __shared__ double part[256];
    __shared__ double vbuf[TS_RB];
    __shared__ double dinv[32];
    const int tid = threadIdx.x, blk = blockIdx.x;
    int64_t r0;
    int nr;
    block_rows(nrows, nblk, blk, r0, nr);
    const bool colfast = (cs == 1);
    for (int e = tid; e < TS_RB * P; e += 256) T[e] = 0.0;
    if (tid < 32) dinv[tid] = 0.0;
    __syncthreads();
    double amax = 0.0;
    for (int e = tid; e < nr * b; e += 256) {
        const int i = colfast ? e / b : e % nr, j = colfast ? e % b : e / nr;
        const double x = X[(r0 + i) * rs + j * cs];
        T[i * P + j] = x;
        amax = fmax(amax, fabs(x));
    }
    part[tid] = amax;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (tid < k) part[tid] = fmax(part[tid], part[tid + k]);
        __syncthreads();
    }
    amax = part[0];
    int ex = 0;
    if (amax > 0.0 && amax < 1.7e308) frexp(amax, &ex);
    const double scl = ldexp(1.0, -ex), iscl = ldexp(1.0, ex);
    const int c = tid & 31, g = tid >> 5, rb = 32 + g * TS_BODY;
    double y[TS_BODY];
#pragma unroll
    for (int k = 0; k < TS_BODY; ++k) y[k] = T[(rb + k) * P + c] * scl;
    __syncthreads();
    for (int e = tid; e < 32 * P; e += 256) T[e] *= scl;          // the head stays in T (rows 0..31)
    const int kmax = b < nr ? b : nr;
    if (c == 0) {
#pragma unroll
        for (int k = 0; k < TS_BODY; ++k) vbuf[rb + k] = y[k];
    }
    for (int j = 0; j < kmax; ++j) {
        __syncthreads();                                    // head updated, body of column j published, part[] free
        double v[TS_BODY];
#pragma unroll
        for (int k = 0; k < TS_BODY; ++k) v[k] = vbuf[rb + k];
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int k = 0; k < TS_BODY; k += 4) {
            s0 += v[k] * y[k]; s1 += v[k + 1] * y[k + 1]; s2 += v[k + 2] * y[k + 2]; s3 += v[k + 3] * y[k + 3];
        }
        // head rows 4g .. 4g+3 (only those below the diagonal)
        double hv[4], hy[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = 4 * g + k;
            hv[k] = (r > j) ? T[r * P + j] : 0.0;
            hy[k] = T[r * P + c];
            s0 += hv[k] * hy[k];
        }
        const double alpha = T[j * P + j], ajc = T[j * P + c];
        part[tid] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        double sc = 0.0, sj = 0.0;                          // sums over the rows below the diagonal
#pragma unroll
        for (int k = 0; k < 8; ++k) { sc += part[k * 32 + c]; sj += part[k * 32 + j]; }
        const double wj = alpha * alpha + sj;
        double tau = 0.0;
        // The block maximum is scaled to [0.5, 1): a squared column norm below 1e-290 is zero or partly subnormal (no
        // longer accurate enough to build an orthogonal reflector).  Such a column is below 1e-145 of the block's
        // largest entry; it is treated as exactly zero from row j down (H = I), which keeps Q orthogonal to rounding.
        if (wj > 1e-290) {
            const double rn = fast_rsqrt(wj), nrm = wj * rn;
            const double beta = -copysign(nrm, alpha), d = alpha - beta, invd = fast_rcp(d);
            tau = 1.0 + fabs(alpha) * rn;
            // column j itself: a_jj -> beta, everything below is kept unscaled (scaled by 1/d when written out)
            const double f = (c > j) ? tau * (ajc + sc * invd) : 0.0;
            const double gf = invd * f;
#pragma unroll
            for (int k = 0; k < TS_BODY; ++k) y[k] = fma(-v[k], gf, y[k]);
            if (c > j) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (4 * g + k > j) T[(4 * g + k) * P + c] = fma(-hv[k], gf, hy[k]);
                if (g == 0) T[j * P + c] = ajc - f;
            }
            if (g == 0 && c == j) { T[j * P + j] = beta; dinv[j] = invd; }
        }
        if (tid == 0) taus[blk * 32 + j] = tau;
        // every read of vbuf for this column happened before the barrier above, so the next column may be published now
        if (c == j + 1) {
#pragma unroll
            for (int k = 0; k < TS_BODY; ++k) vbuf[rb + k] = y[k];
        }
    }
    for (int j = kmax + tid; j < 32; j += 256) taus[blk * 32 + j] = 0.0;
    __syncthreads();
    // assemble the LAPACK-style tile: triangle (with its power-of-two scale) on and above the diagonal, reflectors
    // a_rj / d_j below it (scale free); columns that were skipped (dinv = 0) get zero reflectors
    {
        const double dj = dinv[c];
#pragma unroll
        for (int k = 0; k < TS_BODY; ++k) T[(rb + k) * P + c] = (c < kmax) ? y[k] * dj : y[k] * iscl;
    }
    for (int e = tid; e < 32 * 32; e += 256) {
        const int i = e >> 5, j = e & 31;
        const double x = T[i * P + j];
        T[i * P + j] = (i <= j || j >= kmax) ? x * iscl : x * dinv[j];
    }
    __syncthreads();
    const bool ofast = (ocs == 1);
    for (int e = tid; e < nr * b; e += 256) {
        const int i = ofast ? e / b : e % nr, j = ofast ? e % b : e / nr;
        Vout[(r0 + i) * ors + j * ocs] = T[i * P + j];
    }
    for (int e = tid; e < b * b; e += 256) {
        const int i = e / b, j = e % b;
        Rout[((int64_t)blk * b + i) * b + j] = (i <= j && i < kmax) ? T[i * P + j] : 0.0;
    }
}

// Qout block = H_0 ... H_{kmax-1} [Qin block; 0]   (Qin == nullptr: identity, used at the single-block top level)
__global__ __launch_bounds__(256) void tsqr_apply_kernel(const double* V, int64_t vrs, int64_t vcs, int64_t nrows, int b,
                                                         int nblk, const double* __restrict__ taus,
                                                         const double* __restrict__ Qin, double* Qout, int64_t qrs,
                                                         int64_t qcs) {
    constexpr int P = 33;
    __shared__ double Tv[TS_RB * P];         // reflectors of the block (read only), later the output staging tile
    __shared__ double part[2][256];
    __shared__ double tl[32];
    const int tid = threadIdx.x, blk = blockIdx.x;
    int64_t r0;
    int nr;
    block_rows(nrows, nblk, blk, r0, nr);
    const bool vfast = (vcs == 1), qfast = (qcs == 1);
    for (int e = tid; e < TS_RB * P; e += 256) Tv[e] = 0.0;
    if (tid < 32) tl[tid] = taus[blk * 32 + tid];
    __syncthreads();
    for (int e = tid; e < nr * b; e += 256) {            // reflector j as a full column: 0 above, 1 on, v below the diagonal
        const int i = vfast ? e / b : e % nr, j = vfast ? e % b : e / nr;
        Tv[i * P + j] = (i > j) ? V[(r0 + i) * vrs + j * vcs] : (i == j ? 1.0 : 0.0);
    }
    const int c = tid & 31, g = tid >> 5, rb = g * 32;
    double y[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const int i = rb + k;
        double q = 0.0;
        if (i < b && i < nr && c < b) q = Qin ? Qin[((int64_t)blk * b + i) * b + c] : (i == c ? 1.0 : 0.0);
        y[k] = q;
    }
    __syncthreads();
    const int kmax = b < nr ? b : nr;
    int pb = 0;
    for (int j = kmax - 1; j >= 0; --j) {
        const double tau = tl[j];
        if (tau == 0.0) continue;                            // uniform
        double v[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = Tv[(rb + k) * P + j];
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int k = 0; k < 32; k += 4) {
            s0 += v[k] * y[k]; s1 += v[k + 1] * y[k + 1]; s2 += v[k + 2] * y[k + 2]; s3 += v[k + 3] * y[k + 3];
        }
        part[pb][tid] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        double w = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) w += part[pb][k * 32 + c];
        w *= tau;
#pragma unroll
        for (int k = 0; k < 32; ++k) y[k] = fma(-v[k], w, y[k]);
        pb ^= 1;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; ++k) Tv[(rb + k) * P + c] = y[k];
    __syncthreads();
    for (int e = tid; e < nr * b; e += 256) {
        const int i = qfast ? e / b : e % nr, j = qfast ? e % b : e / nr;
        Qout[(r0 + i) * qrs + j * qcs] = Tv[i * P + j];
    }
}

// ---- host driver -----------------------------------------------------------------------------------------------
constexpr int TS_MAXLEV = 6;

int64_t tsqr_ws_bytes(int64_t nrows, int b) {
    int64_t tot = 0, n = nrows;
    for (int l = 0; l < TS_MAXLEV; ++l) {
        const int64_t nblk = cdiv(n, TS_RB);
        tot += align_up(nblk * 32 * 8, 256) + align_up(nblk * b * b * 8, 256) + align_up(n * b * 8, 256);  // taus, R, Q
        if (nblk == 1) break;
        n = nblk * b;
    }
    return tot + 256;
}

// The nrows x b panel Xin is read once; X (may be Xin) receives an orthonormal basis Q1 of its column space
// (completed arbitrarily where the panel is rank deficient).
int tsqr_orthonormalize(hipStream_t st, const double* Xin, int64_t irs, int64_t ics, double* X, int64_t rs, int64_t cs,
                        int64_t nrows, int b, void* ws, int64_t ws_bytes) {
    TN_CHECK_ARG(b >= 1 && b <= 32, "panel width must be <= 32");
    TN_CHECK_ARG(nrows >= b, "panel must have at least b rows");
    TN_CHECK_ARG(ws_bytes >= tsqr_ws_bytes(nrows, b), "workspace too small");
    struct Lev { double* V; int64_t rs, cs, nrows; int nblk; double *taus, *R, *Q; } lev[TS_MAXLEV];
    char* p = (char*)ws;
    int nl = 0;
    double* cur = X;
    int64_t crs = rs, ccs = cs, n = nrows;
    for (;;) {
        TN_CHECK_ARG(nl < TS_MAXLEV, "panel too tall");
        Lev& L = lev[nl];
        L.V = cur; L.rs = crs; L.cs = ccs; L.nrows = n; L.nblk = (int)cdiv(n, TS_RB);
        L.taus = (double*)p; p += align_up((int64_t)L.nblk * 32 * 8, 256);
        L.R = (double*)p; p += align_up((int64_t)L.nblk * b * b * 8, 256);
        L.Q = (double*)p; p += align_up(n * b * 8, 256);       // explicit Q of this level (level 0 writes into X)
        prof_begin(st, PROF_TSQR);
        if (nl == 0)     // level 0 reads the caller's panel and writes its reflectors into X (fuses the panel copy)
            hipLaunchKernelGGL(tsqr_factor_kernel, dim3(L.nblk), dim3(256), 0, st, Xin, irs, ics, L.V, L.rs, L.cs, L.nrows, b,
                               L.nblk, L.taus, L.R);
        else
            hipLaunchKernelGGL(tsqr_factor_kernel, dim3(L.nblk), dim3(256), 0, st, (const double*)L.V, L.rs, L.cs, L.V, L.rs,
                               L.cs, L.nrows, b, L.nblk, L.taus, L.R);
        TN_CHECK_LAUNCH("tsqr_factor_kernel");
        prof_end(st, PROF_TSQR, 2.0 * L.nrows * b * b, 16.0 * L.nrows * b);
        ++nl;
        if (L.nblk == 1) break;
        cur = L.R; crs = b; ccs = 1; n = (int64_t)L.nblk * b;
    }
    for (int l = nl - 1; l >= 0; --l) {
        Lev& L = lev[l];
        const double* Qin = (l == nl - 1) ? nullptr : lev[l + 1].Q;
        double* Qout = (l == 0) ? X : L.Q;              // level l's output rows are level l-1's Qin blocks
        const int64_t qrs = (l == 0) ? rs : b, qcs = (l == 0) ? cs : 1;
        prof_begin(st, PROF_TSQR);
        hipLaunchKernelGGL(tsqr_apply_kernel, dim3(L.nblk), dim3(256), 0, st, L.V, L.rs, L.cs, L.nrows, b, L.nblk, L.taus, Qin,
                           Qout, qrs, qcs);
        TN_CHECK_LAUNCH("tsqr_apply_kernel");
        prof_end(st, PROF_TSQR, 2.0 * L.nrows * b * b, 16.0 * L.nrows * b);
    }
    return 0;
}

}  // namespace tn
