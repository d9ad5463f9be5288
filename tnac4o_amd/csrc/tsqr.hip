// Panel factorisation for the blocked QR (K3): communication-avoiding TSQR with Householder reflectors.
//
// A tall panel (nrows x b, b <= 32) is cut into row blocks of <= 256 rows, one workgroup each; every block is
// reduced to its b x b triangle by Householder reflections held in LDS (tsqr_factor_kernel); the stacked triangles
// form the next, 8x shorter level, until a single block remains (3 levels for 16384 rows).  Walking the levels back
// down (tsqr_apply_kernel) applies the stored reflectors to [I; 0] and yields the explicit orthonormal panel basis
// Q1, exactly orthonormal to rounding for ANY input (zero or dependent columns just give tau = 0 reflectors).
// Compared with the Gram/Jacobi panel step this needs 2 launches per level instead of ~19 per panel and no iteration.
#include "common.h"

// -DTN_CLOCKS: block 0 / thread 0 of the TSQR kernels records s_memtime at phase boundaries (diagnostics build only)
#ifdef TN_CLOCKS
__device__ long long tn_ts_clk[64];
#define TN_CLK(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) tn_ts_clk[k] = clock64(); } while (0)
extern "C" int tn_debug_clocks(long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(tn_ts_clk), sizeof(long long) * (n < 64 ? n : 64));
}
#else
#define TN_CLK(k) do {} while (0)
#endif

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
    __shared__ __attribute__((aligned(16))) double vbuf[TS_RB];      // accessed as double2 (rb and TS_BODY are even)
    __shared__ double dinv[32];
    const int tid = threadIdx.x, blk = blockIdx.x;
    int64_t r0;
    int nr;
    block_rows(nrows, nblk, blk, r0, nr);
    const bool colfast = (cs == 1);
    TN_CLK(0);
    if (tid < 32) dinv[tid] = 0.0;
    T[tid * P + 32] = 0.0;                               // the pad column
    double amax = 0.0;
    {   // the whole 256 x 32 tile in one memory round trip: all 32 loads of a thread are issued before any is used
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
            T[i * P + j] = xv[u];
            amax = fmax(amax, fabs(xv[u]));
        }
    }
    amax = wave_max(amax);
    if ((tid & 63) == 0) part[tid >> 6] = amax;
    __syncthreads();
    TN_CLK(1);
    amax = fmax(fmax(part[0], part[1]), fmax(part[2], part[3]));
    __syncthreads();                                     // part[] is reused by the column loop
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
        for (int k = 0; k < TS_BODY; k += 2) *reinterpret_cast<double2*>(&vbuf[rb + k]) = make_double2(y[k], y[k + 1]);
    }
    TN_CLK(2);
    for (int j = 0; j < kmax; ++j) {
        if (j == 5) TN_CLK(16);
        __syncthreads();                                    // head updated, body of column j published, part[] free
        if (j == 5) TN_CLK(17);
        double v[TS_BODY];
#pragma unroll
        for (int k = 0; k < TS_BODY; k += 2) {
            const double2 t2 = *reinterpret_cast<const double2*>(&vbuf[rb + k]);
            v[k] = t2.x; v[k + 1] = t2.y;
        }
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
        if (j == 5) TN_CLK(18);
        __syncthreads();
        if (j == 5) TN_CLK(19);
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
        if (j == 5) TN_CLK(20);
        if (tid == 0) taus[blk * 32 + j] = tau;
        // every read of vbuf for this column happened before the barrier above, so the next column may be published now
        if (c == j + 1) {
#pragma unroll
            for (int k = 0; k < TS_BODY; k += 2) *reinterpret_cast<double2*>(&vbuf[rb + k]) = make_double2(y[k], y[k + 1]);
        }
    }
    TN_CLK(3);
    TN_CLK(21);
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
#pragma unroll
    for (int u0 = 0; u0 < 32; u0 += 8) {
        double ov[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + 256 * (u0 + u);
            const int i = ofast ? e >> 5 : e & 255, j = ofast ? e & 31 : e >> 8;
            ov[u] = T[i * P + j];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + 256 * (u0 + u);
            const int i = ofast ? e >> 5 : e & 255, j = ofast ? e & 31 : e >> 8;
            if (i < nr && j < b) Vout[(r0 + i) * ors + j * ocs] = ov[u];
        }
    }
    for (int e = tid; e < b * b; e += 256) {
        const int i = e / b, j = e % b;
        Rout[((int64_t)blk * b + i) * b + j] = (i <= j && i < kmax) ? T[i * P + j] : 0.0;
    }
    TN_CLK(4);
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
    TN_CLK(13);
#pragma unroll
    for (int u0 = 0; u0 < 32; u0 += 8) {             // 8 LDS reads, then 8 stores
        double ov[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + 256 * (u0 + u);
            const int i = qfast ? e >> 5 : e & 255, j = qfast ? e & 31 : e >> 8;
            ov[u] = Tv[i * P + j];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + 256 * (u0 + u);
            const int i = qfast ? e >> 5 : e & 255, j = qfast ? e & 31 : e >> 8;
            if (i < nr && j < b) Qout[(r0 + i) * qrs + j * qcs] = ov[u];
        }
    }
    TN_CLK(14);
}

// Compact-WY form of tsqr_apply_kernel (the default): the block's 32 reflectors are not applied one after the other
// (32 dependent LDS round trips) but as  Q_blk = [Qin; 0] - V (T (V_top^T Qin))  with  T = (diag(1/tau) + striu(V^T V))^-1
// (the dlarft identity; reflectors with tau = 0 are decoupled and get a zero row / column of T):
//   V^T V on the matrix cores (each wave its 64 rows, partial sums through LDS), the 32 x 32 triangular inverse by back
//   substitution in one half-wave (one column per lane), two 32^3 products on the vector ALUs, V G on the matrix cores.
typedef double d4t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void tsqr_apply_wy_kernel(const double* V, int64_t vrs, int64_t vcs, int64_t nrows, int b,
                                                            int nblk, const double* __restrict__ taus,
                                                            const double* __restrict__ Qin, double* Qout, int64_t qrs,
                                                            int64_t qcs) {
    constexpr int P = 33, Q = 33;
    __shared__ double Tv[TS_RB * P];         // reflectors (unit lower trapezoidal); later the output tile
    __shared__ double Sp[4][32 * Q];         // per-wave partial Gram matrices
    __shared__ double Mt[32 * Q];            // diag(1/tau) + striu(V^T V), then T
    __shared__ double Qs[32 * Q];            // Qin (zero padded)
    __shared__ double Z1[32 * Q];            // V_top^T Qin, then G = T Z1
    __shared__ double tl[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, blk = blockIdx.x;
    int64_t r0;
    int nr;
    block_rows(nrows, nblk, blk, r0, nr);
    const bool vfast = (vcs == 1), qfast = (qcs == 1);
    TN_CLK(8);
    {   // all global loads of the prologue in one round trip
        double vv[32], qv[4];
        const double tq = (tid < 32) ? taus[blk * 32 + tid] : 0.0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int e = tid + 256 * t, i = e >> 5, c = e & 31;
            qv[t] = (i < b && i < nr && c < b) ? (Qin ? Qin[((int64_t)blk * b + i) * b + c] : (i == c ? 1.0 : 0.0)) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int e = tid + 256 * u;
            const int i = vfast ? e >> 5 : e & 255, j = vfast ? e & 31 : e >> 8;
            vv[u] = (i < nr && j < b && i > j) ? V[(r0 + i) * vrs + j * vcs] : ((i == j && i < nr && j < b) ? 1.0 : 0.0);
        }
        if (tid < 32) tl[tid] = tq;
        Tv[tid * P + 32] = 0.0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int e = tid + 256 * t;
            Qs[(e >> 5) * Q + (e & 31)] = qv[t];
        }
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int e = tid + 256 * u;
            const int i = vfast ? e >> 5 : e & 255, j = vfast ? e & 31 : e >> 8;
            Tv[i * P + j] = vv[u];
        }
    }
    __syncthreads();
    TN_CLK(9);
    const int li = lane & 15, lk = lane >> 4;
    {   // partial Gram of this wave's 64 rows: tiles (0,0), (0,1), (1,1) of V^T V
        d4t g00 = d4t{0, 0, 0, 0}, g01 = g00, g11 = g00;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int row = wave * 64 + ks * 4 + lk;
            const double f0 = Tv[row * P + li], f1 = Tv[row * P + 16 + li];
            g00 = __builtin_amdgcn_mfma_f64_16x16x4f64(f0, f0, g00, 0, 0, 0);
            g01 = __builtin_amdgcn_mfma_f64_16x16x4f64(f0, f1, g01, 0, 0, 0);
            g11 = __builtin_amdgcn_mfma_f64_16x16x4f64(f1, f1, g11, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = lk + 4 * r;
            Sp[wave][i * Q + li] = g00[r];
            Sp[wave][i * Q + 16 + li] = g01[r];
            Sp[wave][(16 + i) * Q + 16 + li] = g11[r];
        }
    }
    __syncthreads();
    for (int e = tid; e < 32 * 32; e += 256) {
        const int i = e >> 5, j = e & 31;
        const bool di = (tl[i] == 0.0), dj = (tl[j] == 0.0);
        double v = 0.0;
        if (i == j) v = di ? 1.0 : fast_rcp(tl[i]);
        else if (i < j && !di && !dj) v = (Sp[0][i * Q + j] + Sp[1][i * Q + j]) + (Sp[2][i * Q + j] + Sp[3][i * Q + j]);
        Mt[i * Q + j] = v;
    }
    __syncthreads();
    TN_CLK(10);
    double x[32];
    if (tid < 32) {           // column c of the inverse of the upper triangular Mt, bottom-up
        const int c = tid;
        double rdiag[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) rdiag[i] = fast_rcp(Mt[i * Q + i]);       // off the dependent chain
#pragma unroll
        for (int i = 31; i >= 0; --i) {
            double sa[4] = {0.0, 0.0, 0.0, 0.0};                                // four short chains instead of one long one
#pragma unroll
            for (int k = i + 1; k < 32; ++k) sa[k & 3] += Mt[i * Q + k] * x[k];
            const double sacc = (sa[0] + sa[1]) + (sa[2] + sa[3]);
            x[i] = (i > c) ? 0.0 : ((i == c) ? rdiag[i] : -sacc * rdiag[i]);
        }
    }
    __syncthreads();          // every read of Mt happened
    if (tid < 32) {
        const int c = tid;
        const bool dc = (tl[c] == 0.0);
#pragma unroll
        for (int i = 0; i < 32; ++i) Mt[i * Q + c] = (dc && i == c) ? 0.0 : x[i];
    }
    __syncthreads();
    TN_CLK(11);
    // Z1 = V_top^T Qin and G = T Z1: the triangular factors carry explicit zeros, so both run over the full index range with
    // compile-time trip counts (all LDS reads of an entry in flight together, two accumulation chains)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int e = tid + 256 * t, a = e >> 5, c = e & 31;
        double z0 = 0.0, z1 = 0.0;
#pragma unroll
        for (int i = 0; i < 32; i += 2) {
            z0 += Tv[i * P + a] * Qs[i * Q + c];
            z1 += Tv[(i + 1) * P + a] * Qs[(i + 1) * Q + c];
        }
        Z1[a * Q + c] = z0 + z1;
    }
    __syncthreads();
    double gv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {                   // kept in registers until Z1 is free
        const int e = tid + 256 * t, a = e >> 5, c = e & 31;
        double z0 = 0.0, z1 = 0.0;
#pragma unroll
        for (int k = 0; k < 32; k += 2) {
            z0 += Mt[a * Q + k] * Z1[k * Q + c];
            z1 += Mt[a * Q + k + 1] * Z1[(k + 1) * Q + c];
        }
        gv[t] = z0 + z1;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int e = tid + 256 * t;
        Z1[(e >> 5) * Q + (e & 31)] = gv[t];
    }
    __syncthreads();
    TN_CLK(12);
    {   // rows 64 wave .. 64 wave + 63:  out = [Qin; 0] - V G
        double fa[4][8], fb[2][8];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) fa[mt][ks] = Tv[(wave * 64 + mt * 16 + li) * P + ks * 4 + lk];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) fb[nt][ks] = Z1[(ks * 4 + lk) * Q + nt * 16 + li];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                d4t acc = d4t{0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[mt][ks], fb[nt][ks], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wave * 64 + mt * 16 + lk + 4 * r, col = nt * 16 + li;
                    const double top = (row < 32) ? Qs[row * Q + col] : 0.0;
                    Tv[row * P + col] = top - acc[r];
                }
            }
    }
    __syncthreads();
    TN_CLK(13);
#pragma unroll
    for (int u0 = 0; u0 < 32; u0 += 8) {             // 8 LDS reads, then 8 stores
        double ov[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + 256 * (u0 + u);
            const int i = qfast ? e >> 5 : e & 255, j = qfast ? e & 31 : e >> 8;
            ov[u] = Tv[i * P + j];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + 256 * (u0 + u);
            const int i = qfast ? e >> 5 : e & 255, j = qfast ? e & 31 : e >> 8;
            if (i < nr && j < b) Qout[(r0 + i) * qrs + j * qcs] = ov[u];
        }
    }
    TN_CLK(14);
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
        static const bool serial_apply = [] { const char* e = getenv("TN_TSQR_SERIAL_APPLY"); return e && e[0] == '1'; }();
        if (serial_apply)       // reflector-by-reflector form, kept for cross-checking
            hipLaunchKernelGGL(tsqr_apply_kernel, dim3(L.nblk), dim3(256), 0, st, L.V, L.rs, L.cs, L.nrows, b, L.nblk, L.taus, Qin,
                               Qout, qrs, qcs);
        else
            hipLaunchKernelGGL(tsqr_apply_wy_kernel, dim3(L.nblk), dim3(256), 0, st, L.V, L.rs, L.cs, L.nrows, b, L.nblk, L.taus,
                               Qin, Qout, qrs, qcs);
        TN_CHECK_LAUNCH("tsqr_apply_kernel");
        prof_end(st, PROF_TSQR, 2.0 * L.nrows * b * b, 16.0 * L.nrows * b);
    }
    return 0;
}

}  // namespace tn
