// K9 — environments of the beam (reference tnac4o.py:1768-1784 `_setup_RR`, :528-535 left-environment update) and the
// bond balancing of the preconditioner (tnac4o.py:1845-1847: scipy.linalg.matrix_balance = LAPACK dgebal, job 'S').
//
// env_rr: one workgroup per distinct boundary suffix.  The reference contracts, key by key in Python,
//     T = A . RR[parent] ; RR' = T . W[:, :, :, u] ; RR' /= nfactor(RR')
// through a (Dl p) x br intermediate per key (134 MB for the 1024 keys of a bulk site when batched).  Here the small
// operands (RR[parent]: Dr x br, W[..u]: bl x p x br) are staged in LDS, the MPO slice is folded into the right environment
// first (V_d = RR . W_d^T, Dr x bl per physical index d) and the site tensor A is streamed once from L2:
//     RR'[x, l] = sum_d sum_x' A[x, d, x'] V_d[x', l]
// accumulated in registers; the power-of-two normalisation happens before the only store.  fp64 VALU FMAs run at the MFMA
// rate on gfx950, and the contraction per key is ~2.6 MFLOP, so nothing here is worth the matrix cores.
// env_rl: the new left environments are rows of T1 = RL . A (already computed for K8): gather + normalise.
#include "common.h"

namespace tn {

constexpr int ENV_MAXACC = 8;          // register accumulators per thread: Dl * bl <= 256 * 8

__device__ __forceinline__ unsigned long long block_absmax_bits(unsigned long long m, unsigned long long* red4) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(m, o, 64); m = t > m ? t : m; }
    const int tid = threadIdx.x;
    if ((tid & 63) == 0) red4[tid >> 6] = m;
    __syncthreads();
    unsigned long long a = red4[0] > red4[1] ? red4[0] : red4[1], b = red4[2] > red4[3] ? red4[2] : red4[3];
    __syncthreads();
    return a > b ? a : b;
}

__global__ __launch_bounds__(256) void env_rr_kernel(const double* __restrict__ A, const double* __restrict__ RRp,
                                                     const double* __restrict__ W, const int32_t* __restrict__ parent,
                                                     const int32_t* __restrict__ uidx, int Dl, int p, int Dr, int bl, int br,
                                                     int pu, double* __restrict__ out) {
    extern __shared__ double lds[];
    double* sRR = lds;                     // [Dr][br]
    double* sW = sRR + Dr * br;            // [bl][p][br]   (the u = uidx[k] slice of W[l,d,r,u])
    double* sV = sW + bl * p * br;         // [Dr][bl]      (current physical index)
    __shared__ unsigned long long red4[4];
    const int tid = threadIdx.x;
    const int64_t k = blockIdx.x;
    const double* rr = RRp + (int64_t)parent[k] * Dr * br;
    const int u = uidx[k];
    for (int e = tid; e < Dr * br; e += 256) sRR[e] = rr[e];
    for (int e = tid; e < bl * p * br; e += 256) sW[e] = W[(int64_t)e * pu + u];
    double acc[ENV_MAXACC];
#pragma unroll
    for (int i = 0; i < ENV_MAXACC; ++i) acc[i] = 0.0;
    const int nout = Dl * bl;
    for (int d = 0; d < p; ++d) {
        __syncthreads();                   // staging done / previous V consumed
        for (int e = tid; e < Dr * bl; e += 256) {
            const int xr = e / bl, l = e % bl;
            const double* w = sW + (l * p + d) * br;
            const double* r = sRR + xr * br;
            double s = 0.0;
            for (int c = 0; c < br; ++c) s += r[c] * w[c];
            sV[e] = s;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ENV_MAXACC; ++i) {
            const int e = tid + 256 * i;
            if (e < nout) {
                const int x = e / bl, l = e % bl;
                const double* a = A + ((int64_t)x * p + d) * Dr;
                double s = acc[i];
                for (int c = 0; c < Dr; ++c) s += a[c] * sV[c * bl + l];
                acc[i] = s;
            }
        }
    }
    unsigned long long m = 0ULL;
#pragma unroll
    for (int i = 0; i < ENV_MAXACC; ++i) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(fabs(acc[i]));
        m = b > m ? b : m;
    }
    m = block_absmax_bits(m, red4);
    const double inv = 1.0 / ldexp(1.0, (int)((long long)(m >> 52) - 1023));     // 1 / nfactor (mps.py:83-85)
    double* o = out + k * nout;
#pragma unroll
    for (int i = 0; i < ENV_MAXACC; ++i) {
        const int e = tid + 256 * i;
        if (e < nout) o[e] = acc[i] * inv;
    }
}

int env_rr_batched(hipStream_t st, const double* A, const double* RRprev, const double* W, const int32_t* parent,
                   const int32_t* uidx, int64_t nk, int64_t Dl, int64_t p, int64_t Dr, int64_t bl, int64_t br, int64_t pu,
                   double* out) {
    if (nk <= 0) return 0;
    TN_CHECK_ARG(Dl >= 1 && p >= 1 && Dr >= 1 && bl >= 1 && br >= 1 && pu >= 1, "non-positive dimension");
    TN_CHECK_ARG(Dl * bl <= 256 * ENV_MAXACC, "Dl * bl exceeds 2048 (env_rr accumulators)");
    const int64_t lds = (Dr * br + bl * p * br + Dr * bl) * 8;
    TN_CHECK_ARG(lds <= 150 * 1024, "site too large for env_rr");
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)env_rr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    TN_PROF_LAUNCH(st, PROF_MISC, hipLaunchKernelGGL(env_rr_kernel, dim3((unsigned)nk), dim3(256), (size_t)lds, st, A, RRprev, W, parent, uidx,
                       (int)Dl, (int)p, (int)Dr, (int)bl, (int)br, (int)pu, out));
    TN_CHECK_LAUNCH("env_rr_kernel");
    return 0;
}

// out[k, :] = T1[par[k], didx[k], :] / nfactor(that row)      (one wave per key, 4 keys per workgroup)
__global__ __launch_bounds__(256) void env_rl_kernel(const double* __restrict__ T1, const int32_t* __restrict__ par,
                                                     const int32_t* __restrict__ didx, int64_t nk, int p, int Dr,
                                                     double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= nk) return;                                   // whole wave leaves together
    const double* row = T1 + ((int64_t)par[k] * p + didx[k]) * Dr;
    unsigned long long m = 0ULL;
    for (int c = lane; c < Dr; c += 64) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(fabs(row[c]));
        m = b > m ? b : m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(m, o, 64); m = t > m ? t : m; }
    const double inv = 1.0 / ldexp(1.0, (int)((long long)(m >> 52) - 1023));
    for (int c = lane; c < Dr; c += 64) out[k * Dr + c] = row[c] * inv;
}

int env_rl_batched(hipStream_t st, const double* T1, const int32_t* par, const int32_t* didx, int64_t nk, int64_t p, int64_t Dr,
                   double* out) {
    if (nk <= 0) return 0;
    TN_CHECK_ARG(p >= 1 && Dr >= 1, "non-positive dimension");
    TN_PROF_LAUNCH(st, PROF_MISC, hipLaunchKernelGGL(env_rl_kernel, dim3((unsigned)cdiv(nk, 4)), dim3(256), 0, st, T1, par, didx, nk, (int)p, (int)Dr, out));
    TN_CHECK_LAUNCH("env_rl_kernel");
    return 0;
}

// ---- balancing (LAPACK dgebal, job = 'S': scaling only, no permutation; the 3.12 formulation with 2-norms) -----------
// One wave; the matrix (n <= 64) lives in LDS, lane j owns column j / row j reductions.  The loop over i is sequential as in
// LAPACK (the scaling of row/column i changes the norms seen by i+1), the reductions inside a step are parallel.
// scale_out[i] = clamp(scale[i], 1/max_scale, max_scale) as tnac4o.py:1847 does right after the call (max_scale <= 0: no clamp).
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
// sqrt(sum x^2) without spurious over/underflow: scaled by the largest magnitude (a power-of-two-free variant of dnrm2;
// agrees with it to rounding, which only matters if a ratio of norms sits within an ulp of a power of two)
__device__ __forceinline__ double wave_nrm2(double x) {
    const double ax = fabs(x);
    const double mx = wave_max(ax);
    if (mx == 0.0) return 0.0;
    const double t = ax / mx;
    return mx * sqrt(wave_sum(t * t));
}

__global__ __launch_bounds__(64) void balance_kernel(const double* __restrict__ Ain, int64_t rs, int64_t cs, int n,
                                                     double max_scale, double* __restrict__ scale_out, int* __restrict__ iters_out) {
    constexpr int P = 65;
    __shared__ double M[64 * P];
    __shared__ double sc[64];
    const int lane = threadIdx.x;
    for (int e = lane; e < n * n; e += 64) M[(e / n) * P + (e % n)] = Ain[(int64_t)(e / n) * rs + (int64_t)(e % n) * cs];
    if (lane < n) sc[lane] = 1.0;
    __syncthreads();
    const double radix = 2.0, factor = 0.95;
    const double sfmin1 = 2.2250738585072014e-308 / 2.220446049250313e-16;      // dlamch('S') / dlamch('P')
    const double sfmax1 = 1.0 / sfmin1, sfmin2 = sfmin1 * radix, sfmax2 = 1.0 / sfmin2;
    int it = 0;
    bool noconv = true;
    while (noconv && it < 1000) {
        noconv = false;
        ++it;
        for (int i = 0; i < n; ++i) {
            const double colv = lane < n ? M[lane * P + i] : 0.0;        // A(lane, i)
            const double rowv = lane < n ? M[i * P + lane] : 0.0;        // A(i, lane)
            double c = wave_nrm2(colv), r = wave_nrm2(rowv);
            double ca = wave_max(fabs(colv)), ra = wave_max(fabs(rowv));
            if (c == 0.0 || r == 0.0) continue;                          // uniform across the wave
            double g = r / radix, f = 1.0;
            const double s = c + r;
            while (c < g && fmax(f, fmax(c, ca)) < sfmax2 && fmin(r, fmin(g, ra)) > sfmin2) {
                f *= radix; c *= radix; ca *= radix; r /= radix; g /= radix; ra /= radix;
            }
            g = c / radix;
            while (g >= r && fmax(r, ra) < sfmax2 && fmin(fmin(f, c), fmin(g, ca)) > sfmin2) {
                f /= radix; c /= radix; g /= radix; ca /= radix; r *= radix; ra *= radix;
            }
            if (c + r >= factor * s) continue;
            const double si = sc[i];
            if (f < 1.0 && si < 1.0 && f * si <= sfmin1) continue;
            if (f > 1.0 && si > 1.0 && si >= sfmax1 / f) continue;
            noconv = true;
            if (lane == 0) sc[i] = si * f;
            const double ginv = 1.0 / f;
            if (lane < n) {
                M[i * P + lane] *= ginv;                                 // row i
            }
            __syncthreads();
            if (lane < n) {
                M[lane * P + i] *= f;                                    // column i (the diagonal gets both, as in LAPACK)
            }
            __syncthreads();
        }
    }
    if (lane < n) {
        double v = sc[lane];
        if (max_scale > 0.0) v = fmin(fmax(v, 1.0 / max_scale), max_scale);
        scale_out[lane] = v;
    }
    if (lane == 0 && iters_out) *iters_out = it;
}

int balance(hipStream_t st, const double* A, int64_t rs, int64_t cs, int64_t n, double max_scale, double* scale_out, int* iters_out) {
    TN_CHECK_ARG(n >= 1 && n <= 64, "balance handles 1 <= n <= 64");
    TN_PROF_LAUNCH(st, PROF_MISC, hipLaunchKernelGGL(balance_kernel, dim3(1), dim3(64), 0, st, A, rs, cs, (int)n, max_scale, scale_out, iters_out));
    TN_CHECK_LAUNCH("balance_kernel");
    return 0;
}

}  // namespace tn
