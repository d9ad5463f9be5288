// K7 — structured PEPS-factor / MPO builder (reference tnac4o.py:1562-1672 `_peps_tensor` and the sum over the
// physical index at :1686).  The reference allocates the dense 5-leg tensor (q,l,d,r,u) — 134 MB and 1/256 dense for
// chimera cells — at every site of every sweep and search step.  Here only its non-zero factor is formed,
//   F[s,l,u] = exp((Es[s] + E1[s,l]) + E4[s,u]) * Xu[u] * Xl[l] * Xr[rmap[s]] * Xd[dmap[s]]
// (same floating-point evaluation order as the reference; Es/E1/E4 are the beta-scaled, min-shifted energy tables),
// and the row MPO site  W[l,d,r,u] = sum_{s: dmap[s]=d, rmap[s]=r} F[s,l,u]  summed in increasing s.
#include "common.h"

namespace tn {

__global__ __launch_bounds__(256) void peps_factor_kernel(const double* __restrict__ Es, const double* __restrict__ E1,
                                                          const double* __restrict__ E4, const double* __restrict__ Xu,
                                                          const double* __restrict__ Xl, const double* __restrict__ Xr,
                                                          const double* __restrict__ Xd, const int32_t* __restrict__ dmap,
                                                          const int32_t* __restrict__ rmap, int q, int nl, int nu,
                                                          double* __restrict__ F) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)q * nl * nu) return;
    const int u = (int)(e % nu), l = (int)((e / nu) % nl), s = (int)(e / ((int64_t)nu * nl));
    double f = exp((Es[s] + E1[s * nl + l]) + E4[s * nu + u]);
    f = f * Xu[u];
    f = f * Xl[l];
    f = f * Xr[rmap[s]];
    f = f * Xd[dmap[s]];
    F[e] = f;
}

// one thread per output element (l,d,r,u); sequential scan over s keeps the reference's summation order
__global__ __launch_bounds__(256) void mpo_from_factor_kernel(const double* __restrict__ F, const int32_t* __restrict__ dmap,
                                                              const int32_t* __restrict__ rmap, int q, int nl, int nu, int pd,
                                                              int br, double* __restrict__ W) {
    extern __shared__ int32_t maps[];          // dmap | rmap
    for (int s = threadIdx.x; s < q; s += 256) { maps[s] = dmap[s]; maps[q + s] = rmap[s]; }
    __syncthreads();
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)nl * pd * br * nu) return;
    const int u = (int)(e % nu), r = (int)((e / nu) % br), d = (int)((e / ((int64_t)nu * br)) % pd);
    const int l = (int)(e / ((int64_t)nu * br * pd));
    double w = 0.0;
    for (int s = 0; s < q; ++s)
        if (maps[s] == d && maps[q + s] == r) w += F[((int64_t)s * nl + l) * nu + u];
    W[e] = w;
}

int peps_factor(hipStream_t st, const double* Es, const double* E1, const double* E4, const double* Xu, const double* Xl,
                const double* Xr, const double* Xd, const int32_t* dmap, const int32_t* rmap, int64_t q, int64_t nl, int64_t nu,
                double* F) {
    TN_CHECK_ARG(q >= 1 && nl >= 1 && nu >= 1, "non-positive dimension");
    TN_PROF_LAUNCH(st, PROF_MISC, hipLaunchKernelGGL(peps_factor_kernel, dim3((unsigned)cdiv(q * nl * nu, 256)), dim3(256), 0, st, Es, E1, E4, Xu, Xl, Xr, Xd,
                       dmap, rmap, (int)q, (int)nl, (int)nu, F));
    TN_CHECK_LAUNCH("peps_factor_kernel");
    return 0;
}

int mpo_from_factor(hipStream_t st, const double* F, const int32_t* dmap, const int32_t* rmap, int64_t q, int64_t nl, int64_t nu,
                    int64_t pd, int64_t br, double* W) {
    TN_CHECK_ARG(q >= 1 && nl >= 1 && nu >= 1 && pd >= 1 && br >= 1, "non-positive dimension");
    TN_CHECK_ARG(q <= 8192, "too many cell states");
    TN_PROF_LAUNCH(st, PROF_MISC, hipLaunchKernelGGL(mpo_from_factor_kernel, dim3((unsigned)cdiv(nl * pd * br * nu, 256)), dim3(256), (size_t)(2 * q * 4), st, F,
                       dmap, rmap, (int)q, (int)nl, (int)nu, (int)pd, (int)br, W));
    TN_CHECK_LAUNCH("mpo_from_factor_kernel");
    return 0;
}

}  // namespace tn
