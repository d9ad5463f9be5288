// K8 / K9 — the beam-side contractions of search_ground_state (reference tnac4o.py:437-537).
//
// calc_pn: one workgroup per branch.  The reference evaluates, branch by branch in Python,
//   T1 = RL . AT ; T2 = T1 . RR ; Pn[s] = sum_{d,r} AA[s,d,r] T2[d,r]      (tnac4o.py:1792-1794)
// where AA is a one-hot slice of the 134 MB dense PEPS tensor.  Here T1 is shared by all branches with the same
// boundary prefix (it comes from one GEMM over all prefixes), RR by all branches with the same suffix, and AA is
// replaced by its non-zero factor F[s,l,u] with the index maps dmap/rmap, so Pn[s] = F[s,l,u] * T2[dmap[s], rmap[s]].
// The negative-probability rule and normalisation of tnac4o.py:1795-1807 are applied in the same kernel.
#include "common.h"

namespace tn {

__device__ __forceinline__ double block_sum(double v, double* red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (tid < k) red[tid] += red[tid + k];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

__device__ __forceinline__ double block_min(double v, double* red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (tid < k) red[tid] = fmin(red[tid], red[tid + k]);
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void calc_pn_kernel(const double* __restrict__ T1, const double* __restrict__ RR,
                                                      const double* __restrict__ F, const int32_t* __restrict__ dmap,
                                                      const int32_t* __restrict__ rmap, const int32_t* __restrict__ pref,
                                                      const int32_t* __restrict__ suf, const int32_t* __restrict__ lidx,
                                                      const int32_t* __restrict__ uidx, int q, int nl, int nu, int p,
                                                      int Dr, int br, double* __restrict__ P, double* __restrict__ minP,
                                                      const double* __restrict__ parent_log2p, double* __restrict__ log2p_out) {
    extern __shared__ double lds[];
    double* sT1 = lds;                  // [p][Dr]
    double* sRR = sT1 + p * Dr;         // [Dr][br]
    double* sT2 = sRR + Dr * br;        // [p][br]
    double* sP = sT2 + p * br;          // [q]
    __shared__ double red[256];
    const int tid = threadIdx.x;
    const int64_t kk = blockIdx.x;
    const double* t1 = T1 + (int64_t)pref[kk] * p * Dr;
    const double* rr = RR + (int64_t)suf[kk] * Dr * br;
    for (int e = tid; e < p * Dr; e += 256) sT1[e] = t1[e];
    for (int e = tid; e < Dr * br; e += 256) sRR[e] = rr[e];
    __syncthreads();
    for (int e = tid; e < p * br; e += 256) {
        const int d = e / br, r = e % br;
        double s = 0.0;
        for (int c = 0; c < Dr; ++c) s += sT1[d * Dr + c] * sRR[c * br + r];
        sT2[e] = s;
    }
    __syncthreads();
    const int l = lidx[kk], u = uidx[kk];
    double mn = 1.7e308;
    for (int s = tid; s < q; s += 256) {
        const double v = F[((int64_t)s * nl + l) * nu + u] * sT2[dmap[s] * br + rmap[s]];
        sP[s] = v;
        mn = fmin(mn, v);
    }
    double mPn = block_min(mn, red);
    if (mPn < 0.0) {                                   // tnac4o.py:1796-1799
        const double a = fabs(mPn);
        double cnt = 0.0;
        for (int s = tid; s < q; s += 256)
            if (sP[s] < a) { sP[s] = a; cnt += 1.0; }
        mPn *= block_sum(cnt, red);
    }
    double part = 0.0;
    for (int s = tid; s < q; s += 256) part += sP[s];
    const double no = block_sum(part, red);
    double* out = P + kk * q;
    if (no > 0.0) {                                    // tnac4o.py:1800-1803
        const double inv = 1.0 / no;
        for (int s = tid; s < q; s += 256) out[s] = sP[s] * inv;
        mPn *= inv;
    } else {                                           // all zeros -> uniform, flag -1 (tnac4o.py:1804-1806)
        for (int s = tid; s < q; s += 256) out[s] = sP[s] + 1.0 / (double)q;
        mPn = -1.0;
    }
    if (tid == 0) minP[kk] = mPn;
    if (log2p_out) {                                   // tnac4o.py:450-453: log2 of the table plus the parent's log-probability
        const double base = parent_log2p[kk];
        double* lo = log2p_out + kk * q;
        for (int s = tid; s < q; s += 256) lo[s] = log2(out[s]) + base;      // (out[s] was written by this very thread)
    }
}

int calc_pn(hipStream_t st, const double* T1, const double* RR, const double* F, const int32_t* dmap, const int32_t* rmap,
            const int32_t* pref, const int32_t* suf, const int32_t* lidx, const int32_t* uidx, int64_t nb, int64_t q,
            int64_t nl, int64_t nu, int64_t p, int64_t Dr, int64_t br, double* P, double* minP, const double* parent_log2p,
            double* log2p_out) {
    if (nb <= 0) return 0;
    TN_CHECK_ARG((parent_log2p == nullptr) == (log2p_out == nullptr), "parent_log2p and log2p_out go together");
    TN_CHECK_ARG(q >= 1 && nl >= 1 && nu >= 1 && p >= 1 && Dr >= 1 && br >= 1, "non-positive dimension");
    const int64_t lds = (p * Dr + Dr * br + p * br + q) * 8;
    TN_CHECK_ARG(lds <= 150 * 1024, "site too large for calc_pn");
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)calc_pn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    TN_PROF_LAUNCH(st, PROF_MISC, hipLaunchKernelGGL(calc_pn_kernel, dim3((unsigned)nb), dim3(256), (size_t)lds, st, T1, RR, F, dmap, rmap, pref, suf, lidx,
                       uidx, (int)q, (int)nl, (int)nu, (int)p, (int)Dr, (int)br, P, minP, parent_log2p, log2p_out));
    TN_CHECK_LAUNCH("calc_pn_kernel");
    return 0;
}

// each block of `len` doubles divided by 2^floor(log2 max|block|)   (tnac4o.py:533, 1781)
__global__ __launch_bounds__(256) void nfactor_batched_kernel(double* __restrict__ x, int64_t len) {
    __shared__ unsigned long long red[256];
    const int tid = threadIdx.x;
    double* b = x + (int64_t)blockIdx.x * len;
    unsigned long long m = 0ULL;
    for (int64_t i = tid; i < len; i += 256) {
        const unsigned long long v = (unsigned long long)__double_as_longlong(fabs(b[i]));
        m = v > m ? v : m;
    }
    red[tid] = m;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (tid < k) red[tid] = red[tid] > red[tid + k] ? red[tid] : red[tid + k];
        __syncthreads();
    }
    const double inv = 1.0 / ldexp(1.0, (int)((long long)(red[0] >> 52) - 1023));
    for (int64_t i = tid; i < len; i += 256) b[i] *= inv;
}

int nfactor_batched(hipStream_t st, double* x, int64_t batch, int64_t len) {
    if (batch <= 0 || len <= 0) return 0;
    TN_PROF_LAUNCH(st, PROF_MISC, hipLaunchKernelGGL(nfactor_batched_kernel, dim3((unsigned)batch), dim3(256), 0, st, x, len));
    TN_CHECK_LAUNCH("nfactor_batched_kernel");
    return 0;
}

// ---- merge of branches with identical boundary indices (tnac4o.py:481-509), one thread per group ------------------------------
// The candidates of a site-step arrive sorted by group (stable, i.e. in candidate order inside a group): E, log2 p, degeneracy and
// the candidate position of every member, group g = members starts[g] .. starts[g+1]-1.  Per group: the representative is the FIRST
// member of minimal energy, the degeneracies of the members within min_dEng of that minimum are added up, and the group's log2 p is
// the representative's when it is alone, else the mean over those members, summed in member order (a fixed order: the host path of
// tnac4o.search_ground_state adds in the same order, so both give the same bits).
__global__ __launch_bounds__(256) void merge_groups_kernel(const double* __restrict__ E, const double* __restrict__ lp, const int64_t* __restrict__ deg,
                                                           const int64_t* __restrict__ pos, const int64_t* __restrict__ starts, int64_t ng, double min_dEng,
                                                           int64_t* __restrict__ rep_pos, int64_t* __restrict__ degn, double* __restrict__ lpn) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= ng) return;
    const int64_t lo = starts[g], hi = starts[g + 1];
    double emin = E[lo];
    int64_t first = lo;
    for (int64_t i = lo + 1; i < hi; ++i)
        if (E[i] < emin) { emin = E[i]; first = i; }
    int64_t cnt = 0, d = 0;
    double acc = 0.0;
    for (int64_t i = lo; i < hi; ++i)
        if (E[i] - emin <= min_dEng) { ++cnt; d += deg[i]; acc += lp[i]; }
    rep_pos[g] = pos[first];
    degn[g] = d;
    lpn[g] = (cnt > 1) ? acc / (double)cnt : lp[first];
}

int merge_groups(hipStream_t st, const double* E, const double* lp, const int64_t* deg, const int64_t* pos, const int64_t* starts, int64_t ng,
                 double min_dEng, int64_t* rep_pos, int64_t* degn, double* lpn) {
    if (ng <= 0) return 0;
    TN_PROF_LAUNCH(st, PROF_MISC, hipLaunchKernelGGL(merge_groups_kernel, dim3((unsigned)cdiv(ng, 256)), dim3(256), 0, st, E, lp, deg, pos, starts, ng, min_dEng,
                       rep_pos, degn, lpn));
    TN_CHECK_LAUNCH("merge_groups_kernel");
    return 0;
}

}  // namespace tn
