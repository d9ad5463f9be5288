// K6 and small elementwise helpers: power-of-two normalisation factor (reference mps.py:76-85), scaling by its
// inverse (mps.py:782, 797; tnac4o.py:533, 1781), diagonal operator on the physical leg (mps.py:361-366).
#include "common.h"

namespace tn {

struct PinnedSlot {
    void* p = nullptr;
    size_t cap = 0;
    ~PinnedSlot() { if (p) (void)hipHostFree(p); }
};
void* pinned_host(size_t bytes, int slot) {
    thread_local PinnedSlot slots[8];
    if (slot < 0 || slot > 7) return nullptr;
    PinnedSlot& s = slots[slot];
    if (s.cap < bytes) {
        if (s.p) { (void)hipHostFree(s.p); s.p = nullptr; s.cap = 0; }
        const size_t cap = bytes * 2 > 65536 ? bytes * 2 : 65536;
        if (hipHostMalloc(&s.p, cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); s.p = nullptr; return nullptr; }
        s.cap = cap;
    }
    return s.p;
}


// max |x| via atomicMax on the bit pattern (non-negative doubles order like unsigned integers)
__global__ __launch_bounds__(256) void absmax_bits_kernel(const double* __restrict__ x, int64_t n,
                                                          unsigned long long* __restrict__ slot) {
    __shared__ unsigned long long red[256];
    const int tid = threadIdx.x;
    unsigned long long m = 0ULL;
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < n; i += (int64_t)gridDim.x * 256) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(fabs(x[i]));
        m = b > m ? b : m;
    }
    red[tid] = m;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (tid < k) red[tid] = red[tid] > red[tid + k] ? red[tid] : red[tid + k];
        __syncthreads();
    }
    if (tid == 0) atomicMax(slot, red[0]);
}

// out[0] = 2^floor(log2(max|x|)) from the exponent field (2^-1023 for zero/subnormal input, like the reference);
// out[1] = its reciprocal.  NaN/Inf input propagates an Inf factor.
__global__ void nfactor_finish_kernel(const unsigned long long* __restrict__ slot, double* __restrict__ out) {
    const long long e = (long long)(slot[0] >> 52) - 1023;
    const double f = ldexp(1.0, (int)e);
    out[0] = f;
    out[1] = 1.0 / f;
}

__global__ __launch_bounds__(256) void scale_by_kernel(double* __restrict__ x, int64_t n, const double* __restrict__ s) {
    const double f = s[0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] *= f;
}

int nfactor(hipStream_t st, const double* x, int64_t n, double* out2, void* slot8) {
    TN_CHECK_ARG(n >= 1, "empty input");
    hipError_t e = hipMemsetAsync(slot8, 0, 8, st);
    if (e != hipSuccess) return hip_fail(e, "memset slot");
    int64_t nb = cdiv(n, 256 * 8);
    if (nb > 1024) nb = 1024;
    TN_PROF_LAUNCH(st, PROF_MISC, hipLaunchKernelGGL(absmax_bits_kernel, dim3((unsigned)nb), dim3(256), 0, st, x, n, (unsigned long long*)slot8));
    TN_CHECK_LAUNCH("absmax_bits_kernel");
    TN_PROF_LAUNCH(st, PROF_MISC, hipLaunchKernelGGL(nfactor_finish_kernel, dim3(1), dim3(1), 0, st, (const unsigned long long*)slot8, out2));
    TN_CHECK_LAUNCH("nfactor_finish_kernel");
    return 0;
}

// x /= nfactor(x) in two launches and without a memset: per-block maxima into `scratch`, then every block of the scaling
// kernel reduces those (<= 1024 values) itself; block 0 also publishes [nf, 1/nf].
__global__ __launch_bounds__(256) void absmax_blocks_kernel(const double* __restrict__ x, int64_t n,
                                                            unsigned long long* __restrict__ scratch) {
    __shared__ unsigned long long red[4];
    const int tid = threadIdx.x;
    unsigned long long m = 0ULL;
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < n; i += (int64_t)gridDim.x * 256) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(fabs(x[i]));
        m = b > m ? b : m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(m, o, 64); m = t > m ? t : m; }
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
        unsigned long long a = red[0] > red[1] ? red[0] : red[1], b = red[2] > red[3] ? red[2] : red[3];
        scratch[blockIdx.x] = a > b ? a : b;
    }
}

__global__ __launch_bounds__(256) void scale_by_blockmax_kernel(double* __restrict__ x, int64_t n,
                                                                const unsigned long long* __restrict__ scratch, int nparts,
                                                                double* __restrict__ out2) {
    __shared__ unsigned long long red[4];
    const int tid = threadIdx.x;
    unsigned long long m = 0ULL;
    for (int i = tid; i < nparts; i += 256) { const unsigned long long b = scratch[i]; m = b > m ? b : m; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(m, o, 64); m = t > m ? t : m; }
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    unsigned long long a = red[0] > red[1] ? red[0] : red[1], b = red[2] > red[3] ? red[2] : red[3];
    a = a > b ? a : b;
    const double f = ldexp(1.0, (int)((long long)(a >> 52) - 1023)), inv = 1.0 / f;
    if (blockIdx.x == 0 && tid == 0) { out2[0] = f; out2[1] = inv; }
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < n; i += (int64_t)gridDim.x * 256) x[i] *= inv;
}

// ... and in ONE launch for up to 32768 values (the triangular factors and centre matrices of a sweep): a single workgroup of 1024
// threads keeps its 32 values per thread in registers between the maximum and the scaling (same result bit for bit).
__global__ __launch_bounds__(1024) void normalize_small_kernel(double* __restrict__ x, int n, double* __restrict__ out2) {
    __shared__ unsigned long long red[16];
    const int tid = threadIdx.x;
    double v[32];
    unsigned long long m = 0ULL;
#pragma unroll
    for (int u = 0; u < 32; ++u) {
        const int i = tid + 1024 * u;
        v[u] = i < n ? x[i] : 0.0;
        const unsigned long long b = (unsigned long long)__double_as_longlong(fabs(v[u]));
        m = b > m ? b : m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(m, o, 64); m = t > m ? t : m; }
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    unsigned long long a = 0ULL;
#pragma unroll
    for (int w = 0; w < 16; ++w) a = red[w] > a ? red[w] : a;
    const double f = ldexp(1.0, (int)((long long)(a >> 52) - 1023)), inv = 1.0 / f;
    if (tid == 0) { out2[0] = f; out2[1] = inv; }
#pragma unroll
    for (int u = 0; u < 32; ++u) {
        const int i = tid + 1024 * u;
        if (i < n) x[i] = v[u] * inv;
    }
}

int normalize_pow2(hipStream_t st, double* x, int64_t n, double* out2, void* scratch, int64_t scratch_bytes) {
    TN_CHECK_ARG(n >= 1, "empty input");
    if (n <= 32768) {
        TN_PROF_LAUNCH(st, PROF_MISC, hipLaunchKernelGGL(normalize_small_kernel, dim3(1), dim3(1024), 0, st, x, (int)n, out2));
        TN_CHECK_LAUNCH("normalize_small_kernel");
        return 0;
    }
    int64_t nb = cdiv(n, 256 * 8);
    if (nb > 1024) nb = 1024;
    TN_CHECK_ARG(scratch_bytes >= nb * 8, "scratch too small (8 KiB always suffices)");
    TN_PROF_LAUNCH(st, PROF_MISC, hipLaunchKernelGGL(absmax_blocks_kernel, dim3((unsigned)nb), dim3(256), 0, st, x, n, (unsigned long long*)scratch));
    TN_CHECK_LAUNCH("absmax_blocks_kernel");
    int64_t ns = cdiv(n, 256 * 4);
    if (ns > 2048) ns = 2048;
    TN_PROF_LAUNCH(st, PROF_MISC, hipLaunchKernelGGL(scale_by_blockmax_kernel, dim3((unsigned)ns), dim3(256), 0, st, x, n,
                       (const unsigned long long*)scratch, (int)nb, out2));
    TN_CHECK_LAUNCH("scale_by_blockmax_kernel");
    return 0;
}

int scale_by(hipStream_t st, double* x, int64_t n, const double* scalar_dev) {
    if (n <= 0) return 0;
    int64_t nb = cdiv(n, 256 * 4);
    if (nb > 2048) nb = 2048;
    TN_PROF_LAUNCH(st, PROF_MISC, hipLaunchKernelGGL(scale_by_kernel, dim3((unsigned)nb), dim3(256), 0, st, x, n, scalar_dev));
    TN_CHECK_LAUNCH("scale_by_kernel");
    return 0;
}

// A[dl, s, dr] *= diag[s]   (inv != 0: divide)
__global__ __launch_bounds__(256) void scale_phys_kernel(double* __restrict__ A, int64_t Dl, int64_t p, int64_t Dr,
                                                         const double* __restrict__ diag, int inv) {
    const int64_t n = Dl * p * Dr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double d = diag[(i / Dr) % p];
        A[i] = inv ? A[i] / d : A[i] * d;
    }
}

int scale_phys(hipStream_t st, double* A, int64_t Dl, int64_t p, int64_t Dr, const double* diag, int inv) {
    const int64_t n = Dl * p * Dr;
    if (n <= 0) return 0;
    int64_t nb = cdiv(n, 256 * 4);
    if (nb > 2048) nb = 2048;
    TN_PROF_LAUNCH(st, PROF_MISC, hipLaunchKernelGGL(scale_phys_kernel, dim3((unsigned)nb), dim3(256), 0, st, A, Dl, p, Dr, diag, inv));
    TN_CHECK_LAUNCH("scale_phys_kernel");
    return 0;
}

}  // namespace tn
