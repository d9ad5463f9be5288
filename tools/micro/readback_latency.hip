// Micro-benchmark: cost of a small device-to-host read-back behind a kernel (what every host decision of the path pays):
// kernel + hipMemcpyAsync(D2H, 8 KB) + hipStreamSynchronize with a pageable and with a page-locked destination, and the same
// with an extra kernel queued behind (does the copy hold the stream up?).  Build: hipcc --offload-arch=gfx950 -O2 -o readback_latency readback_latency.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(double* x, int n) { double s = 0; for (int i = 0; i < n; ++i) s += x[(i * 7 + threadIdx.x) & 1023]; if (s == 123.456) x[0] = s; }
// mailbox form: the kernel stores its 8 KB result straight into page-locked host memory and then raises a sequence flag (system
// scope); the host polls the flag -- no copy in the queue, no stream synchronisation
__global__ void spin_mail(double* x, int n, double* host_out, unsigned* host_flag, unsigned seq) {
    double s = 0;
    for (int i = 0; i < n; ++i) s += x[(i * 7 + threadIdx.x) & 1023];
    if (s == 123.456) x[0] = s;
    for (int i = threadIdx.x; i < 1024; i += 64) __hip_atomic_store(host_out + i, x[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
int main() {
    double* d; hipMalloc(&d, 1 << 20); hipMemset(d, 0, 1 << 20);
    hipStream_t st; hipStreamCreate(&st);
    std::vector<double> pageable(1024);
    double* pinned; hipHostMalloc(&pinned, 8192, hipHostMallocDefault);
    for (int mode = 0; mode < 2; ++mode) {
        double* dst = mode ? pinned : pageable.data();
        for (int work : {0, 2000}) {
            for (int rep = 0; rep < 2; ++rep) {
                auto t0 = std::chrono::steady_clock::now();
                const int N = 2000;
                for (int i = 0; i < N; ++i) {
                    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, d, work);
                    hipMemcpyAsync(dst, d, 8192, hipMemcpyDeviceToHost, st);
                    hipStreamSynchronize(st);
                }
                double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
                if (rep) printf("%s destination, kernel loop %d: %.2f us per (kernel + 8 KB read-back + sync)\n", mode ? "page-locked" : "pageable   ", work, us);
            }
        }
    }
    {
        double* mail; hipHostMalloc(&mail, 8192, hipHostMallocDefault);
        unsigned* flag; hipHostMalloc(&flag, 64, hipHostMallocDefault);
        *flag = 0;
        for (int work : {0, 2000}) {
            for (int rep = 0; rep < 2; ++rep) {
                auto t0 = std::chrono::steady_clock::now();
                const int N = 2000;
                static unsigned seq = 0;
                for (int i = 0; i < N; ++i) {
                    ++seq;
                    hipLaunchKernelGGL(spin_mail, dim3(1), dim3(64), 0, st, d, work, mail, flag, seq);
                    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) { }
                }
                double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
                if (rep) printf("mailbox (8 KB stored to page-locked memory + flag, host polls), kernel loop %d: %.2f us\n", work, us);
            }
        }
        hipStreamSynchronize(st);
    }
    // baseline: kernel + sync only
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 2000; ++i) { hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, d, 0); hipStreamSynchronize(st); }
    printf("kernel + sync only: %.2f us\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 2000);
    return 0;
}
