// Micro-benchmark: cost of a small device-to-host read-back behind a kernel (what every host decision of the path pays):
// kernel + hipMemcpyAsync(D2H, 8 KB) + hipStreamSynchronize with a pageable and with a page-locked destination, and the same
// with an extra kernel queued behind (does the copy hold the stream up?).  Build: hipcc --offload-arch=gfx950 -O2 -o readback_latency readback_latency.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(double* x, int n) { double s = 0; for (int i = 0; i < n; ++i) s += x[(i * 7 + threadIdx.x) & 1023]; if (s == 123.456) x[0] = s; }
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
    // baseline: kernel + sync only
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 2000; ++i) { hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, d, 0); hipStreamSynchronize(st); }
    printf("kernel + sync only: %.2f us\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 2000);
    return 0;
}
