// Launch-rate micro-benchmark: T host threads, each with its own stream, each enqueueing K tiny dependent kernels; time per launch per
// chain for T = 1, 2, 4 (is the launch path of a process serialised across streams?), with GPU_MAX_HW_QUEUES as given in the environment.
// Build and run on the GPU box:  hipcc --offload-arch=gfx950 -O2 -o /tmp/launch_rate tools/micro/launch_rate.hip && /tmp/launch_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

__global__ void tiny(double* x) { if (threadIdx.x == 0 && blockIdx.x == 0) x[0] += 1.0; }
__global__ void wide(double* x, long n) { for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) x[i] = x[i] * 1.0000001; }

int main() {
    const int K = 20000;
    for (int mode = 0; mode < 2; ++mode)
        for (int T : {1, 2, 4}) {
            std::vector<hipStream_t> st(T);
            std::vector<double*> buf(T);
            std::vector<double> secs(T);
            const long n = mode ? (4L << 20) : 1;
            for (int t = 0; t < T; ++t) { (void)hipStreamCreate(&st[t]); (void)hipMalloc(&buf[t], n * 8); (void)hipMemset(buf[t], 0, n * 8); }
            (void)hipDeviceSynchronize();
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    const int k = mode ? K / 20 : K;
                    auto t0 = std::chrono::steady_clock::now();
                    for (int i = 0; i < k; ++i) {
                        if (mode) hipLaunchKernelGGL(wide, dim3(1024), dim3(256), 0, st[t], buf[t], n);
                        else hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st[t], buf[t]);
                    }
                    (void)hipStreamSynchronize(st[t]);
                    secs[t] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / k;
                });
            for (auto& x : th) x.join();
            double mx = 0;
            for (double s : secs) mx = s > mx ? s : mx;
            printf("%s kernels, %d chain(s): %.2f us per launch per chain\n", mode ? "32 MB" : "tiny", T, 1e6 * mx);
            for (int t = 0; t < T; ++t) { (void)hipFree(buf[t]); (void)hipStreamDestroy(st[t]); }
        }
    return 0;
}
