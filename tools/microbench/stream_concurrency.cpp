// How many HIP streams of one process really run their kernels side by side on an MI355X, as a function of
// GPU_MAX_HW_QUEUES: n streams, one 2 ms nap kernel (one workgroup) each, wall time of the batch.
// hipcc --offload-arch=gfx950 -O2 -o /tmp/stream_concurrency tools/microbench/stream_concurrency.cpp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void nap(unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
int main()
{
    for (int n : {1, 2, 4, 6, 8, 10, 12, 16, 20, 24, 32, 48}) {
        std::vector<hipStream_t> st(n);
        for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        for (auto& s : st) hipLaunchKernelGGL(nap, dim3(1), dim3(64), 0, s, 1000ull);
        hipDeviceSynchronize();
        double best = 1e30;
        for (int rep = 0; rep < 3; ++rep) {
            const auto t0 = std::chrono::steady_clock::now();
            for (auto& s : st) hipLaunchKernelGGL(nap, dim3(1), dim3(64), 0, s, 200000ull);   // 2 ms
            hipDeviceSynchronize();
            best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        }
        std::printf("streams %2d  batch %.2f ms  => %.1f abreast\n", n, best * 1e3, n * 2.0 / (best * 1e3));
        for (auto& s : st) hipStreamDestroy(s);
    }
    return 0;
}
