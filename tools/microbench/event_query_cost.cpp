// What one poll of the speculative pipeline costs: hipEventQuery on a pending / completed event against a read of a
// host-mapped flag.  hipcc -O2 --offload-arch=gfx950 -o /tmp/evq tools/microbench/event_query_cost.cpp && /tmp/evq
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_nap(unsigned long long ticks, volatile int* flag)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (flag) { __threadfence_system(); *flag = 1; }
}
int main()
{
    hipStream_t st[16];
    hipEvent_t ev[16];
    for (int i = 0; i < 16; ++i) { hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking); hipEventCreateWithFlags(&ev[i], hipEventDisableTiming); }
    int* hflag; hipHostMalloc(&hflag, 64 * 16, hipHostMallocMapped);
    int* dflag; hipHostGetDevicePointer((void**)&dflag, hflag, 0);
    for (int i = 0; i < 16; ++i) { hipLaunchKernelGGL(k_nap, dim3(1), dim3(64), 0, st[i], 100ull, nullptr); }
    hipDeviceSynchronize();
    // pending events: 16 kernels napping 50 ms
    for (int i = 0; i < 16; ++i) { hflag[16 * i] = 0; hipLaunchKernelGGL(k_nap, dim3(1), dim3(64), 0, st[i], 5000000ull, dflag + 16 * i); hipEventRecord(ev[i], st[i]); }
    auto t0 = std::chrono::steady_clock::now();
    int rounds = 0, pend = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 0.03) { for (int i = 0; i < 16; ++i) pend += hipEventQuery(ev[i]) == hipErrorNotReady; ++rounds; }
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("hipEventQuery on a pending event: %.2f us per call (%d of %d pending)\n", 1e6 * dt / (16.0 * rounds), pend, 16 * rounds);
    t0 = std::chrono::steady_clock::now(); rounds = 0; long seen = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 0.01) { for (int i = 0; i < 16; ++i) seen += ((volatile int*)hflag)[16 * i]; ++rounds; }
    dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("host-mapped flag read: %.3f us per read\n", 1e6 * dt / (16.0 * rounds));
    // latency from the kernel's flag store to the host seeing it / to the event turning ready
    hipDeviceSynchronize();
    t0 = std::chrono::steady_clock::now(); rounds = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 0.01) { for (int i = 0; i < 16; ++i) pend += hipEventQuery(ev[i]) == hipErrorNotReady; ++rounds; }
    dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("hipEventQuery on a completed event: %.2f us per call\n", 1e6 * dt / (16.0 * rounds));
    double lat_flag = 0, lat_ev = 0;
    for (int rep = 0; rep < 20; ++rep) {
        hflag[0] = 0;
        hipLaunchKernelGGL(k_nap, dim3(1), dim3(64), 0, st[0], 20000ull, dflag);   // 0.2 ms
        hipEventRecord(ev[0], st[0]);
        std::chrono::steady_clock::time_point tf, te;
        bool f = false, e = false;
        while (!f || !e) {
            if (!f && ((volatile int*)hflag)[0]) { tf = std::chrono::steady_clock::now(); f = true; }
            if (!e && hipEventQuery(ev[0]) == hipSuccess) { te = std::chrono::steady_clock::now(); e = true; }
        }
        lat_ev += std::chrono::duration<double>(te - tf).count();
    }
    printf("event ready after the flag was seen: %.1f us later (mean of 20)\n", 1e6 * lat_ev / 20);
    (void)lat_flag;
    return 0;
}
