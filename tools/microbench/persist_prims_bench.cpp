// Micro-benchmarks of the primitives the persistent cluster kernel is built from (MI355X): grid barrier, sc1 round
// trips, the 32 x 32 diagonal-block factorisation, the panel solve.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17
// -ffp-contract=on -x hip tools/microbench/persist_prims_bench.cpp -o build/persist_prims_bench ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "../../ipc_amd/csrc/cluster_persist.hpp"

using namespace ipc;

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(kPT, 1) void k_barrier(PersistCtl* ctl, int reps, unsigned long long* out, double* buf, int payload)
{
    extern __shared__ double lds[];
    GridBar gb{&ctl->bar, 0u, (int)gridDim.x, &ctl->error, nullptr};
    const int g = blockIdx.x, G = gridDim.x, tid = threadIdx.x;
    grid_barrier(gb);
    const unsigned long long t0 = wall_clock64();
    double acc = 0.0;
    for (int r = 0; r < reps; ++r) {
        for (int i = tid; i < payload; i += kPT) st_shared(&buf[(size_t)g * payload + i], (double)(r + i));
        grid_barrier(gb);
        const int src = (g + 1) % G;
        for (int i = tid; i < payload; i += kPT) acc += ld_shared(&buf[(size_t)src * payload + i]);
        if (payload) { lds[tid] = acc; __syncthreads(); }
    }
    if (g == 0 && tid == 0) out[0] = wall_clock64() - t0;
    if (acc == 123.456) out[1] = 1;
}

__global__ void k_chase(const unsigned long long* next, int steps, unsigned long long* out, int sc1)
{
    unsigned long long p = 0;
    const unsigned long long t0 = wall_clock64();
    if (sc1) for (int s = 0; s < steps; ++s) p = __hip_atomic_load(&next[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else for (int s = 0; s < steps; ++s) p = next[p];
    out[0] = wall_clock64() - t0;
    out[1] = p;
}

__global__ void k_store_ack(double* buf, int steps, unsigned long long* out)
{
    const unsigned long long t0 = wall_clock64();
    for (int s = 0; s < steps; ++s) {
        st_shared(&buf[(size_t)s * 64 + threadIdx.x], (double)s);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    out[0] = wall_clock64() - t0;
}

__global__ void k_atomic_rt(unsigned* ctr, int steps, unsigned long long* out)
{
    const unsigned long long t0 = wall_clock64();
    unsigned v = 0;
    for (int s = 0; s < steps; ++s) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v += __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    out[0] = wall_clock64() - t0;
    out[1] = v;
}

__global__ __launch_bounds__(kPT, 1) void k_potrf(const double* src, int reps, unsigned long long* out, double* sink)
{
    extern __shared__ double lds[];
    double (*Dn)[kCB + 1] = reinterpret_cast<double (*)[kCB + 1]>(lds);
    double* Dninv = lds + kCB * (kCB + 1);
    const int tid = threadIdx.x;
    unsigned long long tot = 0;
    for (int r = 0; r < reps; ++r) {
        for (int idx = tid; idx < kCB * kCB; idx += kPT) Dn[idx % kCB][idx / kCB] = src[idx];
        __syncthreads();
        const unsigned long long t0 = wall_clock64();
        if (tid < 64) potrf32_wave(Dn, Dninv);
        __syncthreads();
        tot += wall_clock64() - t0;
    }
    if (tid == 0) { out[0] = tot; sink[0] = Dn[5][3] + Dninv[7]; }
}

// panel solve of 64 rows against a factored 32 x 32 block (the serial part of chol_tile), one wave
__global__ __launch_bounds__(kPT, 1) void k_trsm(const double* src, int reps, unsigned long long* out, double* sink)
{
    extern __shared__ double lds[];
    double* DT = lds;
    double (*P)[65] = reinterpret_cast<double (*)[65]>(lds + 2048);
    const int tid = threadIdx.x, lane = tid & 63;
    for (int idx = tid; idx < kCB * kCB; idx += kPT) {
        const int c = idx >> 5, r = idx & 31;
        DT[idx] = r > c ? src[idx] * 0.01 : (r == c ? 0.9 : 0.0);
    }
    __syncthreads();
    unsigned long long tot = 0;
    for (int r = 0; r < reps; ++r) {
        const unsigned long long t0 = wall_clock64();
        if (tid < 64) {
            double x[kCB];
#pragma unroll
            for (int c = 0; c < kCB; ++c) x[c] = src[c * 64 + lane] + r;
            trsm32<false>(x, DT, [&](int p, double v) { P[p][lane] = v; });
        }
        __syncthreads();
        tot += wall_clock64() - t0;
    }
    if (tid == 0) { out[0] = tot; sink[0] = P[5][3]; }
}

int main()
{
    const int LDS = sizeof(double) * kLdsTotal;
    unsigned long long *d_out, h_out[2];
    PersistCtl* d_ctl;
    double* d_buf;
    CHK(hipMalloc(&d_out, 16));
    CHK(hipMalloc(&d_ctl, sizeof(PersistCtl)));
    CHK(hipMalloc(&d_buf, sizeof(double) * 64 * 65536));
    CHK(hipMemset(d_buf, 0, sizeof(double) * 64 * 65536));
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_barrier), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_potrf), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trsm), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    const int reps = 500;
    for (int payload : {0, 512, 4096}) {
        for (int G : {2, 4, 8, 16, 24, 40, 64, 128}) {
            CHK(hipMemset(d_ctl, 0, sizeof(PersistCtl)));
            hipLaunchKernelGGL(k_barrier, dim3(G), dim3(kPT), LDS, nullptr, d_ctl, reps, d_out, d_buf, payload);
            CHK(hipDeviceSynchronize());
            CHK(hipMemcpy(h_out, d_out, 16, hipMemcpyDeviceToHost));
            printf("barrier G=%3d payload=%5d doubles/WG : %.2f us per (store + barrier + load)\n", G, payload, h_out[0] * 0.01 / reps);
        }
    }
    {
        const int N = 1 << 16;
        std::vector<unsigned long long> nxt(N);
        for (int i = 0; i < N; ++i) nxt[i] = ((unsigned long long)i * 2654435761ull + 12345) % N;
        unsigned long long* d_n;
        CHK(hipMalloc(&d_n, sizeof(unsigned long long) * N));
        CHK(hipMemcpy(d_n, nxt.data(), sizeof(unsigned long long) * N, hipMemcpyHostToDevice));
        for (int sc1 : {0, 1, 0, 1}) {
            hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, nullptr, d_n, 2000, d_out, sc1);
            CHK(hipDeviceSynchronize());
            CHK(hipMemcpy(h_out, d_out, 16, hipMemcpyDeviceToHost));
            printf("dependent load chain (%s, 512 KB table): %.3f us per load\n", sc1 ? "sc1" : "plain", h_out[0] * 0.01 / 2000);
        }
    }
    hipLaunchKernelGGL(k_store_ack, dim3(1), dim3(64), 0, nullptr, d_buf, 2000, d_out);
    CHK(hipDeviceSynchronize());
    CHK(hipMemcpy(h_out, d_out, 16, hipMemcpyDeviceToHost));
    printf("sc1 store + s_waitcnt vmcnt(0): %.3f us\n", h_out[0] * 0.01 / 2000);
    {
        unsigned* d_c;
        CHK(hipMalloc(&d_c, 4));
        CHK(hipMemset(d_c, 0, 4));
        hipLaunchKernelGGL(k_atomic_rt, dim3(1), dim3(1), 0, nullptr, d_c, 2000, d_out);
        CHK(hipDeviceSynchronize());
        CHK(hipMemcpy(h_out, d_out, 16, hipMemcpyDeviceToHost));
        printf("atomic add (no return) + sc1 load of the same word: %.3f us\n", h_out[0] * 0.01 / 2000);
    }
    {
        std::vector<double> M(64 * 64);
        for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) M[c * 32 + r] = (r == c ? 40.0 : 0.0) + 1.0 / (1 + r + c);
        double* d_m;
        CHK(hipMalloc(&d_m, sizeof(double) * 64 * 64));
        CHK(hipMemcpy(d_m, M.data(), sizeof(double) * 64 * 64, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_potrf, dim3(1), dim3(kPT), LDS, nullptr, d_m, 200, d_out, d_buf);
        CHK(hipDeviceSynchronize());
        CHK(hipMemcpy(h_out, d_out, 16, hipMemcpyDeviceToHost));
        printf("potrf32_wave (+ 1 __syncthreads): %.3f us\n", h_out[0] * 0.01 / 200);
        hipLaunchKernelGGL(k_trsm, dim3(1), dim3(kPT), LDS, nullptr, d_m, 200, d_out, d_buf);
        CHK(hipDeviceSynchronize());
        CHK(hipMemcpy(h_out, d_out, 16, hipMemcpyDeviceToHost));
        printf("panel solve, 64 rows x 32 columns, one wave (+ loads, 1 __syncthreads): %.3f us\n", h_out[0] * 0.01 / 200);
    }
    return 0;
}
