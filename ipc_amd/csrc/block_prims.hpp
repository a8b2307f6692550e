// Workgroup-level primitives for the cell solvers (gfx950, wave64).
//
// A "cell" (one consistency sub-problem) is solved by one workgroup of T = 64*W threads;
// every reduction below returns the SAME bit pattern to every thread of the workgroup
// (fixed summation order), so that all threads take identical dog-leg branches.
#pragma once
#include <hip/hip_runtime.h>

namespace ipc {

// ---- wave64 sum via DPP (row_shr 1,2,4,8 + row_bcast15 + row_bcast31), result = lane 63 ----
// v_add_f64 has no DPP encoding on gfx9, so each step moves the two dwords with
// v_mov_b32_dpp (lanes without a source read 0) and adds.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_mov0(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, BANK_MASK, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, BANK_MASK, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double readlane63(double v)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// inclusive prefix sum over the 64 lanes of a wave (classic GCN DPP scan)
__device__ __forceinline__ double wave_inclusive_scan(double v)
{
    v += dpp_mov0<0x111, 0xf, 0xf>(v);           // row_shr:1
    v += dpp_mov0<0x112, 0xf, 0xf>(v);           // row_shr:2
    v += dpp_mov0<0x114, 0xf, 0xf>(v);           // row_shr:4
    v += dpp_mov0<0x118, 0xf, 0xf>(v);           // row_shr:8
    v += dpp_mov0<0x142, 0xa, 0xf>(v);           // row_bcast:15 -> rows 1,3
    v += dpp_mov0<0x143, 0xc, 0xf>(v);           // row_bcast:31 -> rows 2,3
    return v;
}

__device__ __forceinline__ double wave_sum(double v) { return readlane63(wave_inclusive_scan(v)); }

__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- workgroup sum of K doubles; red must hold W*K doubles -------------------------------
template <int W, int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double* red)
{
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
    if constexpr (W > 1) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        __syncthreads();                              // previous readers of red are done
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) red[wave * K + k] = v[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double s = red[k];
#pragma unroll
            for (int w = 1; w < W; ++w) s += red[w * K + k];
            v[k] = s;
        }
    }
}

template <int W>
__device__ __forceinline__ double block_max(double v, double* red)
{
    v = wave_max(v);
    if constexpr (W > 1) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        double s = red[0];
#pragma unroll
        for (int w = 1; w < W; ++w) s = fmax(s, red[w]);
        v = s;
    }
    return v;
}

template <int W>
__device__ __forceinline__ bool block_any(bool p, int* flag)
{
    bool any = __ballot(p) != 0ull;
    if constexpr (W > 1) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        __syncthreads();
        if (lane == 0) flag[wave] = any ? 1 : 0;
        __syncthreads();
        int s = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) s |= flag[w];
        any = s != 0;
    }
    return any;
}

// ---- workgroup exclusive prefix sums of K per-thread totals (thread order) ---------------
// On return excl[k] = sum of v[k] over threads with a smaller threadIdx.  red: W*K doubles.
template <int W, int K>
__device__ __forceinline__ void block_exclusive_scan(const double (&v)[K], double (&excl)[K], double* red)
{
    double inc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) inc[k] = wave_inclusive_scan(v[k]);
    // exclusive value by shifting (inc - v would not be the same floating-point sum)
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double sh = __shfl_up(inc[k], 1, 64);
        excl[k] = (threadIdx.x & 63) == 0 ? 0.0 : sh;
    }
    if constexpr (W > 1) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        __syncthreads();
        if (lane == 63) {
#pragma unroll
            for (int k = 0; k < K; ++k) red[wave * K + k] = inc[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double base = 0.0;
#pragma unroll
            for (int w = 0; w < W; ++w)
                if (w < wave) base += red[w * K + k];
            excl[k] += base;
        }
    }
}

}  // namespace ipc
