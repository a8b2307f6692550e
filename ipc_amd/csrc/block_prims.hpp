// Wave-level primitives for the cell solvers (gfx950, wave64).
//
// A "cell" (one consistency sub-problem) is solved by one workgroup of W waves.  Reductions
// are built so that every thread of the workgroup ends up with the SAME bit pattern (fixed
// summation order) and therefore takes identical dog-leg branches.
//
// FP64 has no DPP encoding on gfx9 (v_add_f64 is VOP3), so cross-lane moves are done on the two
// dwords with v_mov_b32_dpp / v_permlane*_swap and the arithmetic stays a plain v_add_f64.
#pragma once
#include <hip/hip_runtime.h>

namespace ipc {

__device__ __forceinline__ double mk_double(int hi, int lo) { return __hiloint2double(hi, lo); }

// Marks a value as wave-uniform.  Pinning such values into SGPRs with v_readfirstlane was
// tried and abandoned: with the kernel already SGPR-bound, ROCm 7.2's hipcc produced a
// register-allocation-dependent miscompile (wrong dog-leg decisions on single cells that
// disappeared under any perturbation of the code), so this is deliberately the identity.
__device__ __forceinline__ double uni(double v) { return v; }
__device__ __forceinline__ double read_lane(double v, int l)
{
    return mk_double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                     __builtin_amdgcn_readlane(__double2loint(v), l));
}

template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_mov(double v, double old)
{
    int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROW_MASK, BANK_MASK, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROW_MASK, BANK_MASK, false);
    return mk_double(hi, lo);
}

// value of lane-1 (wave_shr:1); lane 0 receives `carry`
__device__ __forceinline__ double lane_prev(double v, double carry) { return dpp_mov<0x138, 0xf, 0xf>(v, carry); }
// value of lane+1 (wave_shl:1); lane 63 receives `carry`
__device__ __forceinline__ double lane_next(double v, double carry) { return dpp_mov<0x130, 0xf, 0xf>(v, carry); }

// inclusive prefix sum inside each row of 16 lanes (row_shr 1,2,4,8; lanes without a source add 0)
__device__ __forceinline__ double row_inclusive_scan(double v)
{
    v += dpp_mov<0x111, 0xf, 0xf>(v, 0.0);
    v += dpp_mov<0x112, 0xf, 0xf>(v, 0.0);
    v += dpp_mov<0x114, 0xf, 0xf>(v, 0.0);
    v += dpp_mov<0x118, 0xf, 0xf>(v, 0.0);
    return v;
}
// inclusive prefix sum over the 64 lanes of a wave
__device__ __forceinline__ double wave_inclusive_scan(double v)
{
    v = row_inclusive_scan(v);
    v += dpp_mov<0x142, 0xa, 0xf>(v, 0.0);       // row_bcast:15 -> rows 1,3
    v += dpp_mov<0x143, 0xc, 0xf>(v, 0.0);       // row_bcast:31 -> rows 2,3
    return v;
}
__device__ __forceinline__ double wave_sum(double v) { return read_lane(wave_inclusive_scan(v), 63); }

__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- packed reductions (gfx950 v_permlane32_swap / v_permlane16_swap) ----------------------
// After pair32(a,b): lanes 0-31 hold a[l] + a[l+32], lanes 32-63 hold b[l-32] + b[l].
__device__ __forceinline__ double pair32(double a, double b)
{
    auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    return mk_double(hi[0], lo[0]) + mk_double(hi[1], lo[1]);
}
// After pair16(a,b): row0 = a.row0+a.row1, row1 = b.row0+b.row1, row2 = a.row2+a.row3,
// row3 = b.row2+b.row3 (rows of 16 lanes, element-wise).
__device__ __forceinline__ double pair16(double a, double b)
{
    auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    return mk_double(hi[0], lo[0]) + mk_double(hi[1], lo[1]);
}
// Wave sums of 16 per-lane values: dst[k] = sum over the 64 lanes of v[k] (written by one lane
// each; dst is LDS or global).  ~100 VALU ops instead of 16 x 20 for separate DPP reductions.
__device__ __forceinline__ void wave_sum16_store(const double (&v)[16], double* dst)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // value index after the two packing levels: row r of q holds v[i + 4 r]
        const double r0 = pair32(v[i], v[i + 8]);          // [v_i | v_{i+8}]
        const double r1 = pair32(v[i + 4], v[i + 12]);     // [v_{i+4} | v_{i+12}]
        double q = pair16(r0, r1);                         // rows: v_i, v_{i+4}, v_{i+8}, v_{i+12}
        q = row_inclusive_scan(q);
        if ((lane & 15) == 15) dst[i + 4 * (lane >> 4)] = q;
    }
}

// After a barrier: totals of K (<= 4) values over W (<= 16) waves from red[k*16 + w]
// (row k of 16 lanes sums wave partials; unused entries must read as zero => caller zero-pads
// by passing nw), returned uniformly.
template <int K>
__device__ __forceinline__ void gather_totals(const double* red, int nw, double (&tot)[K])
{
    const int lane = threadIdx.x & 63;
    const int k = lane >> 4, w = lane & 15;
    double v = (k < K && w < nw) ? red[k * 16 + w] : 0.0;
    v = row_inclusive_scan(v);
#pragma unroll
    for (int q = 0; q < K; ++q) tot[q] = read_lane(v, 16 * q + 15);
}

}  // namespace ipc
