// Dense SPD solve for the capacitance system of a cluster (NS = d * #loops unknowns).
// Right-looking blocked Cholesky, column major, lower triangle.  The right-hand side rides
// along as an extra ROW (row n) of the (n+1) x n array, so the forward substitution falls out of
// the factorisation itself (row n of the factor is y = L^-1 d); only L^T x = y is a separate pass.
// Sizes are small (tens .. a few hundred unknowns) and one factorisation sits on the critical path of
// every dog-leg iteration of the incremental mode, so the kernels are built for LATENCY: one launch
// per block column (the diagonal block is factored redundantly by every workgroup, each trailing tile
// solves the panel rows it needs itself), the factor goes to a second array so that no workgroup
// reads what another one writes, and the triangular kernels run in registers with v_readlane.
#pragma once
#include <hip/hip_runtime.h>

#include "block_prims.hpp"

namespace ipc {

constexpr int kCB = 32;      // block-column width

// 1 / sqrt(x) to full precision: v_rsq_f64 + two Newton steps
__device__ __forceinline__ double rsqrt_newton(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    y = y * fma(-0.5 * x * y, y, 1.5);
    y = y * fma(-0.5 * x * y, y, 1.5);
    return y;
}

// ---- the 64 x 64 x 32 tile product of the trailing update: C = Ai^T Aj (panels [32][65] in LDS) ------------------
// The one dense contraction of the repo.  Two forms, same result bits (each C element is the FMA chain over p = 0 .. 31
// starting from 0, which is also what the FP64 matrix core computes: v_mfma_f64_16x16x4_f64 adds its four products to
// the accumulator one after the other, in k order, each step one fused multiply-add -- tests/test_gpu_persistent.py
// holds the two forms against each other bit for bit):
//   IPC_TILE_MFMA 0: 4 x 4 register block per thread of plain v_fma_f64 (rounds 2-3): 8 LDS reads per 16 FMAs;
//   IPC_TILE_MFMA 1: v_mfma_f64_16x16x4_f64, wave w of the 256-thread group owns the 2 x 2 blocks of 16 x 16 with block
//                    rows j in {2 (w >> 1), +1} and block columns i in {2 (w & 1), +1}: 4 LDS reads per 4 MFMAs
//                    (= 64 FMAs per lane-equivalent), 8 k-steps.  The j index is the MFMA's row (A operand) so that the
//                    result's fast lane index (lane & 15) runs along i, the contiguous direction of the column-major
//                    matrix: lane l, register q of block (jj, ii) holds C[i = 16 bi + (l & 15)][j = 16 bj + (l >> 4) + 4 q].
// MI355X's FP64 matrix rate equals its FP64 vector rate (78.6 TFLOP/s), so the gain is issue slots and LDS operand
// traffic, not peak.
#ifndef IPC_TILE_MFMA
#define IPC_TILE_MFMA 1
#endif
typedef double tile_d4 __attribute__((ext_vector_type(4)));
struct TileOwn {                                     // which elements of the tile a thread holds, in the order e = 0 .. 15
    int wave, lane;
    __device__ __forceinline__ int i_of(int e) const
    {
#if IPC_TILE_MFMA
        return 16 * (2 * (wave & 1) + ((e >> 2) & 1)) + (lane & 15);
#else
        return (lane & 15) + 16 * (e & 3);           // tx + 16 a   (thread t of the group: tx = t & 15, ty = t >> 4)
#endif
    }
    __device__ __forceinline__ int j_of(int e) const
    {
#if IPC_TILE_MFMA
        return 16 * (2 * (wave >> 1) + (e >> 3)) + (lane >> 4) + 4 * (e & 3);
#else
        return (wave * 4 + (lane >> 4)) + 16 * (e >> 2);          // ty + 16 b
#endif
    }
};
// acc[e] = sum_p Ai[p][i_of(e)] * Aj[p][j_of(e)]; every lane of the four waves must be active (MFMA)
__device__ __forceinline__ void tile_product(const double (*Ai)[64 + 1], const double (*Aj)[64 + 1], const TileOwn& own, double (&acc)[16])
{
#if IPC_TILE_MFMA
    const int kq = own.lane >> 4, c = own.lane & 15;
    const int i0 = 16 * 2 * (own.wave & 1) + c, j0 = 16 * 2 * (own.wave >> 1) + c;
    tile_d4 d[2][2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) d[jj][ii] = tile_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < kCB / 4; ++ks) {
        const int p = 4 * ks + kq;
        const double a0 = Aj[p][j0], a1 = Aj[p][j0 + 16], b0 = Ai[p][i0], b1 = Ai[p][i0 + 16];
        d[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, d[0][0], 0, 0, 0);
        d[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, d[0][1], 0, 0, 0);
        d[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, d[1][0], 0, 0, 0);
        d[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, d[1][1], 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = d[e >> 3][(e >> 2) & 1][e & 3];
#else
    const int tx = own.lane & 15, ty = own.wave * 4 + (own.lane >> 4);
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0;
#pragma unroll 8
    for (int p = 0; p < kCB; ++p) {
        double av[4], bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { av[q] = Ai[p][tx + 16 * q]; bv[q] = Aj[p][ty + 16 * q]; }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a + 4 * b] = fma(av[a], bv[b], acc[a + 4 * b]);
    }
#endif
}

// One block column k0 .. k0+nb of the factorisation, one workgroup per 64 x 64 tile (bx >= by) of the trailing
// matrix (rows k1 .. n, row n = right-hand side; columns k1 .. n-1):
//   1. every workgroup loads the nb x nb diagonal block and its first wave factors it in registers (lane r holds
//      row r; column c reaches the other rows through v_readlane, so the 32 elimination steps are straight-line
//      code without LDS round trips);
//   2. waves 0 / 1 solve the 64 panel rows of the tile's row block / column block against it (one row per lane);
//   3. the tile is updated, C -= Ai Aj^T (4 x 4 per thread);
//   4. tiles of the first tile column (by == 0) write their panel rows to the factor Lf, tile (0, 0) the
//      diagonal block as well.
// A (trailing matrix, updated in place tile by tile) and Lf (factor, written once) are different arrays.
__global__ __launch_bounds__(256) void chol_step(double* A, double* Lf, int n, int ld, int k0, int nb, int* info)
{
    __shared__ double D[kCB][kCB + 1];
    __shared__ double Dinv[kCB];
    __shared__ double Ai[kCB][64 + 1], Aj[kCB][64 + 1];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int k1 = k0 + nb;
    const int i0 = k1 + blockIdx.x * 64, j0 = k1 + blockIdx.y * 64;
    if (blockIdx.y > blockIdx.x) return;
    for (int idx = tid; idx < kCB * kCB; idx += 256) {
        const int r = idx % kCB, c = idx / kCB;
        D[r][c] = (r < nb && c < nb && r >= c) ? A[(size_t)(k0 + c) * ld + k0 + r] : (r == c ? 1.0 : 0.0);
    }
    // panel rows of this tile (original values), one row per lane: waves 0 / 1 keep them in registers
    double x[kCB];
    const int prow = wave == 0 ? i0 + lane : j0 + lane;
    const bool pvalid = wave == 0 ? prow <= n : prow < n;          // row n (rhs) only ever is a tile ROW
    if (wave < 2) {
#pragma unroll
        for (int c = 0; c < kCB; ++c) x[c] = (c < nb && pvalid) ? A[(size_t)(k0 + c) * ld + prow] : 0.0;
    }
    __syncthreads();
    bool ok = true;
    if (wave == 0) {
        // rows >= nb are identity rows
        const int r = lane & 31;
        double row[kCB];
#pragma unroll
        for (int c = 0; c < kCB; ++c) row[c] = D[r][c];
#pragma unroll
        for (int c = 0; c < kCB; ++c) {
            const double piv = read_lane(row[c], c);
            if (!(piv > 0)) ok = false;
            const double inv = rsqrt_newton(piv);
            const double lrc = r == c ? piv * inv : (r > c ? row[c] * inv : 0.0);
            row[c] = lrc;
            if (lane == c) Dinv[c] = inv;
#pragma unroll
            for (int cc = c + 1; cc < kCB; ++cc) {
                const double lcc = read_lane(lrc, cc);       // L[cc][c]
                // (rows r < cc pick up garbage above the diagonal, which nothing reads: lrc is forced to 0 for
                // r < c, the pivot is read from lane c, and only the lower triangle is written back)
                row[cc] = fma(-lrc, lcc, row[cc]);
            }
        }
        if (lane < kCB) {
#pragma unroll
            for (int c = 0; c < kCB; ++c) D[r][c] = row[c];
        }
    }
    __syncthreads();
    if (wave < 2) {
        double (*P)[64 + 1] = wave == 0 ? Ai : Aj;
#pragma unroll
        for (int c = 0; c < kCB; ++c) {
            double v = x[c];
#pragma unroll
            for (int p = 0; p < c; ++p) v -= x[p] * D[c][p];
            x[c] = v * Dinv[c];
            // Store at once: the LDS reads of the unrolled triangle are ordered memory operations, the
            // arithmetic is not -- left free it sinks towards one block of stores at the end, every
            // loaded operand stays live and the kernel spills.  A store per column pins the arithmetic.
            P[c][lane] = c < nb ? x[c] : 0.0;
            if (wave == 0 && blockIdx.y == 0 && c < nb && pvalid) Lf[(size_t)(k0 + c) * ld + prow] = x[c];
        }
    }
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        for (int idx = tid; idx < kCB * kCB; idx += 256) {
            const int r = idx % kCB, c = idx / kCB;
            if (r < nb && c < nb && r >= c) Lf[(size_t)(k0 + c) * ld + k0 + r] = D[r][c];
        }
        // info: 0 = positive definite so far, else 1 + the first block column with a non-positive pivot
        if (tid == 0) { if (k0 == 0) *info = ok ? 0 : 1; else if (!ok && *info == 0) *info = k0 + 1; }
    }
    __syncthreads();
    if (j0 >= n) return;                                    // (a tile column of right-hand-side rows only)
    const TileOwn own{wave, lane};
    double acc[16];
    tile_product(Ai, Aj, own, acc);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = i0 + own.i_of(e), j = j0 + own.j_of(e);
        if (j < n && i <= n && i >= j) A[(size_t)j * ld + i] -= acc[e];
    }
}

// L^T x = y with y = row n of the factor; one workgroup.  Per block column (from the last one): the part of the
// sums that involves already solved unknowns is a reduction over rows (16 waves, coalesced column reads), the
// nb x nb triangle is solved in registers by the first wave (constant-lane v_readlane, no LDS round trips).
__global__ __launch_bounds__(1024) void chol_backsolve(const double* Lf, int n, int ld, double* x)
{
    __shared__ double D[kCB][kCB + 1];
    __shared__ double t[kCB];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nblk = (n + kCB - 1) / kCB;
    for (int kb = nblk - 1; kb >= 0; --kb) {
        const int k0 = kb * kCB, nb = min(kCB, n - k0), k1 = k0 + nb;
        for (int c = wave; c < nb; c += 16) {
            double acc = 0.0;
            for (int r = k1 + lane; r < n; r += 64) acc += Lf[(size_t)(k0 + c) * ld + r] * x[r];
            acc = wave_sum(acc);
            if (lane == 0) t[c] = acc;
        }
        for (int idx = tid; idx < kCB * kCB; idx += 1024) {
            const int r = idx % kCB, c = idx / kCB;
            D[r][c] = (r < nb && c < nb && r >= c) ? Lf[(size_t)(k0 + c) * ld + k0 + r] : (r == c ? 1.0 : 0.0);
        }
        __syncthreads();
        if (wave == 0) {
            const int l = lane & 31;
            double v = l < nb ? Lf[(size_t)(k0 + l) * ld + n] - t[l] : 0.0;
            const double dinv = 1.0 / D[l][l];              // one division per lane, not per step
            double col[kCB];                                // col[r] = L[r][l]: what unknown r takes away from row l
#pragma unroll
            for (int r = 0; r < kCB; ++r) col[r] = D[r][l];
#pragma unroll
            for (int r = kCB - 1; r >= 0; --r) {
                const double xr = read_lane(v, r) * read_lane(dinv, r);
                v = l == r ? xr : (l < r ? fma(-col[r], xr, v) : v);
            }
            if (lane < nb) x[k0 + lane] = v;
        }
        __threadfence_block();
        __syncthreads();
    }
}

// factor + solve; A is the (n+1) x n system (destroyed), Lf a second array of the same size that receives the
// factor; the solution lands in x[0..n)
inline hipError_t chol_solve_device(double* A, double* Lf, int n, double* x, int* d_info, hipStream_t st)
{
    const int ld = n + 1;
    for (int k0 = 0; k0 < n; k0 += kCB) {
        const int nb = n - k0 < kCB ? n - k0 : kCB, k1 = k0 + nb;
        const int nti = (n + 1 - k1 + 63) / 64, ntj = (n - k1 + 63) / 64;
        hipLaunchKernelGGL(chol_step, dim3(nti, ntj > 0 ? ntj : 1), dim3(256), 0, st, A, Lf, n, ld, k0, nb, d_info);
    }
    hipLaunchKernelGGL(chol_backsolve, dim3(1), dim3(1024), 0, st, (const double*)Lf, n, ld, x);
    return hipGetLastError();
}

}  // namespace ipc
