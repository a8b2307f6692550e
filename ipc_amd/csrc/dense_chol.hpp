// Dense SPD solve for the capacitance system of a cluster (NS = d * #loops unknowns).
// Right-looking blocked Cholesky, column major, lower triangle.  The right-hand side rides
// along as an extra ROW (row n) of the (n+1) x n array, so the forward substitution falls out of
// the factorisation itself (row n of the factor is y = L^-1 d); only L^T x = y is a separate pass.
// Sizes are small (tens .. a few thousand), so the kernels favour few launches and coalesced
// column accesses over peak FLOPs.
#pragma once
#include <hip/hip_runtime.h>

#include "block_prims.hpp"

namespace ipc {

constexpr int kCB = 32;      // block-column width

// Block column k0: factor the diagonal block (every workgroup redundantly, in LDS) and solve the
// panel rows below it, 64 rows per workgroup (workgroup 0 writes the diagonal block back).
__global__ __launch_bounds__(64) void chol_panel(double* A, int n, int ld, int k0, int* info)
{
    __shared__ double D[kCB][kCB + 1];
    __shared__ double Dinv[kCB];
    const int lane = threadIdx.x;
    const int nb = min(kCB, n - k0);
    for (int idx = lane; idx < kCB * kCB; idx += 64) {
        const int r = idx % kCB, c = idx / kCB;
        D[r][c] = (r < nb && c < nb && r >= c) ? A[(size_t)(k0 + c) * ld + k0 + r] : (r == c ? 1.0 : 0.0);
    }
    __syncthreads();
    // Factor the diagonal block in registers: lane r holds row r; column c of the factor reaches the
    // other rows through v_readlane (constant lane), so the 32 elimination steps are straight-line
    // code without LDS round trips.  Rows >= nb are identity rows.
    bool ok = true;
    {
        const int r = lane & 31;
        double row[kCB];
#pragma unroll
        for (int c = 0; c < kCB; ++c) row[c] = D[r][c];
#pragma unroll
        for (int c = 0; c < kCB; ++c) {
            const double piv = read_lane(row[c], c);
            if (!(piv > 0)) ok = false;
            const double d = sqrt(piv), inv = 1.0 / d;
            const double lrc = r == c ? d : (r > c ? row[c] * inv : 0.0);
            row[c] = lrc;
#pragma unroll
            for (int cc = c + 1; cc < kCB; ++cc) {
                const double lcc = read_lane(lrc, cc);       // L[cc][c]
                if (r >= cc) row[cc] = fma(-lrc, lcc, row[cc]);
            }
        }
        __syncthreads();
        if (lane < kCB) {
#pragma unroll
            for (int c = 0; c < kCB; ++c) D[r][c] = row[c];
        }
        // reciprocal of the diagonal, one division per lane instead of one per row and column
#pragma unroll
        for (int c = 0; c < kCB; ++c) {
            const double dcc = read_lane(row[c], c);
            if (lane == c) Dinv[c] = 1.0 / dcc;
        }
        __syncthreads();
    }
    if (blockIdx.x == 0) {
        for (int idx = lane; idx < kCB * kCB; idx += 64) {
            const int r = idx % kCB, c = idx / kCB;
            if (r < nb && c < nb && r >= c) A[(size_t)(k0 + c) * ld + k0 + r] = D[r][c];
        }
        if (!ok && lane == 0) *info = k0 + 1;
        return;
    }
    const int row = k0 + nb + (blockIdx.x - 1) * 64 + lane;
    if (row > n) return;                                   // row n = right-hand side
    double x[kCB];
#pragma unroll
    for (int c = 0; c < kCB; ++c) x[c] = c < nb ? A[(size_t)(k0 + c) * ld + row] : 0.0;
#pragma unroll
    for (int c = 0; c < kCB; ++c) {
        double v = x[c];
#pragma unroll
        for (int p = 0; p < c; ++p) v -= x[p] * D[c][p];
        x[c] = v * Dinv[c];
        // Store at once: the LDS reads of the unrolled triangle are ordered memory operations, the
        // arithmetic is not -- left free it sinks towards one block of stores at the end, every
        // loaded operand stays live and the kernel spills 600 registers.  A store per column pins
        // the arithmetic of that column in place.
        if (c < nb) A[(size_t)(k0 + c) * ld + row] = x[c];
    }
}

// Trailing update C[i][j] -= sum_p A[i][k0+p] A[j][k0+p] for k1 <= j <= i, i <= n (row n = rhs),
// j < n; 64 x 64 tile per workgroup, 4 x 4 per thread.
__global__ __launch_bounds__(256) void chol_update(double* A, int n, int ld, int k0, int nb)
{
    if (blockIdx.y > blockIdx.x) return;
    __shared__ double Ai[kCB][64], Aj[kCB][64];
    const int k1 = k0 + nb;
    const int i0 = k1 + blockIdx.x * 64, j0 = k1 + blockIdx.y * 64;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < kCB * 64; idx += 256) {
        const int p = idx >> 6, r = idx & 63;
        Ai[p][r] = (p < nb && i0 + r <= n) ? A[(size_t)(k0 + p) * ld + i0 + r] : 0.0;
        Aj[p][r] = (p < nb && j0 + r < n) ? A[(size_t)(k0 + p) * ld + j0 + r] : 0.0;
    }
    __syncthreads();
    const int tx = tid & 15, ty = tid >> 4;
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
#pragma unroll 8
    for (int p = 0; p < kCB; ++p) {
        double av[4], bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { av[q] = Ai[p][tx + 16 * q]; bv[q] = Aj[p][ty + 16 * q]; }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = fma(av[a], bv[b], acc[a][b]);
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int j = j0 + ty + 16 * b;
        if (j >= n) continue;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int i = i0 + tx + 16 * a;
            if (i <= n && i >= j) A[(size_t)j * ld + i] -= acc[a][b];
        }
    }
}

// L^T x = y with y = row n of the factored array; one workgroup
__global__ __launch_bounds__(1024) void chol_backsolve(const double* A, int n, int ld, double* x)
{
    __shared__ double D[kCB][kCB + 1];
    __shared__ double t[kCB];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nblk = (n + kCB - 1) / kCB;
    for (int kb = nblk - 1; kb >= 0; --kb) {
        const int k0 = kb * kCB, nb = min(kCB, n - k0), k1 = k0 + nb;
        for (int c = wave; c < nb; c += 16) {
            double acc = 0.0;
            for (int r = k1 + lane; r < n; r += 64) acc += A[(size_t)(k0 + c) * ld + r] * x[r];
            acc = wave_sum(acc);
            if (lane == 0) t[c] = acc;
        }
        for (int idx = tid; idx < kCB * kCB; idx += 1024) {
            const int r = idx % kCB, c = idx / kCB;
            D[r][c] = (r < nb && c < nb && r >= c) ? A[(size_t)(k0 + c) * ld + k0 + r] : 0.0;
        }
        __syncthreads();
        if (wave == 0) {
            double v = lane < nb ? A[(size_t)(k0 + lane) * ld + n] - t[lane] : 0.0;
            const double dinv = lane < nb ? 1.0 / D[lane][lane] : 0.0;      // one division per lane, not per step
            for (int r = nb - 1; r >= 0; --r) {
                const double xr = __shfl(v, r, 64) * __shfl(dinv, r, 64);
                if (lane == r) v = xr;
                else if (lane < r) v -= D[r][lane] * xr;
            }
            if (lane < nb) x[k0 + lane] = v;
        }
        __threadfence_block();
        __syncthreads();
    }
}

// factor + solve; the solution lands in x[0..n)
inline hipError_t chol_solve_device(double* A, int n, double* x, int* d_info, hipStream_t st)
{
    const int ld = n + 1;
    hipError_t e = hipMemsetAsync(d_info, 0, sizeof(int), st);
    if (e != hipSuccess) return e;
    for (int k0 = 0; k0 < n; k0 += kCB) {
        const int nb = n - k0 < kCB ? n - k0 : kCB, k1 = k0 + nb;
        const int rows = n + 1 - k1;
        hipLaunchKernelGGL(chol_panel, dim3(1 + (rows + 63) / 64), dim3(64), 0, st, A, n, ld, k0, d_info);
        const int nti = (n + 1 - k1 + 63) / 64, ntj = (n - k1 + 63) / 64;
        if (ntj > 0) hipLaunchKernelGGL(chol_update, dim3(nti, ntj), dim3(256), 0, st, A, n, ld, k0, nb);
    }
    hipLaunchKernelGGL(chol_backsolve, dim3(1), dim3(1024), 0, st, (const double*)A, n, ld, x);
    return hipGetLastError();
}

}  // namespace ipc
