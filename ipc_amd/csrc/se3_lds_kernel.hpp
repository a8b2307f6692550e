// Persistent kernels of the LDS-pose SE(3) cell solver (se3_lds_cell.hpp): workgroups of four waves
// (one per SIMD, up to 512 registers each) take cells from a work queue -- four independent cells at
// a time (W = 1) or one cell on the whole workgroup (W = 4).
#pragma once
#include <algorithm>

#include "cell_kernels.hpp"
#include "se3_lds_cell.hpp"

namespace ipc {

// Teams per workgroup: four waves (one per SIMD, up to 512 registers each) by default.  The short-chain variants
// (one wave per cell, M <= 3) need fewer than 256 registers and 9 - 16 KB of LDS per team, so eight teams share a CU:
// two waves per SIMD, each hiding the other's LDS / L2 / transcendental latencies.
#ifndef IPC_SE3_DENSE_MAXM
#define IPC_SE3_DENSE_MAXM 1
#endif
template <int W, int M>
constexpr int se3_lds_teams() { return (W == 1 && M <= IPC_SE3_DENSE_MAXM) ? 8 : 4 / W; }

template <int W, int M, int NL>
__global__ __launch_bounds__((64 * W * se3_lds_teams<W, M>()), 1) void se3_lds_kernel(Se3View P, const int2* cells, int ncells,
                                                                                   unsigned* counter, SolveParams prm, CellOut out)
{
    extern __shared__ double2 dyn_lds2[];
    using T = Se3Lds<W, M, NL>;
    constexpr size_t TS = (sizeof(T) + 15) / 16;      // team stride in double2 units
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, team = wave / W;
    T& sh = *reinterpret_cast<T*>(dyn_lds2 + team * TS);
    for (;;) {
        unsigned c = 0;
        if constexpr (W == 1) {
            if (lane == 0) c = atomicAdd(counter, 1u);
            c = (unsigned)__builtin_amdgcn_readfirstlane((int)c);
        } else {
            if (threadIdx.x == 0) sh.cell = (int)atomicAdd(counter, 1u);
            __syncthreads();
            c = (unsigned)__builtin_amdgcn_readfirstlane(sh.cell);
        }
        if (c >= (unsigned)ncells) break;
        const int2 cc = cells[c];
        int cand[2] = {cc.x, cc.y};
        int lo = min(P.cand_from[cc.x], P.cand_to[cc.x]), hi = max(P.cand_from[cc.x], P.cand_to[cc.x]);
        if (NL == 2) {
            lo = min(lo, min(P.cand_from[cc.y], P.cand_to[cc.y]));
            hi = max(hi, max(P.cand_from[cc.y], P.cand_to[cc.y]));
        }
        const int L = hi - lo;
        const int base = NL == 1 ? prm.fast_iter : prm.slow_iter;
        const int iterations = (L + NL > 100) ? base * 5 : base;       // consensus_utils.cpp:12-13
        CellResult3 r;
        se3_lds_solve<W, M, NL>(P, lo, L, cand, iterations, sh, r);
        if ((W == 1 || wave == 0) && lane == 0) {
            out.max_chi2[c] = r.max_chi2;
            out.chi2_total[c] = r.chi2_total;
            out.meta[c] = make_int4(r.iterations, r.tries, r.flags, r.evals);
        }
        if constexpr (W == 1) wave_sync3();
        else __syncthreads();                          // the team's LDS is reused by the next cell
    }
}

template <int W, int M, int NL>
static hipError_t launch_se3_lds_one(int n, hipStream_t st, const Se3View& P, const int2* cells, SolveParams prm, CellOut out,
                                     unsigned* counter, int n_cu)
{
    using T = Se3Lds<W, M, NL>;
    constexpr int kTeams = se3_lds_teams<W, M>();
    constexpr size_t kBytes = kTeams * ((sizeof(T) + 15) / 16) * 16;
    static_assert(kBytes <= 160 * 1024, "team state exceeds the CU's LDS");
    auto k = se3_lds_kernel<W, M, NL>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBytes);
    if (e != hipSuccess) return e;
    const int groups = std::max(1, std::min(n_cu, (n + kTeams - 1) / kTeams));
    hipLaunchKernelGGL(k, dim3(groups), dim3(64 * W * kTeams), kBytes, st, P, cells, n, counter, prm, out);
    return hipGetLastError();
}

#define IPC_SE3_LDS_UNIT(WW, MM)                                                                                       \
    namespace ipc {                                                                                                    \
    hipError_t launch_se3_lds_##WW##_##MM(int nl, int n, hipStream_t st, const Se3View& P, const int2* cells,          \
                                          SolveParams prm, CellOut out, unsigned* counter, int n_cu)                   \
    {                                                                                                                  \
        return nl == 1 ? launch_se3_lds_one<WW, MM, 1>(n, st, P, cells, prm, out, counter, n_cu)                       \
                       : launch_se3_lds_one<WW, MM, 2>(n, st, P, cells, prm, out, counter, n_cu);                      \
    }                                                                                                                  \
    }

}  // namespace ipc
