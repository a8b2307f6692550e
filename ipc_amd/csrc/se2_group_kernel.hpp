// Group kernels of the SE(2) cell solver (se2_wave_cell.hpp with W = 2 or 4): persistent
// workgroups of four waves; W cooperating waves solve one cell at a time from a shared work queue
// (W = 2: two independent pairs per workgroup; W = 4: the whole workgroup on one cell).
#pragma once
#include "cell_kernels.hpp"
#include "se2_wave_cell.hpp"

namespace ipc {

constexpr int kWavesPerGroup = 4;
#ifndef IPC_LDS_BUDGET
#define IPC_LDS_BUDGET (160 * 1024)
#endif
constexpr int kLdsBudget = IPC_LDS_BUDGET;

// The four waves of a workgroup form 4 / W teams, each team solves one cell together (wave w of the
// team owns poses 64 M w + 1 .. 64 M (w + 1)); the teams are independent and share the staged chain
// constants.
template <int W, int M, int NL, bool STAGED>
__global__ __launch_bounds__(64 * kWavesPerGroup, 1) void se2_group_kernel(Se2View P, const int2* cells, int ncells,
                                                                           unsigned* counter, SolveParams prm,
                                                                           CellOut out, int wlo, int wlen, int wstride)
{
    extern __shared__ double dyn_lds[];
    constexpr int kPairs = kWavesPerGroup / W;
    constexpr int kScratchDoubles = (sizeof(WaveScratch<NL>) + 7) / 8;
    constexpr int kBoxDoubles = (sizeof(PairBox) + 7) / 8;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, pair = wave / W, wsub = wave & (W - 1);
    WaveScratch<NL>& sh = *reinterpret_cast<WaveScratch<NL>*>(dyn_lds + pair * kScratchDoubles);
    PairBox* box = reinterpret_cast<PairBox*>(dyn_lds + kPairs * kScratchDoubles + pair * kBoxDoubles);
    double* cst = dyn_lds + kPairs * (kScratchDoubles + kBoxDoubles);
    if (STAGED) {
        for (int f = 0; f < (int)F_SG; ++f)
            for (int i = threadIdx.x; i < wstride; i += 64 * kWavesPerGroup)
                cst[i * (int)F_SG + f] = i < wlen ? P.chain[(size_t)f * P.estride + wlo + i] : 0.0;
    }
    if (lane == 0) box->flag[wsub] = 0;
    __syncthreads();
    int seq = 0;
    for (;;) {
        // the team's first wave takes a cell from the queue and tells the others
        ++seq;
        unsigned c = 0;
        if (wsub == 0 && lane == 0) {
            c = atomicAdd(counter, 1u);
            mb_store(&box->data[0][seq & 1][0], (double)c);
        }
        wave_sync();
        if (lane == 0) __hip_atomic_store(&box->flag[wsub], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int o = 0; o < W; ++o) {
            if (o == wsub) continue;
            while (__hip_atomic_load(&box->flag[o], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) - seq < 0)
                IPC_SPIN_WAIT();
        }
        wave_sync();
        c = (unsigned)mb_load(&box->data[0][seq & 1][0]);
        c = (unsigned)__builtin_amdgcn_readfirstlane((int)c);
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
        CellResult r;
        se2_wave_solve<M, NL, STAGED, W>(P, lo, L, cand, iterations, sh, cst, wlo, wstride, r, box, seq, &seq);
        if (wsub == 0 && lane == 0) {
            out.max_chi2[c] = r.max_chi2;
            out.chi2_total[c] = r.chi2_total;
            out.meta[c] = make_int4(r.iterations, r.tries, r.flags, r.evals);
        }
        wave_sync();
    }
}

template <int W, int M, int NL>
static hipError_t launch_group(int n, hipStream_t st, const Se2View& P, const int2* cells, SolveParams prm, CellOut out,
                              unsigned* counter, int n_cu)
{
    const int E = P.V - 1;
    const int wstride = E + 32;
    constexpr int kPairs = kWavesPerGroup / W;
    const size_t scratch = kPairs * (((sizeof(WaveScratch<NL>) + 7) / 8) + ((sizeof(PairBox) + 7) / 8)) * sizeof(double);
    const size_t staged = scratch + sizeof(double) * (size_t)F_SG * wstride;
    const int groups = std::max(1, std::min(n_cu, (n + kPairs - 1) / kPairs));
    if (staged <= (size_t)kLdsBudget) {
        auto k = se2_group_kernel<W, M, NL, true>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)staged);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(groups), dim3(64 * kWavesPerGroup), staged, st, P, cells, n, counter, prm, out, 0, E, wstride);
    } else {
        auto k = se2_group_kernel<W, M, NL, false>;
        hipLaunchKernelGGL(k, dim3(groups), dim3(64 * kWavesPerGroup), scratch, st, P, cells, n, counter, prm, out, 0, E, wstride);
    }
    return hipGetLastError();
}

}  // namespace ipc
