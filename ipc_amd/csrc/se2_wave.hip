// Wave kernels of the SE(2) cell solver (se2_wave_cell.hpp): persistent workgroups of four
// independent waves, each solving one cell at a time from a shared work queue.
#include "cell_kernels.hpp"
#include "se2_wave_cell.hpp"

using namespace ipc;

// Waves per workgroup: four (one per SIMD, up to 512 registers each); the shortest chains (M <= IPC_SE2_DENSE_MAXM
// poses per lane) fit 256 registers, so eight of their waves share a CU -- two per SIMD, each hiding the other's
// latencies -- and the staged chain constants.
#ifndef IPC_SE2_DENSE_MAXM
#define IPC_SE2_DENSE_MAXM 1
#endif
template <int M>
constexpr int se2_waves() { return M <= IPC_SE2_DENSE_MAXM ? 8 : 4; }

template <int M, int NL, bool STAGED>
__global__ __launch_bounds__((64 * se2_waves<M>()), 1) void se2_wave_kernel(Se2View P, const int2* cells, int ncells,
                                                                          unsigned* counter, SolveParams prm,
                                                                          CellOut out, int wlo, int wlen, int wstride)
{
    extern __shared__ double dyn_lds[];
    constexpr int kWavesPerGroup = se2_waves<M>();
    constexpr int kScratchDoubles = (sizeof(WaveScratch<NL>) + 7) / 8;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    WaveScratch<NL>& sh = *reinterpret_cast<WaveScratch<NL>*>(dyn_lds + wave * kScratchDoubles);
    double* cst = dyn_lds + kWavesPerGroup * kScratchDoubles;
    if (STAGED) {
        // residual-pass constants of the whole window, once per workgroup (rows padded with zeros)
        for (int f = 0; f < (int)F_SG; ++f)
            for (int i = threadIdx.x; i < wstride; i += 64 * kWavesPerGroup)
                cst[i * (int)F_SG + f] = i < wlen ? P.chain[(size_t)f * P.estride + wlo + i] : 0.0;
    }
    __syncthreads();
    for (;;) {
        unsigned c = 0;
        if (lane == 0) c = atomicAdd(counter, 1u);
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
        se2_wave_solve<M, NL, STAGED>(P, lo, L, cand, iterations, sh, cst, wlo, wstride, r);
        if (lane == 0) {
            out.max_chi2[c] = r.max_chi2;
            out.chi2_total[c] = r.chi2_total;
            out.meta[c] = make_int4(r.iterations, r.tries, r.flags, r.evals);
        }
        wave_sync();                                  // the scratch is reused by the next cell
    }
}

#ifndef IPC_LDS_BUDGET
#define IPC_LDS_BUDGET (160 * 1024)
#endif
constexpr int kLdsBudget = IPC_LDS_BUDGET;

template <int M, int NL>
static hipError_t launch_one(int n, hipStream_t st, const Se2View& P, const int2* cells, SolveParams prm, CellOut out,
                             unsigned* counter, int n_cu)
{
    constexpr int kWavesPerGroup = se2_waves<M>();
    const int E = P.V - 1;
    const int wstride = E + 32;                       // padding: a partially filled lane reads up to M-1 records past the end
    const size_t scratch = kWavesPerGroup * ((sizeof(WaveScratch<NL>) + 7) / 8) * sizeof(double);
    const size_t staged = scratch + sizeof(double) * (size_t)F_SG * wstride;
    const int groups = std::max(1, std::min(n_cu, (n + kWavesPerGroup - 1) / kWavesPerGroup));
    if (staged <= (size_t)kLdsBudget) {
        auto k = se2_wave_kernel<M, NL, true>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)staged);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(groups), dim3(64 * kWavesPerGroup), staged, st, P, cells, n, counter, prm, out, 0, E, wstride);
    } else {
        auto k = se2_wave_kernel<M, NL, false>;
        hipLaunchKernelGGL(k, dim3(groups), dim3(64 * kWavesPerGroup), scratch, st, P, cells, n, counter, prm, out, 0, E, wstride);
    }
    return hipGetLastError();
}

namespace ipc {
hipError_t launch_se2_wave(int nl, int M, int n, hipStream_t st, const Se2View& P, const int2* cells, SolveParams prm,
                           CellOut out, unsigned* counter, int n_cu)
{
#define IPC_WCASE(MM)                                                                              \
    case MM:                                                                                       \
        return nl == 1 ? launch_one<MM, 1>(n, st, P, cells, prm, out, counter, n_cu)               \
                       : launch_one<MM, 2>(n, st, P, cells, prm, out, counter, n_cu);
    switch (M) {
#ifdef IPC_WAVE_ONLY_M
        IPC_WCASE(IPC_WAVE_ONLY_M)
#else
        IPC_WCASE(1)
        IPC_WCASE(3)
        IPC_WCASE(5)
        IPC_WCASE(7)
        IPC_WCASE(9)
        IPC_WCASE(11)
        IPC_WCASE(13)
#endif
        default: return hipErrorInvalidValue;
    }
#undef IPC_WCASE
}

}  // namespace ipc
