// Block kernels of the SE(3) cell solver (se3_cell.hpp): one workgroup per cell.
#include "cell_kernels.hpp"

using namespace ipc;

// ---- SE3 cell kernel: same contract; variants kept few (the code is large) ----
template <int W, int M, int NL>
__global__ __launch_bounds__(64 * W) void se3_cells_kernel(Se3View P, const int2* cells, int ncells,
                                                           SolveParams prm, CellOut out)
{
    __shared__ Se3Shared<W, M, NL> sh;
    const int cell = blockIdx.x;
    if (cell >= ncells) return;
    const int2 cc = cells[cell];
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
    se3_solve_cell<W, M, NL>(P, lo, L, cand, iterations, sh, r);
    if (threadIdx.x == 0) {
        out.max_chi2[cell] = r.max_chi2;
        out.chi2_total[cell] = r.chi2_total;
        out.meta[cell] = make_int4(r.iterations, r.tries, r.flags, r.evals);
    }
}

template <int NL>
static hipError_t launch_se3(int variant, int n, hipStream_t st, const Se3View& P, const int2* cells,
                             SolveParams prm, CellOut out)
{
#define IPC_CASE3(idx, WW, MM)                                                                    \
    case idx:                                                                                     \
        hipLaunchKernelGGL((se3_cells_kernel<WW, MM, NL>), dim3(n), dim3(64 * WW), 0, st, P, cells, n, prm, out); \
        break;
    switch (variant) {
        IPC_CASE3(0, 1, 1)
        IPC_CASE3(1, 2, 1)
        IPC_CASE3(2, 4, 1)
        IPC_CASE3(3, 8, 1)
        IPC_CASE3(4, 16, 1)
        IPC_CASE3(5, 16, 2)
        IPC_CASE3(6, 16, 4)
        IPC_CASE3(7, 4, 2)
        IPC_CASE3(8, 4, 4)
        IPC_CASE3(9, 8, 2)
        IPC_CASE3(10, 8, 4)
        IPC_CASE3(11, 4, 8)
        IPC_CASE3(12, 8, 5)
        IPC_CASE3(13, 1, 2)
        IPC_CASE3(14, 2, 2)
        IPC_CASE3(15, 1, 4)
        IPC_CASE3(16, 2, 4)
        IPC_CASE3(17, 1, 3)
        IPC_CASE3(18, 2, 3)
        default: return hipErrorInvalidValue;
    }
#undef IPC_CASE3
    return hipGetLastError();
}


namespace ipc {
hipError_t launch_se3_block(int nl, int variant, int n, hipStream_t st, const Se3View& P, const int2* cells,
                            SolveParams prm, CellOut out)
{
    return nl == 1 ? launch_se3<1>(variant, n, st, P, cells, prm, out) : launch_se3<2>(variant, n, st, P, cells, prm, out);
}
}  // namespace ipc
