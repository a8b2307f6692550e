// Block kernels of the SE(2) cell solver (se2_cell.hpp): one workgroup per cell.
#include "cell_kernels.hpp"

using namespace ipc;

// ------------------------------------------------------------------------------------------
// the hot kernel
// ------------------------------------------------------------------------------------------
// M == 1 variants are held to 128 VGPRs (4 waves per SIMD => 16 waves per CU: two 8-wave cells
// or four 4-wave cells in flight per CU, so one cell's barrier waits hide behind another's work)
#ifndef IPC_MINW
#define IPC_MINW 1
#endif
template <int W, int M, int NL>
__global__ __launch_bounds__(64 * W, (M == 1 ? IPC_MINW : (W == 2 ? 2 : 1))) void se2_cells_kernel(Se2View P, const int2* cells, int ncells,
                                                           SolveParams prm, CellOut out)
{
    __shared__ Se2Shared<W, M, NL> sh;
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
    CellResult r;
    se2_solve_cell<W, M, NL>(P, lo, L, cand, iterations, sh, r);
    if (threadIdx.x == 0) {
        out.max_chi2[cell] = r.max_chi2;
        out.chi2_total[cell] = r.chi2_total;
        out.meta[cell] = make_int4(r.iterations, r.tries, r.flags, r.evals);
    }
}

template <int NL>
static hipError_t launch_se2(int variant, int n, hipStream_t st, const Se2View& P, const int2* cells,
                             SolveParams prm, CellOut out)
{
#define IPC_CASE(idx, WW, MM)                                                                     \
    case idx:                                                                                     \
        hipLaunchKernelGGL((se2_cells_kernel<WW, MM, NL>), dim3(n), dim3(64 * WW), 0, st, P, cells, n, prm, out); \
        break;
    switch (variant) {
        IPC_CASE(0, 1, 1)
        IPC_CASE(1, 2, 1)
        IPC_CASE(2, 3, 1)
        IPC_CASE(3, 4, 1)
        IPC_CASE(4, 5, 1)
        IPC_CASE(5, 6, 1)
        IPC_CASE(6, 7, 1)
        IPC_CASE(7, 8, 1)
        IPC_CASE(8, 10, 1)
        IPC_CASE(9, 12, 1)
        IPC_CASE(10, 14, 1)
        IPC_CASE(11, 16, 1)
        IPC_CASE(12, 4, 2)
        IPC_CASE(13, 5, 2)
        IPC_CASE(14, 6, 2)
        IPC_CASE(15, 7, 2)
        IPC_CASE(16, 8, 2)
        IPC_CASE(17, 10, 2)
        IPC_CASE(18, 12, 2)
        IPC_CASE(19, 16, 2)
        IPC_CASE(20, 5, 3)
        IPC_CASE(21, 6, 3)
        IPC_CASE(22, 7, 3)
        IPC_CASE(23, 8, 3)
        IPC_CASE(24, 4, 4)
        IPC_CASE(25, 6, 4)
        IPC_CASE(26, 8, 4)
        IPC_CASE(27, 16, 4)
        IPC_CASE(28, 16, 8)
        IPC_CASE(29, 16, 16)
        IPC_CASE(30, 2, 3)
        IPC_CASE(31, 2, 4)
        IPC_CASE(32, 2, 5)
        IPC_CASE(33, 2, 6)
        IPC_CASE(34, 3, 4)
        IPC_CASE(35, 3, 5)
        IPC_CASE(36, 3, 6)
        IPC_CASE(37, 4, 3)
        IPC_CASE(38, 4, 5)
        default: return hipErrorInvalidValue;
    }
#undef IPC_CASE
    return hipGetLastError();
}


namespace ipc {
hipError_t launch_se2_block(int nl, int variant, int n, hipStream_t st, const Se2View& P, const int2* cells,
                            SolveParams prm, CellOut out)
{
    return nl == 1 ? launch_se2<1>(variant, n, st, P, cells, prm, out) : launch_se2<2>(variant, n, st, P, cells, prm, out);
}
}  // namespace ipc
