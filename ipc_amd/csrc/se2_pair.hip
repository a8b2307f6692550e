// Pair kernels of the SE(2) cell solver: two waves per cell (se2_group_kernel.hpp with W = 2).
#include "se2_group_kernel.hpp"

namespace ipc {
hipError_t launch_se2_pair(int nl, int M, int n, hipStream_t st, const Se2View& P, const int2* cells, SolveParams prm,
                           CellOut out, unsigned* counter, int n_cu)
{
#define IPC_PCASE(MM)                                                                              \
    case MM:                                                                                       \
        return nl == 1 ? launch_group<2, MM, 1>(n, st, P, cells, prm, out, counter, n_cu)          \
                       : launch_group<2, MM, 2>(n, st, P, cells, prm, out, counter, n_cu);
    switch (M) {
#ifdef IPC_PAIR_ONLY_M                                 // (tools/maxilp_repro.py: one instantiation, seconds to compile)
        IPC_PCASE(IPC_PAIR_ONLY_M)
#else
        IPC_PCASE(5)
        IPC_PCASE(7)
        IPC_PCASE(9)
        IPC_PCASE(11)
#endif
        default: return hipErrorInvalidValue;
    }
#undef IPC_PCASE
}
}  // namespace ipc
