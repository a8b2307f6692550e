// Launch interface of the cell-solver kernels.  The kernels are heavily templated, so they live
// in their own translation units (se2_block.hip, se2_wave.hip, se2_pair.hip, se2_quad.hip, se3_block.hip) that are compiled
// in parallel and linked into libipc_amd.so; engine.hip only sees these prototypes.
#pragma once
#include <hip/hip_runtime.h>

#include "se2_cell.hpp"
#include "se3_cell.hpp"

namespace ipc {

struct CellOut {
    double* max_chi2;
    double* chi2_total;
    int4* meta;               // iterations, tries, flags, error evaluations
};

// ---- block kernels: one workgroup of W waves per cell, M poses per lane; capacity 64*W*M ----
struct Variant { int W, M; };
static const Variant kVariants[] = {
    // M = 1
    {1, 1}, {2, 1}, {3, 1}, {4, 1}, {5, 1}, {6, 1}, {7, 1}, {8, 1}, {10, 1}, {12, 1}, {14, 1}, {16, 1},
    // M = 2
    {4, 2}, {5, 2}, {6, 2}, {7, 2}, {8, 2}, {10, 2}, {12, 2}, {16, 2},
    // M = 3
    {5, 3}, {6, 3}, {7, 3}, {8, 3},
    // M = 4 and longer chains
    {4, 4}, {6, 4}, {8, 4}, {16, 4}, {16, 8}, {16, 16},
    // two waves per cell, two cells per CU
    {2, 3}, {2, 4}, {2, 5}, {2, 6},
    // three / four fat waves per cell
    {3, 4}, {3, 5}, {3, 6}, {4, 3}, {4, 5},
};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);
static const Variant kVariants3[] = { {1, 1}, {2, 1}, {4, 1}, {8, 1}, {16, 1}, {16, 2}, {16, 4},
                                      {4, 2}, {4, 4}, {8, 2}, {8, 4}, {4, 8}, {8, 5},
                                      {1, 2}, {2, 2}, {1, 4}, {2, 4}, {1, 3}, {2, 3} };
constexpr int kNumVariants3 = sizeof(kVariants3) / sizeof(kVariants3[0]);

hipError_t launch_se2_block(int nl, int variant, int n, hipStream_t st, const Se2View& P, const int2* cells,
                            SolveParams prm, CellOut out);
hipError_t launch_se3_block(int nl, int variant, int n, hipStream_t st, const Se3View& P, const int2* cells,
                            SolveParams prm, CellOut out);

// ---- LDS-pose kernels (SE3, se3_lds_cell.hpp): teams of W = 1 or 4 waves per cell, M poses per lane;
// capacity 64*W*M.  Policy tokens "wM" (W = 1) and "gM" (W = 4); plan variant id = base + table index.
struct LdsVariant { int W, M; };
static const LdsVariant kLdsVariants3[] = { {1, 1}, {1, 2}, {1, 3}, {1, 4}, {1, 6}, {1, 8},
                                            {4, 2}, {4, 3}, {4, 4}, {4, 5}, {4, 6}, {4, 7}, {4, 8}, {4, 9}, {4, 10} };
constexpr int kNumLdsVariants3 = sizeof(kLdsVariants3) / sizeof(kLdsVariants3[0]);
constexpr int kLdsVariantBase3 = 400;
#define IPC_SE3_LDS_DECL(WW, MM)                                                                                \
    hipError_t launch_se3_lds_##WW##_##MM(int nl, int n, hipStream_t st, const Se3View& P, const int2* cells,   \
                                          SolveParams prm, CellOut out, unsigned* counter, int n_cu);
IPC_SE3_LDS_DECL(1, 1) IPC_SE3_LDS_DECL(1, 2) IPC_SE3_LDS_DECL(1, 3) IPC_SE3_LDS_DECL(1, 4) IPC_SE3_LDS_DECL(1, 6)
IPC_SE3_LDS_DECL(1, 8) IPC_SE3_LDS_DECL(4, 2) IPC_SE3_LDS_DECL(4, 3) IPC_SE3_LDS_DECL(4, 4) IPC_SE3_LDS_DECL(4, 5)
IPC_SE3_LDS_DECL(4, 6) IPC_SE3_LDS_DECL(4, 7) IPC_SE3_LDS_DECL(4, 8) IPC_SE3_LDS_DECL(4, 9) IPC_SE3_LDS_DECL(4, 10)
#undef IPC_SE3_LDS_DECL

// ---- wave kernels (SE2): one wave per cell, M consecutive poses per lane; capacity 64*M ----
static const int kWaveM[] = {1, 3, 5, 7, 9, 11, 13};
constexpr int kNumWaveM = sizeof(kWaveM) / sizeof(kWaveM[0]);
constexpr int kWaveVariantBase = 100;         // plan variant id of the wave kernel with M poses per lane = base + M

// counter: one zeroed unsigned in HBM (work queue of the launch); n_cu: workgroups to launch.
// The chain window [0, V-1) is staged in LDS when it fits, else the constants are read from L2.
hipError_t launch_se2_wave(int nl, int M, int n, hipStream_t st, const Se2View& P, const int2* cells,
                           SolveParams prm, CellOut out, unsigned* counter, int n_cu);

// ---- pair kernels (SE2): two waves per cell, M consecutive poses per lane; capacity 128*M ----
static const int kPairM[] = {5, 7, 9, 11};
constexpr int kNumPairM = sizeof(kPairM) / sizeof(kPairM[0]);
constexpr int kPairVariantBase = 200;
hipError_t launch_se2_pair(int nl, int M, int n, hipStream_t st, const Se2View& P, const int2* cells,
                           SolveParams prm, CellOut out, unsigned* counter, int n_cu);

// ---- quad kernels (SE2): the four waves of a workgroup on one cell; capacity 256*M ----
static const int kQuadM[] = {7, 9, 11, 13};
constexpr int kNumQuadM = sizeof(kQuadM) / sizeof(kQuadM[0]);
constexpr int kQuadVariantBase = 300;
hipError_t launch_se2_quad(int nl, int M, int n, hipStream_t st, const Se2View& P, const int2* cells,
                           SolveParams prm, CellOut out, unsigned* counter, int n_cu);

}  // namespace ipc
