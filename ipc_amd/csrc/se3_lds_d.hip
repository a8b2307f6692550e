// LDS-pose SE(3) cell kernels (se3_lds_kernel.hpp): one translation unit per group of (W, M) variants
#include "se3_lds_kernel.hpp"

IPC_SE3_LDS_UNIT(1, 8)
IPC_SE3_LDS_UNIT(4, 3)
IPC_SE3_LDS_UNIT(4, 7)
