// Levenberg retry at band sizes (round 6).  g2o's "dl_var" retries a solve whose factorisation met a
// non-positive pivot with lambda on the diagonal of H AT ANY SIZE (reference src/utils.cpp:104-105 ->
// OptimizationAlgorithmDogleg::solve); the capacitance formulation of the cluster solvers cannot carry lambda, so damped
// solves factor the literal normal equations (cluster_common.hpp::cluster_dogleg).  Rounds 3 - 5 stored them DENSE and gave
// up beyond 24 000 pose unknowns (Fail -> the candidate is rejected): C5's clusters (10 000 - 34 000 poses) were out of
// reach, C4's (15 000 unknowns) a dense 1.8 GB factorisation per retry.  Here the same system goes into the banded +
// bordered layout of cluster_band.hpp (poses in chain order, the later end of every loop that spans more than the band to
// the border: pose_band_plan) and is factored by the same bband_factor the large-cluster kernel uses.
#pragma once

namespace ipc {

struct LiteralBand {
    PoseBandPlan plan;
    BandLayout B{};
    double* d_A = nullptr; size_t capA = 0;         // system + factor
    double *d_dinv = nullptr, *d_x = nullptr; size_t capN = 0;
    int* d_ublk = nullptr; size_t capU = 0;
    PersistCtl* d_ctl = nullptr;
    double* d_zero = nullptr;
    long solves = 0;
    ~LiteralBand() { hipFree(d_A); hipFree(d_dinv); hipFree(d_x); hipFree(d_ublk); hipFree(d_ctl); hipFree(d_zero); }
};
inline void literal_band_free(LiteralBand* p) { delete p; }

inline hipError_t band_system_solve(const BandLayout& B, double* A, double* Lf, double* dinv, double* x, PersistCtl* ctl, int* info,
                                    const double* zero, int workgroups, hipStream_t st)
{
    static std::once_flag once;
    static hipError_t rc = hipSuccess;
    static int resident = 40;
    std::call_once(once, [] {
        rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&bband_test_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)(sizeof(double) * kLdsTotal));
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        if (rc == hipSuccess && hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&bband_test_kernel), kPT,
                                                         sizeof(double) * kLdsTotal) == hipSuccess && per_cu > 0)
            resident = per_cu * prop.multiProcessorCount;
    });
    IPC_CL_CHK(rc);
    IPC_CL_CHK(hipMemsetAsync(ctl, 0, sizeof(PersistCtl), st));
    BandArgs Q{B, A, Lf, dinv, nullptr, nullptr, 0, 0, nullptr, zero, -1, BandLayout{}, nullptr, nullptr, nullptr, nullptr};
    const int G = std::max(1, std::min(workgroups, resident));
    hipLaunchKernelGGL(bband_test_kernel, dim3(G), dim3(kPT), sizeof(double) * kLdsTotal, st, Q, x, ctl, info);
    return hipGetLastError();
}

// (H + lambda I) h = b of the solver's current linearisation through the banded store; the solution lands in x_out in
// the dense order [d (p - 1) + k].  used == false: the band does not pay for this cluster (the caller stores H dense).
template <class Solver>
hipError_t literal_band_damped(Solver& S, double lambda, bool& used, double* x_out, int* d_info)
{
    constexpr int d = Solver::kD;
    used = false;
    if (!S.lband) S.lband = new LiteralBand();
    LiteralBand& W = *S.lband;
    const LoopTables& T = S.tables();
    const int L = T.L, nl = T.nl;
    hipStream_t st = S.stream();
    if (S.lband_stale) {
        const int* lf = T.host.data();
        W.plan = pose_band_plan(d, L, nl, lf, lf + nl);
        S.lband_stale = false;
        if (W.plan.use) {
            if ((size_t)L + 1 > W.capU) {
                hipFree(W.d_ublk); W.d_ublk = nullptr;
                W.capU = (size_t)L + 1 + L / 2;
                IPC_CL_CHK(hipMalloc(&W.d_ublk, sizeof(int) * W.capU));
            }
            IPC_CL_CHK(hipMemcpyAsync(W.d_ublk, W.plan.ublk.data(), sizeof(int) * ((size_t)L + 1), hipMemcpyHostToDevice, st));
            IPC_CL_CHK(hipStreamSynchronize(st));           // (the plan's vector may be rebuilt by the next solve)
            BandLayout& B = W.B;
            B.nb = d * W.plan.nbp; B.m = d * W.plan.nborder + 1; B.W = std::max(64, d * (W.plan.S + 1));
            B.ldb = B.W + B.m; B.n = B.nb + B.m - 1;
        }
    }
    if (!W.plan.use) return hipSuccess;
    const BandLayout B = W.B;
    if (B.doubles() >= ((size_t)1 << 31)) return hipSuccess;       // (32-bit addressing of the factorisation)
    if (!W.d_ctl) {
        IPC_CL_CHK(hipMalloc(&W.d_ctl, sizeof(PersistCtl)));
        IPC_CL_CHK(hipMalloc(&W.d_zero, sizeof(double) * 8));
        IPC_CL_CHK(hipMemsetAsync(W.d_zero, 0, sizeof(double) * 8, st));
    }
    if (2 * B.doubles() > W.capA) {
        hipFree(W.d_A); W.d_A = nullptr;
        W.capA = 2 * B.doubles() + B.doubles() / 4;
        IPC_CL_CHK(hipMalloc(&W.d_A, sizeof(double) * W.capA));
    }
    if ((size_t)B.n + 64 > W.capN) {
        hipFree(W.d_dinv); hipFree(W.d_x); W.d_dinv = W.d_x = nullptr;
        W.capN = (size_t)B.n + 64 + B.n / 4;
        IPC_CL_CHK(hipMalloc(&W.d_dinv, sizeof(double) * W.capN));
        IPC_CL_CHK(hipMalloc(&W.d_x, sizeof(double) * W.capN));
    }
    IPC_CL_CHK(hipMemsetAsync(W.d_A, 0, sizeof(double) * 2 * B.doubles(), st));
    S.launch_literal_H(BandStore{W.d_A, B, W.d_ublk, d}, lambda);
    // workgroups: one per two 64 x 64 tiles of a block column's trailing window (cluster_persist.hpp's rule for the band kernel)
    const int R = B.W + B.m, nti = (R + 63) / 64, tiles = nti * (nti + 1) / 2;
    const int G = 1 + std::min(39, (tiles + kPSG - 1) / kPSG);
    IPC_CL_CHK(band_system_solve(B, W.d_A, W.d_A + B.doubles(), W.d_dinv, W.d_x, W.d_ctl, d_info, W.d_zero, G, st));
    hipLaunchKernelGGL(gk_unpermute_blocks, dim3((L + 1 + kGB - 1) / kGB), dim3(kGB), 0, st, (const double*)W.d_x, (const int*)W.d_ublk, d, L, x_out);
    IPC_CL_CHK(hipGetLastError());
    ++W.solves;
    used = true;
    return hipSuccess;
}

inline long ClusterSolver2::literal_band_solves() const { return lband ? lband->solves : 0; }
inline long ClusterSolver3::literal_band_solves() const { return lband ? lband->solves : 0; }

}  // namespace ipc
