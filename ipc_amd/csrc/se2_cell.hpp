// SE(2) consistency cell solver -- one workgroup per cell, hand-written for gfx950.
//
// A cell is the reference's isAgreeingWithCurrentState (reference src/consensus_utils.cpp:7-22)
// applied to: the odometry chain lo..hi (gauge = pose lo fixed, reference
// src/consensus_utils.cpp:29-43), the open-loop initial guess (src/consensus_utils.cpp:99-116),
// odometry information pre-scaled by s (src/consensus_utils.cpp:124-130) and ONE (diagonal cell)
// or TWO (pair cell) loop-closure edges; the optimiser is g2o's dog-leg
// ("dl_var", src/utils.cpp:105) run for iter_base*(5 if #edges>100) iterations, then
// max_e chi2_e is compared with the threshold by the caller.
//
// MI355X-first formulation (this is not how g2o does the linear algebra):
//   * poses 1..L of the chain are spread over the T threads of the workgroup, M consecutive
//     poses per thread, all state in VGPRs; LDS is used only for neighbour hand-off, loop
//     end-point broadcast and reductions.
//   * the Gauss-Newton step H h = b is NOT obtained by factoring the (block tridiagonal +
//     arrow) H.  With u = Jc h (Jc = square block-bidiagonal Jacobian of the odometry chain)
//     the normal equations become  (Om_c + G^T Om_l G) u = Jc^-T b  with G = Jl Jc^-1, and for
//     SE(2) Jc^-1 has the closed form "propagate a perturbation down the chain":
//         h_theta(i) = sum_{j<=i} rho_theta(j),
//         h_t(i)     = sum_{j<=i} [ rho_t(j) + J (t_j - t_{j-1}) h_theta(j-1) ],   J=[[0,-1],[1,0]]
//     so G is element-wise, the capacitance matrix S = Cov_l + sum_j G_j Cov_j G_j^T is a
//     3x3 / 6x6 reduction, and h follows from two dependent prefix sums.  Everything is
//     element-wise FP64 work + workgroup reductions/scans: no sequential factorisation.
//   * dog-leg bookkeeping follows g2o exactly (delta, rho, <=100 trials, Terminate); two
//     shortcuts are taken that are bit-exact w.r.t. the un-shortcut loop:
//       - a rejected Gauss-Newton trial repeats identically while ||h_gn|| < delta, so those
//         repeats only halve delta and count trials;
//       - a trial whose update leaves every pose bit-identical has newChi == currentChi, hence
//         rho = 0 -> rejected; for steepest-descent steps all later (halved) steps are no-ops
//         too, so the trial counter is fast-forwarded to Terminate.
#pragma once
#include "block_prims.hpp"

namespace ipc {

constexpr double kPi = 3.14159265358979323846;

// chain / candidate record fields (structure of arrays, field-major)
enum Se2Field { F_TZX = 0, F_TZY, F_CZ, F_SZ, F_THZ, F_OM = 5, F_SG = 11, F_NFIELDS = 17 };

struct Se2View {
    const double* chain;      // [F_NFIELDS][estride], index = edge k (joins pose k -> k+1)
    int estride;
    const double* pose0;      // [3][V] open-loop poses x, y, theta
    int V;
    const double* cand;       // [F_NFIELDS][cstride] per loop candidate
    int cstride;
    const int* cand_from;     // candidate "from" / "to" vertex ids
    const int* cand_to;
};

struct SolveParams {
    int fast_iter, slow_iter;
};

__device__ __forceinline__ double normalize_theta(double theta)
{
    if (theta >= -kPi && theta < kPi) return theta;
    double multiplier = floor(theta / (2 * kPi));
    theta = theta - multiplier * 2 * kPi;
    if (theta >= kPi) theta -= 2 * kPi;
    if (theta < -kPi) theta += 2 * kPi;
    return theta;
}

struct Sym3 {                 // symmetric 3x3: 00 01 02 11 12 22
    double a00, a01, a02, a11, a12, a22;
    __device__ __forceinline__ void mul(double x, double y, double z, double& ox, double& oy, double& oz) const
    {
        ox = a00 * x + a01 * y + a02 * z;
        oy = a01 * x + a11 * y + a12 * z;
        oz = a02 * x + a12 * y + a22 * z;
    }
    __device__ __forceinline__ double quad(double x, double y, double z) const
    {
        double ox, oy, oz;
        mul(x, y, z, ox, oy, oz);
        return x * ox + y * oy + z * oz;
    }
};

__device__ __forceinline__ Sym3 load_sym3(const double* base, int stride, int field0, int idx)
{
    Sym3 s;
    s.a00 = base[(size_t)(field0 + 0) * stride + idx];
    s.a01 = base[(size_t)(field0 + 1) * stride + idx];
    s.a02 = base[(size_t)(field0 + 2) * stride + idx];
    s.a11 = base[(size_t)(field0 + 3) * stride + idx];
    s.a12 = base[(size_t)(field0 + 4) * stride + idx];
    s.a22 = base[(size_t)(field0 + 5) * stride + idx];
    return s;
}

struct Pose2 { double x, y, th, c, s; };

// loop edge as seen by every thread (uniform)
struct Loop2 {
    int f, t;                 // local pose indices 0..L
    int lo, hi;               // min/max of (f,t)
    double sigma;             // +1 if t > f, -1 otherwise
    double tzx, tzy, cz, sz, thz;
    Sym3 om, sg;
    // per-state values
    Pose2 pf, pt;
    double ex, ey, eth;       // error
    double gx, gy, gth;       // world-frame force  (R_f Rz q_t, q_theta),  q = Om e
    double chi;
};

// error of an SE2 edge a -> b with measurement (tz, cz, sz, thz); also returns r = R_a^T (t_b - t_a)
__device__ __forceinline__ void se2_error(const Pose2& a, const Pose2& b, double tzx, double tzy,
                                          double cz, double sz, double thz, double& ex, double& ey,
                                          double& eth, double& rx, double& ry)
{
    double dx = b.x - a.x, dy = b.y - a.y;
    rx = a.c * dx + a.s * dy;
    ry = -a.s * dx + a.c * dy;
    double lx = rx - tzx, ly = ry - tzy;
    ex = cz * lx + sz * ly;
    ey = -sz * lx + cz * ly;
    eth = normalize_theta(normalize_theta(b.th - a.th) - thz);
}

// 6x6 (or 3x3) SPD solve by Cholesky; n <= 6.  A is full row-major n x n, overwritten.
template <int N>
__device__ __forceinline__ bool chol_solve(double (&A)[N][N], double (&b)[N])
{
    bool ok = true;
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double sum = A[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) sum -= A[i][k] * A[j][k];
            if (j < i) A[i][j] = sum / A[j][j];
            else { if (!(sum > 0)) ok = false; A[i][i] = sqrt(sum); }
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double sum = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) sum -= A[i][k] * b[k];
        b[i] = sum / A[i][i];
    }
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        double sum = b[i];
#pragma unroll
        for (int k = i + 1; k < N; ++k) sum -= A[k][i] * b[k];
        b[i] = sum / A[i][i];
    }
    return ok;
}

struct CellResult {
    double max_chi2;
    double chi2_total;
    int iterations;
    int tries;
    int flags;                // bit0: dog-leg Terminate, bit1: capacitance not PD (Fail)
    int evals;                // error evaluations (sincos + residual passes) executed
};

template <int T, int M>
struct Se2Shared {
    static constexpr int W = T / 64;
    double prev[5][T];        // last-slot pose of every thread (x, y, th, c, s)
    double nxt[3][T];         // first-slot hand-back vector of every thread
    double bc[2][2][5];       // loop end-point poses (loop, from/to, fields)
    double bv[2][2][3];       // loop end-point vectors (b or h)
    double red[W * 32];
    int flag[W];
};

template <int T, int M, int NL>
__device__ void se2_solve_cell(const Se2View& P, int lo_abs, int L, const int (&cand)[2], int iterations,
                               Se2Shared<T, M>& sh, CellResult& res)
{
    constexpr int W = T / 64;
    const int tid = threadIdx.x;
    const int j0 = tid * M + 1;                    // first pose index owned by this thread

    // ---------------- loop edges (uniform) ----------------
    Loop2 lp[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const int c = cand[l];
        lp[l].f = P.cand_from[c] - lo_abs;
        lp[l].t = P.cand_to[c] - lo_abs;
        lp[l].lo = min(lp[l].f, lp[l].t);
        lp[l].hi = max(lp[l].f, lp[l].t);
        lp[l].sigma = lp[l].t > lp[l].f ? 1.0 : -1.0;
        lp[l].tzx = P.cand[(size_t)F_TZX * P.cstride + c];
        lp[l].tzy = P.cand[(size_t)F_TZY * P.cstride + c];
        lp[l].cz = P.cand[(size_t)F_CZ * P.cstride + c];
        lp[l].sz = P.cand[(size_t)F_SZ * P.cstride + c];
        lp[l].thz = P.cand[(size_t)F_THZ * P.cstride + c];
        lp[l].om = load_sym3(P.cand, P.cstride, F_OM, c);
        lp[l].sg = load_sym3(P.cand, P.cstride, F_SG, c);
    }

    // ---------------- per-thread state ----------------
    Pose2 X[M], Xb[M];
    double ex[M], ey[M], eth[M];                   // odometry errors of edge j (j-1 -> j)
    double bx[M], by[M], bth[M];                   // b = -J^T Om e
    double hx[M], hy[M], hth[M];                   // Gauss-Newton step
    double dlx[M], dly[M], dlth[M];                // dog-leg step actually tried
    bool valid[M];
    Pose2 gauge;
    {
        gauge.x = P.pose0[lo_abs];
        gauge.y = P.pose0[(size_t)P.V + lo_abs];
        gauge.th = P.pose0[(size_t)2 * P.V + lo_abs];
        sincos(gauge.th, &gauge.s, &gauge.c);
    }
#pragma unroll
    for (int s = 0; s < M; ++s) {
        const int j = j0 + s;
        valid[s] = j <= L;
        const int ja = valid[s] ? lo_abs + j : lo_abs;
        X[s].x = P.pose0[ja];
        X[s].y = P.pose0[(size_t)P.V + ja];
        X[s].th = P.pose0[(size_t)2 * P.V + ja];
        sincos(X[s].th, &X[s].s, &X[s].c);
        hx[s] = hy[s] = hth[s] = 0.0;
        bx[s] = by[s] = bth[s] = 0.0;
        ex[s] = ey[s] = eth[s] = 0.0;
    }

    // edge constants are re-read from L1/L2 where needed (the chain is shared by every cell)
    auto EK = [&](int s) { return lo_abs + j0 + s - 1; };   // absolute edge index of slot s
    auto ld = [&](int field, int s) { return P.chain[(size_t)field * P.estride + EK(s)]; };

    Pose2 prevPose;                                 // pose j0-1 (neighbour's last slot or gauge)

    // publish poses: neighbour hand-off + loop end points; fills prevPose and lp[].pf/pt
    auto exchange_poses = [&]() {
        __syncthreads();
        sh.prev[0][tid] = X[M - 1].x; sh.prev[1][tid] = X[M - 1].y; sh.prev[2][tid] = X[M - 1].th;
        sh.prev[3][tid] = X[M - 1].c; sh.prev[4][tid] = X[M - 1].s;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int pj = e == 0 ? lp[l].f : lp[l].t;
                if (pj == 0) {
                    if (tid == 0) {
                        sh.bc[l][e][0] = gauge.x; sh.bc[l][e][1] = gauge.y; sh.bc[l][e][2] = gauge.th;
                        sh.bc[l][e][3] = gauge.c; sh.bc[l][e][4] = gauge.s;
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < M; ++s)
                        if (j0 + s == pj) {
                            sh.bc[l][e][0] = X[s].x; sh.bc[l][e][1] = X[s].y; sh.bc[l][e][2] = X[s].th;
                            sh.bc[l][e][3] = X[s].c; sh.bc[l][e][4] = X[s].s;
                        }
                }
            }
        }
        __syncthreads();
        if (tid == 0) prevPose = gauge;
        else {
            prevPose.x = sh.prev[0][tid - 1]; prevPose.y = sh.prev[1][tid - 1]; prevPose.th = sh.prev[2][tid - 1];
            prevPose.c = sh.prev[3][tid - 1]; prevPose.s = sh.prev[4][tid - 1];
        }
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            lp[l].pf = Pose2{sh.bc[l][0][0], sh.bc[l][0][1], sh.bc[l][0][2], sh.bc[l][0][3], sh.bc[l][0][4]};
            lp[l].pt = Pose2{sh.bc[l][1][0], sh.bc[l][1][1], sh.bc[l][1][2], sh.bc[l][1][3], sh.bc[l][1][4]};
        }
    };

    // computeActiveErrors + activeRobustChi2 at the current poses
    int evals = 0;
    auto eval_errors = [&]() -> double {
        ++evals;
        exchange_poses();
        double part[1] = {0.0};
#pragma unroll
        for (int s = 0; s < M; ++s) {
            if (!valid[s]) continue;
            const Pose2& a = s == 0 ? prevPose : X[s > 0 ? s - 1 : 0];
            double rx, ry;
            se2_error(a, X[s], ld(F_TZX, s), ld(F_TZY, s), ld(F_CZ, s), ld(F_SZ, s), ld(F_THZ, s),
                      ex[s], ey[s], eth[s], rx, ry);
            Sym3 om = load_sym3(P.chain, P.estride, F_OM, EK(s));
            part[0] += om.quad(ex[s], ey[s], eth[s]);
        }
        block_sum<W, 1>(part, sh.red);
        double chi = part[0];
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            double rx, ry;
            se2_error(lp[l].pf, lp[l].pt, lp[l].tzx, lp[l].tzy, lp[l].cz, lp[l].sz, lp[l].thz,
                      lp[l].ex, lp[l].ey, lp[l].eth, rx, ry);
            double qx, qy, qth;
            lp[l].om.mul(lp[l].ex, lp[l].ey, lp[l].eth, qx, qy, qth);
            lp[l].chi = lp[l].ex * qx + lp[l].ey * qy + lp[l].eth * qth;
            const double cP = lp[l].pf.c * lp[l].cz - lp[l].pf.s * lp[l].sz;
            const double sP = lp[l].pf.s * lp[l].cz + lp[l].pf.c * lp[l].sz;
            lp[l].gx = cP * qx - sP * qy;
            lp[l].gy = sP * qx + cP * qy;
            lp[l].gth = qth;
            chi += lp[l].chi;
        }
        return chi;
    };

    // hand a per-pose 3-vector of the previous pose (j0-1) and of the loop end points around
    auto exchange_vec = [&](const double (&vx)[M], const double (&vy)[M], const double (&vth)[M],
                            double& px, double& py, double& pth, double (&lv)[NL][2][3]) {
        __syncthreads();
        sh.prev[0][tid] = vx[M - 1]; sh.prev[1][tid] = vy[M - 1]; sh.prev[2][tid] = vth[M - 1];
#pragma unroll
        for (int l = 0; l < NL; ++l) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int pj = e == 0 ? lp[l].f : lp[l].t;
                if (pj == 0) {
                    if (tid == 0) { sh.bv[l][e][0] = 0.0; sh.bv[l][e][1] = 0.0; sh.bv[l][e][2] = 0.0; }
                } else {
#pragma unroll
                    for (int s = 0; s < M; ++s)
                        if (j0 + s == pj) { sh.bv[l][e][0] = vx[s]; sh.bv[l][e][1] = vy[s]; sh.bv[l][e][2] = vth[s]; }
                }
            }
        }
        __syncthreads();
        if (tid == 0) { px = py = pth = 0.0; }
        else { px = sh.prev[0][tid - 1]; py = sh.prev[1][tid - 1]; pth = sh.prev[2][tid - 1]; }
#pragma unroll
        for (int l = 0; l < NL; ++l)
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int k = 0; k < 3; ++k) lv[l][e][k] = sh.bv[l][e][k];
    };

    // || J v ||^2_Omega  = v^T H v   (partial sum of this thread; loops added by the caller)
    auto quad_form_partial = [&](const double (&vx)[M], const double (&vy)[M], const double (&vth)[M],
                                 double px, double py, double pth) -> double {
        double acc = 0.0;
#pragma unroll
        for (int s = 0; s < M; ++s) {
            if (!valid[s]) continue;
            const Pose2& a = s == 0 ? prevPose : X[s > 0 ? s - 1 : 0];
            const double ax = s == 0 ? px : vx[s > 0 ? s - 1 : 0];
            const double ay = s == 0 ? py : vy[s > 0 ? s - 1 : 0];
            const double ath = s == 0 ? pth : vth[s > 0 ? s - 1 : 0];
            const double dx = X[s].x - a.x, dy = X[s].y - a.y;
            const double rx = a.c * dx + a.s * dy, ry = -a.s * dx + a.c * dy;
            const double ddx = vx[s] - ax, ddy = vy[s] - ay;
            // R_a^T (dv_t) - J r v_theta(a),  J r = (-ry, rx)
            const double lx = a.c * ddx + a.s * ddy + ry * ath;
            const double ly = -a.s * ddx + a.c * ddy - rx * ath;
            const double cz = ld(F_CZ, s), sz = ld(F_SZ, s);
            const double wx = cz * lx + sz * ly, wy = -sz * lx + cz * ly, wth = vth[s] - ath;
            Sym3 om = load_sym3(P.chain, P.estride, F_OM, EK(s));
            acc += om.quad(wx, wy, wth);
        }
        return acc;
    };
    auto quad_form_loops = [&](const double (&lv)[NL][2][3]) -> double {
        double acc = 0.0;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const Pose2& a = lp[l].pf;
            const Pose2& b = lp[l].pt;
            const double dx = b.x - a.x, dy = b.y - a.y;
            const double rx = a.c * dx + a.s * dy, ry = -a.s * dx + a.c * dy;
            const double ddx = lv[l][1][0] - lv[l][0][0], ddy = lv[l][1][1] - lv[l][0][1];
            const double ath = lv[l][0][2];
            const double lx = a.c * ddx + a.s * ddy + ry * ath;
            const double ly = -a.s * ddx + a.c * ddy - rx * ath;
            const double wx = lp[l].cz * lx + lp[l].sz * ly, wy = -lp[l].sz * lx + lp[l].cz * ly;
            const double wth = lv[l][1][2] - ath;
            acc += lp[l].om.quad(wx, wy, wth);
        }
        return acc;
    };

    // ---------------- dog-leg (g2o OptimizationAlgorithmDogleg::solve) ----------------
    double delta = 1e4;
    const int maxTrials = 100;
    int it_done = 0, tries_total = 0, flags = 0;

    double currentChi = eval_errors();               // consensus_utils.cpp:11

    for (int it = 0; it < iterations; ++it) {
        // errors at the current state are valid here (currentChi)
        // ---- b = -J^T Om e ----
        double mfx, mfy, mfth;                        // m of my first slot (handed to tid-1)
        {
            double gxs[M], gys[M], gths[M], mx[M], my[M], mth[M];
#pragma unroll
            for (int s = 0; s < M; ++s) {
                gxs[s] = gys[s] = gths[s] = mx[s] = my[s] = mth[s] = 0.0;
                if (!valid[s]) continue;
                const Pose2& a = s == 0 ? prevPose : X[s > 0 ? s - 1 : 0];
                Sym3 om = load_sym3(P.chain, P.estride, F_OM, EK(s));
                double qx, qy, qth;
                om.mul(ex[s], ey[s], eth[s], qx, qy, qth);
                const double cz = ld(F_CZ, s), sz = ld(F_SZ, s);
                const double cP = a.c * cz - a.s * sz, sP = a.s * cz + a.c * sz;
                gxs[s] = cP * qx - sP * qy;
                gys[s] = sP * qx + cP * qy;
                gths[s] = qth;
                const double dx = X[s].x - a.x, dy = X[s].y - a.y;
                mx[s] = gxs[s];
                my[s] = gys[s];
                mth[s] = gths[s] + (-dy * gxs[s] + dx * gys[s]);
            }
            mfx = mx[0]; mfy = my[0]; mfth = mth[0];
            __syncthreads();
            sh.nxt[0][tid] = mfx; sh.nxt[1][tid] = mfy; sh.nxt[2][tid] = mfth;
            __syncthreads();
            double nx = 0.0, ny = 0.0, nth = 0.0;     // m of pose j0+M (zero past the end)
            if (tid + 1 < T && j0 + M <= L) { nx = sh.nxt[0][tid + 1]; ny = sh.nxt[1][tid + 1]; nth = sh.nxt[2][tid + 1]; }
#pragma unroll
            for (int s = 0; s < M; ++s) {
                if (!valid[s]) { bx[s] = by[s] = bth[s] = 0.0; continue; }
                const bool last = (s == M - 1);
                const bool nextValid = last ? (j0 + M <= L) : valid[s + 1 < M ? s + 1 : s];
                const double ux = last ? nx : mx[s + 1 < M ? s + 1 : s];
                const double uy = last ? ny : my[s + 1 < M ? s + 1 : s];
                const double uth = last ? nth : mth[s + 1 < M ? s + 1 : s];
                bx[s] = (nextValid ? ux : 0.0) - gxs[s];
                by[s] = (nextValid ? uy : 0.0) - gys[s];
                bth[s] = (nextValid ? uth : 0.0) - gths[s];
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    if (j0 + s == lp[l].t) { bx[s] -= lp[l].gx; by[s] -= lp[l].gy; bth[s] -= lp[l].gth; }
                    if (j0 + s == lp[l].f) {
                        const double dx = lp[l].pt.x - lp[l].pf.x, dy = lp[l].pt.y - lp[l].pf.y;
                        bx[s] += lp[l].gx; by[s] += lp[l].gy;
                        bth[s] += lp[l].gth + (-dy * lp[l].gx + dx * lp[l].gy);
                    }
                }
            }
        }
        // ---- alpha = b^T b / b^T H b ----
        double bb, bHb;
        {
            double pbx, pby, pbth, lv[NL][2][3];
            exchange_vec(bx, by, bth, pbx, pby, pbth, lv);
            double part[2] = {0.0, 0.0};
#pragma unroll
            for (int s = 0; s < M; ++s)
                if (valid[s]) part[0] += bx[s] * bx[s] + by[s] * by[s] + bth[s] * bth[s];
            part[1] = quad_form_partial(bx, by, bth, pbx, pby, pbth);
            block_sum<W, 2>(part, sh.red);
            bb = part[0];
            bHb = part[1] + quad_form_loops(lv);
        }
        const double alpha = bb / bHb;
        const double hsdNorm = sqrt(alpha * alpha * bb);

        // ---- Gauss-Newton step through the chain closed form ----
        double hgnNorm;
        {
            constexpr int NS = NL * 3;
            constexpr int KR = NS + NS * (NS + 1) / 2;
            double part[KR];
#pragma unroll
            for (int k = 0; k < KR; ++k) part[k] = 0.0;
            double G[M][NL][6];                       // per slot/loop: cR, sR (rotation), kx, ky, on(1/0)*sigma
#pragma unroll
            for (int s = 0; s < M; ++s) {
                const Pose2& a = s == 0 ? prevPose : X[s > 0 ? s - 1 : 0];
                const double cz = valid[s] ? ld(F_CZ, s) : 1.0, sz = valid[s] ? ld(F_SZ, s) : 0.0;
                const double cP = a.c * cz - a.s * sz, sP = a.s * cz + a.c * sz;
                Sym3 sg = load_sym3(P.chain, P.estride, F_SG, valid[s] ? EK(s) : lo_abs);
                double Gm[NL][3][3];
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    const int j = j0 + s;
                    const bool on = valid[s] && j > lp[l].lo && j <= lp[l].hi;
                    const double sgn = on ? lp[l].sigma : 0.0;
                    // Rg = Rzl^T R_f^T P_j ; kv = Rzl^T R_f^T J (t_to - t_j)
                    const double cf = lp[l].pf.c, sf = lp[l].pf.s;
                    const double cq = cf * cP + sf * sP, sq = cf * sP - sf * cP;       // R_f^T P
                    const double cR = lp[l].cz * cq + lp[l].sz * sq, sR = lp[l].cz * sq - lp[l].sz * cq;
                    const double jx = -(lp[l].pt.y - X[s].y), jy = (lp[l].pt.x - X[s].x);
                    const double fx = cf * jx + sf * jy, fy = -sf * jx + cf * jy;
                    const double kx = lp[l].cz * fx + lp[l].sz * fy, ky = -lp[l].sz * fx + lp[l].cz * fy;
                    G[s][l][0] = cR; G[s][l][1] = sR; G[s][l][2] = kx; G[s][l][3] = ky; G[s][l][4] = sgn;
                    Gm[l][0][0] = sgn * cR; Gm[l][0][1] = -sgn * sR; Gm[l][0][2] = sgn * kx;
                    Gm[l][1][0] = sgn * sR; Gm[l][1][1] = sgn * cR;  Gm[l][1][2] = sgn * ky;
                    Gm[l][2][0] = 0.0;      Gm[l][2][1] = 0.0;       Gm[l][2][2] = sgn;
                    // d_l partial: sum_j G e_j
#pragma unroll
                    for (int r = 0; r < 3; ++r)
                        part[l * 3 + r] += Gm[l][r][0] * ex[s] + Gm[l][r][1] * ey[s] + Gm[l][r][2] * eth[s];
                }
                // S partial: G_l Sg G_l'^T  (upper triangle of the NS x NS matrix, row-major packed)
                double Hm[NL][3][3];
#pragma unroll
                for (int l = 0; l < NL; ++l)
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const double g0 = Gm[l][r][0], g1 = Gm[l][r][1], g2 = Gm[l][r][2];
                        Hm[l][r][0] = g0 * sg.a00 + g1 * sg.a01 + g2 * sg.a02;
                        Hm[l][r][1] = g0 * sg.a01 + g1 * sg.a11 + g2 * sg.a12;
                        Hm[l][r][2] = g0 * sg.a02 + g1 * sg.a12 + g2 * sg.a22;
                    }
                int idx = NS;
#pragma unroll
                for (int r = 0; r < NS; ++r)
#pragma unroll
                    for (int c = r; c < NS; ++c) {
                        const int l1 = r / 3, r1 = r % 3, l2 = c / 3, r2 = c % 3;
                        part[idx] += Hm[l1][r1][0] * Gm[l2][r2][0] + Hm[l1][r1][1] * Gm[l2][r2][1] +
                                     Hm[l1][r1][2] * Gm[l2][r2][2];
                        ++idx;
                    }
            }
            block_sum<W, KR>(part, sh.red);
            double S[NS][NS], mu[NS];
            {
                int idx = NS;
#pragma unroll
                for (int r = 0; r < NS; ++r)
#pragma unroll
                    for (int c = r; c < NS; ++c) { S[r][c] = part[idx]; S[c][r] = part[idx]; ++idx; }
            }
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                S[3 * l + 0][3 * l + 0] += lp[l].sg.a00; S[3 * l + 0][3 * l + 1] += lp[l].sg.a01; S[3 * l + 0][3 * l + 2] += lp[l].sg.a02;
                S[3 * l + 1][3 * l + 0] += lp[l].sg.a01; S[3 * l + 1][3 * l + 1] += lp[l].sg.a11; S[3 * l + 1][3 * l + 2] += lp[l].sg.a12;
                S[3 * l + 2][3 * l + 0] += lp[l].sg.a02; S[3 * l + 2][3 * l + 1] += lp[l].sg.a12; S[3 * l + 2][3 * l + 2] += lp[l].sg.a22;
                mu[3 * l + 0] = lp[l].ex - part[3 * l + 0];
                mu[3 * l + 1] = lp[l].ey - part[3 * l + 1];
                mu[3 * l + 2] = lp[l].eth - part[3 * l + 2];
            }
            if (!chol_solve<NS>(S, mu)) { flags |= 2; break; }

            // u_j = -Sg_j sum_l G_lj^T mu_l - e_j ; rho_j = (P_j u_t, u_theta)
            double rx_[M], ry_[M], rth_[M];
            double tot[1] = {0.0};
#pragma unroll
            for (int s = 0; s < M; ++s) {
                rx_[s] = ry_[s] = rth_[s] = 0.0;
                if (!valid[s]) continue;
                const Pose2& a = s == 0 ? prevPose : X[s > 0 ? s - 1 : 0];
                double wx = 0.0, wy = 0.0, wth = 0.0;  // sum_l G^T mu
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    const double sgn = G[s][l][4], cR = G[s][l][0], sR = G[s][l][1], kx = G[s][l][2], ky = G[s][l][3];
                    const double m0 = mu[3 * l], m1 = mu[3 * l + 1], m2 = mu[3 * l + 2];
                    wx += sgn * (cR * m0 + sR * m1);
                    wy += sgn * (-sR * m0 + cR * m1);
                    wth += sgn * (kx * m0 + ky * m1 + m2);
                }
                Sym3 sg = load_sym3(P.chain, P.estride, F_SG, EK(s));
                double vx, vy, vth;
                sg.mul(wx, wy, wth, vx, vy, vth);
                const double ux = -vx - ex[s], uy = -vy - ey[s], uth = -vth - eth[s];
                const double cz = ld(F_CZ, s), sz = ld(F_SZ, s);
                const double cP = a.c * cz - a.s * sz, sP = a.s * cz + a.c * sz;
                rx_[s] = cP * ux - sP * uy;
                ry_[s] = sP * ux + cP * uy;
                rth_[s] = uth;
                tot[0] += uth;
            }
            // theta prefix sum
            double exc1[1];
            block_exclusive_scan<W, 1>(tot, exc1, sh.red);
            double run = exc1[0];
            double tt[2] = {0.0, 0.0};
#pragma unroll
            for (int s = 0; s < M; ++s) {
                if (!valid[s]) { hth[s] = 0.0; continue; }
                const Pose2& a = s == 0 ? prevPose : X[s > 0 ? s - 1 : 0];
                const double dx = X[s].x - a.x, dy = X[s].y - a.y;
                // term = rho_t + J dt * h_theta(j-1)
                rx_[s] += -dy * run;
                ry_[s] += dx * run;
                run += rth_[s];
                hth[s] = run;
                tt[0] += rx_[s];
                tt[1] += ry_[s];
            }
            double exc2[2];
            block_exclusive_scan<W, 2>(tt, exc2, sh.red);
            double runx = exc2[0], runy = exc2[1];
            double nrm[1] = {0.0};
#pragma unroll
            for (int s = 0; s < M; ++s) {
                if (!valid[s]) { hx[s] = hy[s] = 0.0; continue; }
                runx += rx_[s]; runy += ry_[s];
                hx[s] = runx; hy[s] = runy;
                nrm[0] += hx[s] * hx[s] + hy[s] * hy[s] + hth[s] * hth[s];
            }
            block_sum<W, 1>(nrm, sh.red);
            hgnNorm = sqrt(nrm[0]);
        }

        // ---- trial loop ----
        bool goodStep = false;
        int numTries = 0;
        do {
            ++numTries;
            int stepType;                             // 0 GN, 1 SD, 2 DL
            double beta = 0.0, sdScale = 0.0;
            if (hgnNorm < delta) stepType = 0;
            else if (hsdNorm > delta) { stepType = 1; sdScale = delta / hsdNorm; }
            else {
                stepType = 2;
                double part[2] = {0.0, 0.0};          // c = hsd.(hgn-hsd), |hgn-hsd|^2
#pragma unroll
                for (int s = 0; s < M; ++s) {
                    if (!valid[s]) continue;
                    const double sx = alpha * bx[s], sy = alpha * by[s], sth = alpha * bth[s];
                    const double ax = hx[s] - sx, ay = hy[s] - sy, ath = hth[s] - sth;
                    part[0] += sx * ax + sy * ay + sth * ath;
                    part[1] += ax * ax + ay * ay + ath * ath;
                }
                block_sum<W, 2>(part, sh.red);
                const double c = part[0], bma = part[1], hsdSq = alpha * alpha * bb;
                if (c <= 0.) beta = (-c + sqrt(c * c + bma * (delta * delta - hsdSq))) / bma;
                else beta = (delta * delta - hsdSq) / (c + sqrt(c * c + bma * (delta * delta - hsdSq)));
            }
            bool changed = false;
#pragma unroll
            for (int s = 0; s < M; ++s) {
                if (stepType == 0) { dlx[s] = hx[s]; dly[s] = hy[s]; dlth[s] = hth[s]; }
                else if (stepType == 1) {
                    dlx[s] = sdScale * (alpha * bx[s]); dly[s] = sdScale * (alpha * by[s]); dlth[s] = sdScale * (alpha * bth[s]);
                } else {
                    const double sx = alpha * bx[s], sy = alpha * by[s], sth = alpha * bth[s];
                    dlx[s] = sx + beta * (hx[s] - sx); dly[s] = sy + beta * (hy[s] - sy); dlth[s] = sth + beta * (hth[s] - sth);
                }
                if (!valid[s]) { dlx[s] = dly[s] = dlth[s] = 0.0; continue; }
                const double nx = X[s].x + dlx[s], ny = X[s].y + dly[s];
                const double nth = normalize_theta(X[s].th + dlth[s]);
                changed |= (nx != X[s].x) || (ny != X[s].y) || (nth != X[s].th);
            }
            changed = block_any<W>(changed, sh.flag);

            double rho, hdlNorm;
            if (!changed) {
                // bit-identical state => newChi == currentChi => rho = 0 (rejected)
                rho = 0.0;
                hdlNorm = 0.0;                        // only used when rho > 0.75
                if (stepType == 1) numTries = maxTrials;  // all later SD trials are no-ops too
            } else {
                // linear gain = -hdl^T H hdl + 2 b^T hdl
                double pdx, pdy, pdth, lv[NL][2][3];
                exchange_vec(dlx, dly, dlth, pdx, pdy, pdth, lv);
                double part[3] = {0.0, 0.0, 0.0};
                part[0] = quad_form_partial(dlx, dly, dlth, pdx, pdy, pdth);
#pragma unroll
                for (int s = 0; s < M; ++s)
                    if (valid[s]) {
                        part[1] += bx[s] * dlx[s] + by[s] * dly[s] + bth[s] * dlth[s];
                        part[2] += dlx[s] * dlx[s] + dly[s] * dly[s] + dlth[s] * dlth[s];
                    }
                block_sum<W, 3>(part, sh.red);
                double linearGain = -1 * (part[0] + quad_form_loops(lv)) + 2 * part[1];
                hdlNorm = sqrt(part[2]);
                // push, update, new errors
#pragma unroll
                for (int s = 0; s < M; ++s) {
                    Xb[s] = X[s];
                    if (!valid[s]) continue;
                    X[s].x += dlx[s]; X[s].y += dly[s];
                    X[s].th = normalize_theta(X[s].th + dlth[s]);
                    sincos(X[s].th, &X[s].s, &X[s].c);
                }
                const double newChi = eval_errors();
                const double nonLinearGain = currentChi - newChi;
                if (fabs(linearGain) < 1e-12) linearGain = 1e-12;
                rho = nonLinearGain / linearGain;
                if (rho > 0) { goodStep = true; currentChi = newChi; }
                else {
#pragma unroll
                    for (int s = 0; s < M; ++s) X[s] = Xb[s];
                    (void)eval_errors();              // pop: errors of the restored state
                }
            }
            if (rho > 0.75) delta = fmax(delta, 3 * hdlNorm);
            else if (rho < 0.25) delta *= 0.5;
            if (!goodStep && stepType == 0) {
                // identical GN trial repeats while hgnNorm < delta: each halves delta
                while (numTries < maxTrials && hgnNorm < delta) { ++numTries; delta *= 0.5; }
            }
        } while (!goodStep && numTries < maxTrials);
        it_done = it + 1;
        tries_total += numTries;
        if (numTries == maxTrials || !goodStep) { flags |= 1; break; }
    }

    // ---- per-edge chi2 (consensus_utils.cpp:15-19) ----
    double mx = 0.0;
#pragma unroll
    for (int s = 0; s < M; ++s) {
        if (!valid[s]) continue;
        Sym3 om = load_sym3(P.chain, P.estride, F_OM, EK(s));
        const double c = om.quad(ex[s], ey[s], eth[s]);
        mx = (c > mx || c != c) ? c : mx;
    }
    // NaN-propagating max: a NaN chi2 must survive (g2o's "chi2 > th" is false for NaN)
    const bool anyNan = block_any<W>(mx != mx, sh.flag);
    mx = block_max<W>(mx == mx ? mx : 0.0, sh.red);
#pragma unroll
    for (int l = 0; l < NL; ++l) mx = (lp[l].chi > mx || lp[l].chi != lp[l].chi) ? lp[l].chi : mx;
    if (anyNan) mx = __longlong_as_double(0x7ff8000000000000ll);
    res.max_chi2 = mx;
    res.chi2_total = currentChi;
    res.iterations = it_done;
    res.tries = tries_total;
    res.flags = flags;
    res.evals = evals;
}

}  // namespace ipc
