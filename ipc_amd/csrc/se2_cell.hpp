// SE(2) consistency cell solver -- one workgroup per cell, hand-written for gfx950 (wave64).
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
//   * Pose j (1..L) of the chain lives in the registers of ONE lane: wave w owns poses
//     [w*64*M+1, (w+1)*64*M], slot s / lane l holds pose w*64*M + s*64 + l + 1.  Consecutive
//     lanes hold consecutive poses, so every load of the shared odometry chain is a fully
//     coalesced 512-byte wave access and the chain neighbour (j-1 / j+1) is one DPP
//     wave_shr/wave_shl away; only the 2*W wave-boundary values cross LDS.
//   * The Gauss-Newton step H h = b is NOT obtained by factoring the (block tridiagonal +
//     arrow) H.  With u = Jc h (Jc = square block-bidiagonal Jacobian of the odometry chain)
//     the normal equations become  (Om_c + G^T Om_l G) u = Jc^-T b  with G = Jl Jc^-1, and for
//     SE(2) Jc^-1 has the closed form "propagate a perturbation down the chain":
//         h_theta(i) = sum_{j<=i} rho_theta(j),
//         h_t(i)     = sum_{j<=i} [ rho_t(j) + J (t_j - t_{j-1}) h_theta(j-1) ],   J=[[0,-1],[1,0]]
//     so G is element-wise, the capacitance matrix S = Cov_l + sum_j G_j Cov_j G_j^T is a
//     3x3 / 6x6 reduction, and h follows from prefix sums (DPP wave scans + one LDS hop).
//   * Dog-leg bookkeeping follows g2o exactly (delta, rho, <=100 trials, Terminate).  A trial
//     is evaluated WITHOUT committing (trial poses/errors sit beside the current ones), so a
//     rejected trial costs no restore pass.  Two shortcuts are taken that are bit-exact w.r.t.
//     the un-shortcut loop:
//       - a rejected Gauss-Newton trial repeats identically while ||h_gn|| < delta, so those
//         repeats only halve delta and count trials;
//       - a steepest-descent trial whose update leaves every pose bit-identical has
//         newChi == currentChi (rho = 0, rejected) and so have all later, halved ones: the
//         trial counter is fast-forwarded to Terminate.
//   * The dog-leg's linear gain  -h^T H h + 2 b^T h  of a trial step is assembled from four
//     per-iteration scalars (b^T b, b^T H b, b^T h_gn, h_gn^T H h_gn; b^T H h_gn = b^T b because
//     H h_gn = b), since every trial step is a combination of b and h_gn.  A trial therefore
//     costs one residual pass and ONE reduced scalar (the new chi2).
//   * Every phase is: element-wise work -> one __syncthreads -> read what the other waves
//     published.  All LDS scratch is double buffered on the phase parity, which makes one
//     barrier per phase sufficient.  Wide reductions (the 11 / 29 capacitance partials) use
//     gfx950's v_permlane32_swap / v_permlane16_swap to reduce four values per register, and
//     the 6x6 capacitance solve runs on wave 0 only.
#pragma once
#include <type_traits>

#include "block_prims.hpp"

namespace ipc {

constexpr double kPi = 3.14159265358979323846;

// chain / candidate record fields (structure of arrays, field-major)
enum Se2Field { F_TZX = 0, F_TZY, F_CZ, F_SZ, F_THZ, F_OM = 5, F_SG = 11, F_NFIELDS = 17 };

struct Se2View {
    const double* chain;      // [F_NFIELDS][estride], index = edge k (joins pose k -> k+1)
    int estride;
    const double* chain_rec;  // the same values record-major: [edge][F_NFIELDS] (+ zero padding)
    const double* pose0;      // [3][V] open-loop poses x, y, theta
    int V;
    const double* cand;       // [F_NFIELDS][cstride] per loop candidate
    int cstride;
    const int* cand_from;     // candidate "from" / "to" vertex ids
    const int* cand_to;
    double* dbg;              // debug side channel (NULL in production)
    // Convergence test of the dog-leg (0 = off: g2o's literal loop).  g2o has none: a converged problem leaves
    // optimize() only through Terminate, i.e. after an iteration whose trials all fail -- about 40 evaluated
    // trials (delta halves until no pose moves any more), more than half of all residual passes of a matrix.
    // With the Gauss-Newton step h of the current linearisation, an edge's chi2 moves by at most
    // 2 sqrt(chi2_e h^T H h) = 2 sqrt(chi2_e b^T h).  The solve stops (with the Terminate flag) when
    //   * the previous iteration accepted the full Gauss-Newton step at its first trial and this one's lies
    //     inside the trust region (Newton regime: steps shrink quadratically), and
    //   * |b^T h| * #edges < term_eps * chi2_total   (chi2_max >= chi2_total / #edges),
    // i.e. when the next step could change no edge's chi2 by more than 2 sqrt(term_eps) relative (6e-7
    // worst case with the default 1e-13).  A problem that is still moving on trust-region-limited steps
    // (slow crawls along a flat valley, up to the iteration cap) runs g2o's loop unchanged: cutting those
    // short would move their per-edge chi2 by ~5e-5.
    double term_eps;
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

// sin/cos for |theta| <= pi (every pose angle is kept normalised to [-pi, pi)): Cody-Waite
// reduction by multiples of pi/2 (two-term, exact with FMA for |n| <= 2) and the classic
// degree-13/14 minimax kernels on [-pi/4, pi/4].  Branch-free, ~45 VALU ops, absolute error
// < 1.2e-16 -- OCML's sincos carries a Payne-Hanek large-argument path that costs ~60 VGPRs we
// need elsewhere.
__device__ __forceinline__ void sincos_pi(double th, double& sn, double& cs)
{
    const double n = rint(th * 6.36619772367581382433e-01);            // th * 2/pi
    double r = fma(-n, 1.57079632673412561417e+00, th);                // pio2_1 (33 bits)
    r = fma(-n, 6.07710050650619224932e-11, r);                        // pio2_1t
    const double z = r * r;
    // sin kernel
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double rs = fma(z, fma(z, fma(z, fma(z, S6, S5), S4), S3), S2);
    const double sr = fma(z * r, fma(z, rs, S1), r);
    // cos kernel
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double rc = z * fma(z, fma(z, fma(z, fma(z, fma(z, C6, C5), C4), C3), C2), C1);
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    const double cr = w + (((1.0 - w) - hz) + z * rc);
    const int q = ((int)n) & 3;
    const double s0 = (q & 1) ? cr : sr;
    const double c0 = (q & 1) ? sr : cr;
    sn = (q & 2) ? -s0 : s0;
    cs = ((q + 1) & 2) ? -c0 : c0;
}

// theta mod 2 pi into [-pi, pi] in three VALU ops (mul, rint, fma).  Equal to g2o's
// normalize_theta except at the exact boundary value +pi (which g2o maps to -pi; same angle,
// same chi2), valid for |theta| < ~1e15.
__device__ __forceinline__ double wrap_pi(double theta)
{
    return fma(-6.283185307179586476925, rint(theta * 0.15915494309189533577), theta);
}

// (cos, sin) of theta + d from (c, s) of theta for |d| < 2^-6: Taylor kernels to d^7 / d^8
// (truncation < 2e-22), applied as c' = c + (c (cos d - 1) - s sin d) so no 1 + tiny rounding.
__device__ __forceinline__ void rotate_small(double c, double s, double d, double& cn, double& sn)
{
    const double z = d * d;
    const double sd = d * fma(z, fma(z, fma(z, -1.0 / 5040.0, 1.0 / 120.0), -1.0 / 6.0), 1.0);
    const double cm = z * fma(z, fma(z, fma(z, 1.0 / 40320.0, -1.0 / 720.0), 1.0 / 24.0), -0.5);
    cn = fma(-s, sd, fma(c, cm, c));
    sn = fma(c, sd, fma(s, cm, s));
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

// error of an SE2 edge a -> b with measurement (tz, cz, sz, thz)
__device__ __forceinline__ void se2_error(const Pose2& a, const Pose2& b, double tzx, double tzy,
                                          double cz, double sz, double thz, double& ex, double& ey,
                                          double& eth)
{
    const double dx = b.x - a.x, dy = b.y - a.y;
    const double rx = a.c * dx + a.s * dy;
    const double ry = -a.s * dx + a.c * dy;
    const double lx = rx - tzx, ly = ry - tzy;
    ex = cz * lx + sz * ly;
    ey = -sz * lx + cz * ly;
    eth = wrap_pi((b.th - a.th) - thz);
}

// w = J_edge applied to the pose perturbations (va at pose a, vb at pose b) of an edge a -> b
// with current poses a, b and measurement rotation (cz, sz):
//   w_t = Rz^T [ R_a^T (vb_t - va_t) - J r va_theta ],  r = R_a^T (t_b - t_a);  w_th = vb_th - va_th
__device__ __forceinline__ void se2_apply_J(const Pose2& a, const Pose2& b, double cz, double sz,
                                            double vax, double vay, double vath, double vbx, double vby,
                                            double vbth, double& wx, double& wy, double& wth)
{
    const double dx = b.x - a.x, dy = b.y - a.y;
    const double rx = a.c * dx + a.s * dy, ry = -a.s * dx + a.c * dy;
    const double ddx = vbx - vax, ddy = vby - vay;
    const double lx = a.c * ddx + a.s * ddy + ry * vath;
    const double ly = -a.s * ddx + a.c * ddy - rx * vath;
    wx = cz * lx + sz * ly;
    wy = -sz * lx + cz * ly;
    wth = vbth - vath;
}

// SPD solve by Cholesky (n = 3 or 6), full row-major, overwritten.  One reciprocal per pivot.
template <int N>
__device__ __forceinline__ bool chol_solve(double (&A)[N][N], double (&b)[N])
{
    bool ok = true;
    double inv[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double sum = A[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) sum -= A[i][k] * A[j][k];
            if (j < i) A[i][j] = sum * inv[j];
            else {
                if (!(sum > 0)) ok = false;
                A[i][i] = sqrt(sum);
                inv[i] = 1.0 / A[i][i];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double sum = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) sum -= A[i][k] * b[k];
        b[i] = sum * inv[i];
    }
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        double sum = b[i];
#pragma unroll
        for (int k = i + 1; k < N; ++k) sum -= A[k][i] * b[k];
        b[i] = sum * inv[i];
    }
    return ok;
}

struct CellResult {
    double max_chi2;
    double chi2_total;
    int iterations;
    int tries;
    int flags;                // bit0: dog-leg Terminate, bit1: capacitance not PD (Fail)
    int evals;                // trial evaluations (sincos + residual passes) executed
};

// loop-closure edge: constants and per-state values, all in LDS (uniform data)
struct LoopConst {
    int f, t, lo, hi;         // local pose indices 0..L; lo/hi = min/max(f,t)
    double sigma;             // +1 if t > f else -1
    double tzx, tzy, cz, sz, thz;
    double om[6], sg[6];
};
struct LoopState {
    double pf[5], pt[5];      // end-point poses (x y th c s)
    double e[3];              // error
    double g[3];              // world-frame force (R_f Rz q_t, q_theta), q = Om e
    double chi;
};

template <int W, int NL>
struct Se2Scratch {           // per-phase hand-off, double buffered
    double red[32 * 16];      // wide reductions: [wave][32]; narrow ones: [value][16 waves]
    double hi_pose[W][5];     // pose held by lane 63 / last slot of each wave
    double hi_vec[W][3];      // vector held by lane 63 / last slot
    double lo_vec[W][3];      // vector held by lane 0 / slot 0
    double lvec[NL][2][3];    // vectors at the loop end points (from, to)
    double scan[W][5];        // per-wave scan totals
    double sol[NL * 3 + 3 + NL];   // nu per loop, b^T b, b^T H b (odometry part), ok flag, loop parts of b^T H b
};

// Chain records of the cell are staged once into LDS (17 doubles per edge) when the variant's
// capacity allows it (<= 1024 poses: 136 KB of the CU's 160 KB); every later use is then a
// conflict-free ds_read_b64 instead of an L2 round trip in the middle of a latency-bound phase.
template <int W, int M>
struct Se2Cap {
    static constexpr int CAP = 64 * W * M;
    // all 17 fields up to 1024 poses; up to 1536 poses only the 11 fields of the residual pass
    // (measurement + Omega), Sigma then stays an L2 read (used twice per outer iteration)
    // two-wave variants (W == 2) are meant to run two cells per CU (one wave per SIMD, 512 VGPRs
    // each), so they only take half of the LDS: the 11 residual-pass fields up to 768 poses
    static constexpr int NSTAGED = W == 2 ? (CAP <= 512 ? (int)F_NFIELDS : (CAP <= 768 ? (int)F_SG : 0))
                                          : (CAP <= 1024 ? (int)F_NFIELDS : (CAP <= 1536 ? (int)F_SG : 0));
    static constexpr int ROWS = NSTAGED > 0 ? CAP : 1;
    static constexpr int FROWS = NSTAGED > 0 ? NSTAGED : 1;
};

template <int W, int M, int NL>
struct Se2Shared {
    Se2Scratch<W, NL> scr[2];
    LoopConst lc[NL];
    LoopState ls[2][NL];      // [buffer][loop]; `cur` selects the committed one
    double cst[Se2Cap<W, M>::FROWS][Se2Cap<W, M>::ROWS];
    double wtmp[W][64];       // wave-private temporaries of the lane-parallel capacitance solve
};

template <int W, int M, int NL>
__device__ void se2_solve_cell(const Se2View& P, int lo_abs, int L, const int (&cand)[2], int iterations,
                               Se2Shared<W, M, NL>& sh, CellResult& res)
{
    constexpr int NS = NL * 3;
    const double term_scale = P.term_eps / (double)(L + NL);   // 0: the test is off
    bool lastGN = false;
    constexpr int KR = 2 + NS + NS * (NS + 1) / 2;   // b^T b, b^T H b, d, S (upper triangle)
    static_assert(KR <= 32 && W <= 16, "reduction scratch layout");
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int jbase = 1 + wave * 64 * M + lane;      // pose index of slot 0

    // ---------------- loop constants -> LDS ----------------
    if (tid < NL) {
        const int l = tid, c = cand[l];
        LoopConst& q = sh.lc[l];
        q.f = P.cand_from[c] - lo_abs;
        q.t = P.cand_to[c] - lo_abs;
        q.lo = min(q.f, q.t);
        q.hi = max(q.f, q.t);
        q.sigma = q.t > q.f ? 1.0 : -1.0;
        q.tzx = P.cand[(size_t)F_TZX * P.cstride + c];
        q.tzy = P.cand[(size_t)F_TZY * P.cstride + c];
        q.cz = P.cand[(size_t)F_CZ * P.cstride + c];
        q.sz = P.cand[(size_t)F_SZ * P.cstride + c];
        q.thz = P.cand[(size_t)F_THZ * P.cstride + c];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            q.om[k] = P.cand[(size_t)(F_OM + k) * P.cstride + c];
            q.sg[k] = P.cand[(size_t)(F_SG + k) * P.cstride + c];
        }
    }

    // ---------------- per-lane state ----------------
    Pose2 X[M];                                      // committed poses
    Pose2 Xn[M];                                     // trial poses
    double ex[M], ey[M], eth[M];                     // committed odometry errors of edge j (j-1 -> j)
    double bx[M], by[M], bth[M];                     // b = -J^T Om e
    double hx[M], hy[M], hth[M];                     // Gauss-Newton step
    bool valid[M];
    int ek[M];                                       // absolute edge index of the slot
    Pose2 gauge;
    gauge.x = P.pose0[lo_abs];
    gauge.y = P.pose0[(size_t)P.V + lo_abs];
    gauge.th = P.pose0[(size_t)2 * P.V + lo_abs];
    sincos_pi(gauge.th, gauge.s, gauge.c);
#pragma unroll
    for (int s = 0; s < M; ++s) {
        const int j = jbase + s * 64;
        valid[s] = j <= L;
        ek[s] = valid[s] ? lo_abs + j - 1 : lo_abs;
        const int ja = valid[s] ? lo_abs + j : lo_abs;
        Xn[s].x = P.pose0[ja];
        Xn[s].y = P.pose0[(size_t)P.V + ja];
        Xn[s].th = P.pose0[(size_t)2 * P.V + ja];
        sincos_pi(Xn[s].th, Xn[s].s, Xn[s].c);
        X[s] = Xn[s];
        hx[s] = hy[s] = hth[s] = 0.0;
        bx[s] = by[s] = bth[s] = 0.0;
        ex[s] = ey[s] = eth[s] = 0.0;
    }
    __syncthreads();                                 // sh.lc visible
    int lf[NL], lt[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        lf[l] = __builtin_amdgcn_readfirstlane(sh.lc[l].f);
        lt[l] = __builtin_amdgcn_readfirstlane(sh.lc[l].t);
    }
    // gauge end points never change: write them into both state buffers once
    if (tid == 0) {
#pragma unroll
        for (int l = 0; l < NL; ++l)
#pragma unroll
            for (int bsel = 0; bsel < 2; ++bsel) {
                double* pf = sh.ls[bsel][l].pf;
                double* pt = sh.ls[bsel][l].pt;
                if (lf[l] == 0) { pf[0] = gauge.x; pf[1] = gauge.y; pf[2] = gauge.th; pf[3] = gauge.c; pf[4] = gauge.s; }
                if (lt[l] == 0) { pt[0] = gauge.x; pt[1] = gauge.y; pt[2] = gauge.th; pt[3] = gauge.c; pt[4] = gauge.s; }
            }
    }

    // The chain constants are deliberately re-read where used (coalesced, L1/L2 resident):
    // keeping all 17 per slot would cost 34 VGPRs per pose.  Loads use a uniform (SGPR) field
    // base + one 32-bit byte offset per slot, so no 64-bit per-field addresses are kept alive.
    // opaque() stops the compiler from hoisting those loads out of the iteration loop.
    unsigned eoff[M];
#pragma unroll
    for (int s = 0; s < M; ++s) eoff[s] = (unsigned)ek[s] << 3;
    // (the per-field bases are recomputed from `fstride` with two SALU ops per use instead of
    // pinning 17 SGPR pairs: the kernel is SGPR-bound otherwise)
    unsigned fstride = (unsigned)P.estride << 3;      // bytes between two fields
    constexpr int NSTAGED = Se2Cap<W, M>::NSTAGED;
    int jl[M];                                       // local edge index of the slot (0 when invalid)
#pragma unroll
    for (int s = 0; s < M; ++s) jl[s] = valid[s] ? jbase + s * 64 - 1 : 0;
    if constexpr (NSTAGED > 0) {
        for (int f = 0; f < NSTAGED; ++f)
            for (int i = tid; i < L; i += 64 * W) sh.cst[f][i] = P.chain[(size_t)f * P.estride + lo_abs + i];
        // visible after the next __syncthreads() (the first one of evaluate())
    }
    auto opaque = [&]() {
        if constexpr (NSTAGED < (int)F_NFIELDS) {
#pragma unroll
            for (int s = 0; s < M; ++s) asm volatile("" : "+v"(eoff[s]));
        }
    };
    auto ldc = [&](int field, int s) -> double {
        if (field < NSTAGED) return sh.cst[field < NSTAGED ? field : 0][jl[s]];
        const char* fb = reinterpret_cast<const char*>(P.chain) + (size_t)field * fstride;
        return *reinterpret_cast<const double*>(fb + eoff[s]);
    };
    auto ldsym = [&](int field0, int s) -> Sym3 {
        Sym3 m;
        m.a00 = ldc(field0 + 0, s); m.a01 = ldc(field0 + 1, s); m.a02 = ldc(field0 + 2, s);
        m.a11 = ldc(field0 + 3, s); m.a12 = ldc(field0 + 4, s); m.a22 = ldc(field0 + 5, s);
        return m;
    };

    int phase = 0;                                   // scratch parity
    int cur = 0;                                     // committed loop-state buffer

    // pose of chain neighbour j-1 for every slot, from poses Y (DPP + wave-boundary pose `edge`)
    auto prev_pose = [&](const Pose2 (&Y)[M], const Pose2& edge, Pose2 (&A)[M]) {
        Pose2 carry = edge;
#pragma unroll
        for (int s = 0; s < M; ++s) {
            A[s].x = lane_prev(Y[s].x, carry.x);
            A[s].y = lane_prev(Y[s].y, carry.y);
            A[s].th = lane_prev(Y[s].th, carry.th);
            A[s].c = lane_prev(Y[s].c, carry.c);
            A[s].s = lane_prev(Y[s].s, carry.s);
            if (s + 1 < M) {
                carry.x = read_lane(Y[s].x, 63); carry.y = read_lane(Y[s].y, 63); carry.th = read_lane(Y[s].th, 63);
                carry.c = read_lane(Y[s].c, 63); carry.s = read_lane(Y[s].s, 63);
            }
        }
    };
    // publish the wave's last pose (lane 63, slot M-1) and the loop end points of poses Y
    auto publish_poses = [&](const Pose2 (&Y)[M], Se2Scratch<W, NL>& S, int lsbuf) {
        if (lane == 63) {
            S.hi_pose[wave][0] = Y[M - 1].x; S.hi_pose[wave][1] = Y[M - 1].y; S.hi_pose[wave][2] = Y[M - 1].th;
            S.hi_pose[wave][3] = Y[M - 1].c; S.hi_pose[wave][4] = Y[M - 1].s;
        }
#pragma unroll
        for (int l = 0; l < NL; ++l)
#pragma unroll
            for (int s = 0; s < M; ++s) {
                const int j = jbase + s * 64;
                if (j == lf[l]) {
                    double* q = sh.ls[lsbuf][l].pf;
                    q[0] = Y[s].x; q[1] = Y[s].y; q[2] = Y[s].th; q[3] = Y[s].c; q[4] = Y[s].s;
                }
                if (j == lt[l]) {
                    double* q = sh.ls[lsbuf][l].pt;
                    q[0] = Y[s].x; q[1] = Y[s].y; q[2] = Y[s].th; q[3] = Y[s].c; q[4] = Y[s].s;
                }
            }
    };
    // publish a per-pose vector at the loop end points (the gauge end point carries zero)
    auto publish_endpoint_vec = [&](const double (&vx)[M], const double (&vy)[M], const double (&vth)[M],
                                    Se2Scratch<W, NL>& S) {
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            if (tid == 0) {
                if (lf[l] == 0) { S.lvec[l][0][0] = 0.0; S.lvec[l][0][1] = 0.0; S.lvec[l][0][2] = 0.0; }
                if (lt[l] == 0) { S.lvec[l][1][0] = 0.0; S.lvec[l][1][1] = 0.0; S.lvec[l][1][2] = 0.0; }
            }
#pragma unroll
            for (int s = 0; s < M; ++s) {
                const int j = jbase + s * 64;
                if (j == lf[l]) { S.lvec[l][0][0] = vx[s]; S.lvec[l][0][1] = vy[s]; S.lvec[l][0][2] = vth[s]; }
                if (j == lt[l]) { S.lvec[l][1][0] = vx[s]; S.lvec[l][1][1] = vy[s]; S.lvec[l][1][2] = vth[s]; }
            }
        }
    };
    auto edge_pose_of = [&](const Se2Scratch<W, NL>& S) -> Pose2 {
        Pose2 p = gauge;
        if (wave > 0) {
            p.x = S.hi_pose[wave - 1][0]; p.y = S.hi_pose[wave - 1][1]; p.th = S.hi_pose[wave - 1][2];
            p.c = S.hi_pose[wave - 1][3]; p.s = S.hi_pose[wave - 1][4];
        }
        p.x = uni(p.x); p.y = uni(p.y); p.th = uni(p.th); p.c = uni(p.c); p.s = uni(p.s);
        return p;
    };

    // loop l: error / force / chi2 from its end-point poses in state buffer `bsel` (one lane)
    auto loop_eval = [&](int l, int bsel) -> double {
        const LoopConst& q = sh.lc[l];
        LoopState& st = sh.ls[bsel][l];
        Pose2 a{st.pf[0], st.pf[1], st.pf[2], st.pf[3], st.pf[4]};
        Pose2 b{st.pt[0], st.pt[1], st.pt[2], st.pt[3], st.pt[4]};
        double e0, e1, e2;
        se2_error(a, b, q.tzx, q.tzy, q.cz, q.sz, q.thz, e0, e1, e2);
        Sym3 om{q.om[0], q.om[1], q.om[2], q.om[3], q.om[4], q.om[5]};
        double q0, q1, q2;
        om.mul(e0, e1, e2, q0, q1, q2);
        const double chi = e0 * q0 + e1 * q1 + e2 * q2;
        st.e[0] = e0; st.e[1] = e1; st.e[2] = e2;
        st.chi = chi;
        return chi;
    };
    // loop l: world-frame force g = (R_f Rz q_t, q_theta), q = Om e, of the committed state
    // (only needed once per outer iteration, so it is kept out of the trial evaluations)
    auto loop_force = [&](int l) {
        const LoopConst& q = sh.lc[l];
        LoopState& st = sh.ls[cur][l];
        Sym3 om{q.om[0], q.om[1], q.om[2], q.om[3], q.om[4], q.om[5]};
        double q0, q1, q2;
        om.mul(st.e[0], st.e[1], st.e[2], q0, q1, q2);
        const double cP = st.pf[3] * q.cz - st.pf[4] * q.sz, sP = st.pf[4] * q.cz + st.pf[3] * q.sz;
        st.g[0] = cP * q0 - sP * q1; st.g[1] = sP * q0 + cP * q1; st.g[2] = q2;
    };
    // loop l: w^T Om w with w = J_l applied to the end-point vectors in S.lvec (uniform data)
    auto loop_quad = [&](int l, const Se2Scratch<W, NL>& S) -> double {
        const LoopConst& q = sh.lc[l];
        const LoopState& st = sh.ls[cur][l];
        Pose2 a{st.pf[0], st.pf[1], st.pf[2], st.pf[3], st.pf[4]};
        Pose2 b{st.pt[0], st.pt[1], st.pt[2], st.pt[3], st.pt[4]};
        double wx, wy, wth;
        se2_apply_J(a, b, q.cz, q.sz, S.lvec[l][0][0], S.lvec[l][0][1], S.lvec[l][0][2], S.lvec[l][1][0],
                    S.lvec[l][1][1], S.lvec[l][1][2], wx, wy, wth);
        Sym3 om{q.om[0], q.om[1], q.om[2], q.om[3], q.om[4], q.om[5]};
        return om.quad(wx, wy, wth);
    };

    // One residual pass over poses Y: publishes the hand-off values, computes the odometry errors
    // into (ox,oy,oth) and the loop state into buffer `bsel`; returns total chi2 and whether any
    // lane flagged `changed`.  Two barriers.
    Pose2 edge = gauge;                              // committed pose of this wave's predecessor
    Pose2 edgeN = gauge;                             // same for the trial state
#ifdef IPC_PHASE_TIMING
    unsigned long long te[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tq = 0;
#define IPC_ETICK(k) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); te[k] += t_ - tq; tq = t_; }
#define IPC_ESTART() { tq = __builtin_amdgcn_s_memtime(); }
#else
#define IPC_ETICK(k)
#define IPC_ESTART()
#endif
    // keep: store the odometry errors (committed state); trials only need chi2 -- an accepted
    // trial recomputes its errors once (recommit_errors), which is cheaper than carrying a second
    // error array in registers through every phase.
    auto evaluate = [&](const Pose2 (&Y)[M], auto keep_c, int bsel, bool changed, bool& anyChanged) -> double {
        constexpr bool KEEP = decltype(keep_c)::value != 0;
        IPC_ESTART()
        Se2Scratch<W, NL>& S = sh.scr[phase & 1];
        publish_poses(Y, S, bsel);
        IPC_ETICK(0)
        __syncthreads();
        IPC_ETICK(1)
        edgeN = edge_pose_of(S);
        ++phase;
        Pose2 An[M];
        prev_pose(Y, edgeN, An);
        double part = 0.0;
#pragma unroll
        for (int s = 0; s < M; ++s) {
            if (!valid[s]) continue;
            double e0, e1, e2;
            se2_error(An[s], Y[s], ldc(F_TZX, s), ldc(F_TZY, s), ldc(F_CZ, s), ldc(F_SZ, s), ldc(F_THZ, s), e0, e1, e2);
            part += ldsym(F_OM, s).quad(e0, e1, e2);
            if (KEEP) { ex[s] = e0; ey[s] = e1; eth[s] = e2; }
        }
        IPC_ETICK(2)
        if (tid < NL) part += loop_eval(tid, bsel);
        part = wave_sum(part);
        const bool wchg = __ballot(changed) != 0ull;
        Se2Scratch<W, NL>& S2 = sh.scr[phase & 1];
        if (lane == 0) { S2.red[wave] = part; S2.red[16 + wave] = wchg ? 1.0 : 0.0; }
        IPC_ETICK(3)
        __syncthreads();
        IPC_ETICK(4)
        double tot[2];
        gather_totals<2>(S2.red, W, tot);
        ++phase;
        anyChanged = tot[1] != 0.0;
        IPC_ETICK(5)
        return tot[0];
    };

    // ---------------- initial errors (consensus_utils.cpp:11) ----------------
    int evals = 0;
    double currentChi;
    {
        bool dummy;
        currentChi = evaluate(X, std::integral_constant<int, 1>{}, cur, false, dummy);
        edge = edgeN;
        ++evals;
    }

    // ---------------- dog-leg (g2o OptimizationAlgorithmDogleg::solve) ----------------
    double delta = 1e4;
    const int maxTrials = 100;
    int it_done = 0, tries_total = 0, flags = 0;

#ifdef IPC_PHASE_TIMING
    unsigned long long tmA = 0, tmB = 0, tmC = 0, tmT = 0, tm0 = __builtin_amdgcn_s_memtime();
#define IPC_TICK(acc) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc += t_ - tm0; tm0 = t_; }
#else
#define IPC_TICK(acc)
#endif
    for (int it = 0; it < iterations; ++it) {
        opaque();
        IPC_TICK(tmT)
        // committed poses X (+ edge), errors e and loop state ls[cur] are consistent here
        Pose2 A[M];                                  // committed pose j-1 per slot
        prev_pose(X, edge, A);
        double cP[M], sP[M];                         // P_j = R_{j-1} Rz_j
        // ---- phase A: forces g, hand-back m -> b ----
        double pbx, pby, pbth;                        // b of this wave's predecessor pose (uniform)
        {
            double gx[M], gy[M], gth[M], mx[M], my[M], mth[M];
#pragma unroll
            for (int s = 0; s < M; ++s) {
                gx[s] = gy[s] = gth[s] = mx[s] = my[s] = mth[s] = 0.0;
                cP[s] = 1.0; sP[s] = 0.0;
                if (!valid[s]) continue;
                const double cz = ldc(F_CZ, s), sz = ldc(F_SZ, s);
                cP[s] = A[s].c * cz - A[s].s * sz;
                sP[s] = A[s].s * cz + A[s].c * sz;
                double qx, qy, qth;
                ldsym(F_OM, s).mul(ex[s], ey[s], eth[s], qx, qy, qth);
                gx[s] = cP[s] * qx - sP[s] * qy;
                gy[s] = sP[s] * qx + cP[s] * qy;
                gth[s] = qth;
                const double dx = X[s].x - A[s].x, dy = X[s].y - A[s].y;
                mx[s] = gx[s]; my[s] = gy[s];
                mth[s] = gth[s] + (-dy * gx[s] + dx * gy[s]);
            }
            Se2Scratch<W, NL>& S = sh.scr[phase & 1];
            if (tid < NL) loop_force(tid);            // read by the end-point owners after the barrier
            if (lane == 0) { S.lo_vec[wave][0] = mx[0]; S.lo_vec[wave][1] = my[0]; S.lo_vec[wave][2] = mth[0]; }
            if (lane == 63) { S.hi_vec[wave][0] = gx[M - 1]; S.hi_vec[wave][1] = gy[M - 1]; S.hi_vec[wave][2] = gth[M - 1]; }
            __syncthreads();
            // m of pose j+1: next lane / next slot's lane 0 / next wave's lane 0
            double n0 = 0.0, n1 = 0.0, n2 = 0.0;
            if (wave + 1 < W) { n0 = S.lo_vec[wave + 1][0]; n1 = S.lo_vec[wave + 1][1]; n2 = S.lo_vec[wave + 1][2]; }
            // b of the predecessor pose jp = jbase0 - 1 (zero for the gauge): m(jp+1) - g(jp) + loops
            pbx = pby = pbth = 0.0;
            if (wave > 0) {
                pbx = read_lane(mx[0], 0) - S.hi_vec[wave - 1][0];
                pby = read_lane(my[0], 0) - S.hi_vec[wave - 1][1];
                pbth = read_lane(mth[0], 0) - S.hi_vec[wave - 1][2];
                const int jp = wave * 64 * M;
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    const LoopState& st = sh.ls[cur][l];
                    if (jp == lt[l]) { pbx -= st.g[0]; pby -= st.g[1]; pbth -= st.g[2]; }
                    if (jp == lf[l]) {
                        const double dx = st.pt[0] - st.pf[0], dy = st.pt[1] - st.pf[1];
                        pbx += st.g[0]; pby += st.g[1];
                        pbth += st.g[2] + (-dy * st.g[0] + dx * st.g[1]);
                    }
                }
            }
            ++phase;
#pragma unroll
            for (int s = M - 1; s >= 0; --s) {
                const double ux = lane_next(mx[s], n0), uy = lane_next(my[s], n1), uth = lane_next(mth[s], n2);
                if (s > 0) { n0 = read_lane(mx[s], 0); n1 = read_lane(my[s], 0); n2 = read_lane(mth[s], 0); }
                // invalid slots hold m = 0, so poses past the end contribute nothing
                bx[s] = ux - gx[s]; by[s] = uy - gy[s]; bth[s] = uth - gth[s];
                if (!valid[s]) { bx[s] = by[s] = bth[s] = 0.0; }
            }
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const LoopState& st = sh.ls[cur][l];
#pragma unroll
                for (int s = 0; s < M; ++s) {
                    const int j = jbase + s * 64;
                    if (j == lt[l]) { bx[s] -= st.g[0]; by[s] -= st.g[1]; bth[s] -= st.g[2]; }
                    if (j == lf[l]) {
                        const double dx = st.pt[0] - st.pf[0], dy = st.pt[1] - st.pf[1];
                        bx[s] += st.g[0]; by[s] += st.g[1];
                        bth[s] += st.g[2] + (-dy * st.g[0] + dx * st.g[1]);
                    }
                }
            }
        }
        IPC_TICK(tmA)
        // ---- phase B: b^T b, b^T H b, capacitance partials; solve on wave 0 ----
        // G_{l,j} factors as Gamma_l * Phi_j with
        //   Phi_j   = [[P_j, -kappa_j],[0,1]],  kappa_j = J (t_j - o)        (per edge, loop-free)
        //   Gamma_l = sigma_l [[Lam_l, Lam_l K_l],[0,1]],  Lam_l = Rzl^T R_f^T,  K_l = J (t_to - o)
        // (o = gauge position), so only Psi_j = Phi_j Cov_j Phi_j^T (6 values) and w_j = Phi_j e_j
        // (3 values) are accumulated per loop range: S_ll' = Gamma_l M_ll' Gamma_l'^T with
        // M_ll' = sum over range_l & range_l' of Psi_j, and d_l = e_l - Gamma_l W_l.
        double bb, bHb, alpha, hsdNorm;
        double nu[3], nu2[3];                         // Gamma_l^T mu_l  (loop 1, loop 2)
        int lo1, hi1, lo2 = 0, hi2 = 0;
        lo1 = sh.lc[0].lo; hi1 = sh.lc[0].hi;
        if constexpr (NL == 2) { lo2 = sh.lc[1].lo; hi2 = sh.lc[1].hi; }
        {
            Se2Scratch<W, NL>& S = sh.scr[phase & 1];
            IPC_ESTART()
            publish_endpoint_vec(bx, by, bth, S);
            // group 1: b^T b, b^T H b, W_1 (3), M_11 (6)            -> red[wave][0..10]
            // group 2 (pair cells): W_2 (3), M_22 (6), M_12 (6)     -> red[wave][16..30]
            double v1[16], v2[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) { v1[k] = 0.0; v2[k] = 0.0; }
            double cbx = pbx, cby = pby, cbth = pbth;  // b of pose j-1 (lane 0 carry)
#pragma unroll
            for (int s = 0; s < M; ++s) {
                const double qbx = lane_prev(bx[s], cbx), qby = lane_prev(by[s], cby), qbth = lane_prev(bth[s], cbth);
                if (s + 1 < M) { cbx = read_lane(bx[s], 63); cby = read_lane(by[s], 63); cbth = read_lane(bth[s], 63); }
                if (!valid[s]) continue;
                v1[0] += bx[s] * bx[s] + by[s] * by[s] + bth[s] * bth[s];
                double wx, wy, wth;
                se2_apply_J(A[s], X[s], ldc(F_CZ, s), ldc(F_SZ, s), qbx, qby, qbth, bx[s], by[s], bth[s], wx, wy, wth);
                v1[1] += ldsym(F_OM, s).quad(wx, wy, wth);
                const Sym3 sg = ldsym(F_SG, s);
                const double c = cP[s], sn = sP[s];
                const double kx = -(X[s].y - gauge.y), ky = X[s].x - gauge.x;     // kappa_j
                // C = P Cov_tt P^T, c' = P (cov_02, cov_12)
                const double cc = c * c, ss = sn * sn, cs = c * sn;
                const double C00 = cc * sg.a00 - 2 * cs * sg.a01 + ss * sg.a11;
                const double C01 = cs * (sg.a00 - sg.a11) + (cc - ss) * sg.a01;
                const double C11 = ss * sg.a00 + 2 * cs * sg.a01 + cc * sg.a11;
                const double c0 = c * sg.a02 - sn * sg.a12, c1 = sn * sg.a02 + c * sg.a12;
                const double sth = sg.a22;
                double psi[6];                       // 00 01 02 11 12 22
                psi[2] = c0 - sth * kx;
                psi[4] = c1 - sth * ky;
                psi[0] = C00 - kx * c0 - kx * psi[2];          // C00 - 2 kx c0 + s kx^2
                psi[1] = C01 - kx * c1 - ky * psi[2];          // C01 - kx c1 - ky c0 + s kx ky
                psi[3] = C11 - ky * c1 - ky * psi[4];
                psi[5] = sth;
                const double w0 = c * ex[s] - sn * ey[s] - kx * eth[s];
                const double w1 = sn * ex[s] + c * ey[s] - ky * eth[s];
                const double w2 = eth[s];
                const int j = jbase + s * 64;
                const double m1 = (j > lo1 && j <= hi1) ? 1.0 : 0.0;
                v1[2] += m1 * w0; v1[3] += m1 * w1; v1[4] += m1 * w2;
#pragma unroll
                for (int k = 0; k < 6; ++k) v1[5 + k] += m1 * psi[k];
                if constexpr (NL == 2) {
                    const double m2 = (j > lo2 && j <= hi2) ? 1.0 : 0.0;
                    const double m12 = m1 * m2;
                    v2[0] += m2 * w0; v2[1] += m2 * w1; v2[2] += m2 * w2;
#pragma unroll
                    for (int k = 0; k < 6; ++k) { v2[3 + k] += m2 * psi[k]; v2[9 + k] += m12 * psi[k]; }
                }
            }
            IPC_ETICK(6)
            // Gamma_l only depends on the loop state: the last wave (usually the least loaded one)
            // prepares it for wave 0 ahead of the barrier.  Lane l*9+i*3+a : Gamma_l[i][a].
            if (wave == W - 1 && lane < NL * 9) {
                const int l = lane / 9, i = (lane % 9) / 3, a = lane % 3;
                const LoopConst& q = sh.lc[l];
                const LoopState& st = sh.ls[cur][l];
                const double Aq = st.pf[3] * q.cz - st.pf[4] * q.sz, Bq = st.pf[4] * q.cz + st.pf[3] * q.sz;
                const double Kx = -(st.pt[1] - gauge.y), Ky = st.pt[0] - gauge.x;
                double g;
                if (i == 0) g = a == 0 ? Aq : (a == 1 ? Bq : Aq * Kx + Bq * Ky);
                else if (i == 1) g = a == 0 ? -Bq : (a == 1 ? Aq : -Bq * Kx + Aq * Ky);
                else g = a == 2 ? 1.0 : 0.0;
                sh.wtmp[0][32 + lane] = q.sigma * g;
            }
            wave_sum16_store(v1, &S.red[wave * 32]);
            if constexpr (NL == 2) wave_sum16_store(v2, &S.red[wave * 32 + 16]);
            IPC_ETICK(7)
            __syncthreads();
            IPC_ETICK(8)
            ++phase;
            Se2Scratch<W, NL>& S2 = sh.scr[phase & 1];
            if (wave == 0) {
            // ---- capacitance solve, lane-parallel on wave 0:
            //   lane t < 32      : total of partial t over the waves          -> wtmp[t]
            //   lane l*9+i*3+a   : Gamma_l[i][a]                              -> wtmp[32 + .]
            //   lane r*(NS+1)+c  : S[r][c] (c < NS) / rhs d[r] (c == NS), then Gauss-Jordan in place
            double* wt = sh.wtmp[0];
            if (lane < 32) {
                double acc = 0.0;
#pragma unroll
                for (int w = 0; w < W; ++w) acc += S.red[w * 32 + lane];
                wt[lane] = acc;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            constexpr int RS = NS + 1;                    // row stride of the augmented system
            double val = 0.0;
            const int r = lane / RS, c = lane % RS;
            if (lane < NS * RS) {
                const int l1 = r / 3, i = r % 3;
                const double g10 = wt[32 + l1 * 9 + i * 3], g11 = wt[32 + l1 * 9 + i * 3 + 1], g12 = wt[32 + l1 * 9 + i * 3 + 2];
                if (c < NS) {
                    const int l2 = c / 3, k = c % 3;
                    const int mb = l1 == l2 ? (l1 == 0 ? 5 : 19) : 25;
                    const double m00 = wt[mb], m01 = wt[mb + 1], m02 = wt[mb + 2], m11 = wt[mb + 3], m12 = wt[mb + 4], m22 = wt[mb + 5];
                    const double t0 = g10 * m00 + g11 * m01 + g12 * m02;
                    const double t1 = g10 * m01 + g11 * m11 + g12 * m12;
                    const double t2 = g10 * m02 + g11 * m12 + g12 * m22;
                    val = t0 * wt[32 + l2 * 9 + k * 3] + t1 * wt[32 + l2 * 9 + k * 3 + 1] + t2 * wt[32 + l2 * 9 + k * 3 + 2];
                    if (l1 == l2) {
                        const int lo_ = i < k ? i : k, hi_ = i < k ? k : i;
                        val += sh.lc[l1].sg[lo_ * 3 - lo_ * (lo_ - 1) / 2 + (hi_ - lo_)];
                    }
                } else {
                    const int wb = l1 == 0 ? 2 : 16;
                    val = sh.ls[cur][l1].e[i] - (g10 * wt[wb] + g11 * wt[wb + 1] + g12 * wt[wb + 2]);
                }
            }
            bool okS = true;
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const double piv = read_lane(val, k * RS + k);
                okS = okS && (piv > 0);
                // reciprocal by v_rcp_f64 + two Newton steps (~1 ulp) instead of the ~30-op IEEE divide:
                // it sits on the serial critical path of the elimination
                double inv = __builtin_amdgcn_rcp(piv);
                inv = fma(fma(-piv, inv, 1.0), inv, inv);
                inv = fma(fma(-piv, inv, 1.0), inv, inv);
                const double rowk = __shfl(val, k * RS + c, 64);
                const double colk = __shfl(val, r * RS + k, 64);
                val = (r == k) ? rowk * inv : fma(-(colk * rowk), inv, val);
            }
            // nu_l[cc] = sum_rr Gamma_l[rr][cc] mu_{3l+rr}; mu_r sits in lane r*RS + NS
            {
                const int l = (lane < NS) ? lane / 3 : 0, cc = lane % 3;
                double nv = 0.0;
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) {
                    const double mu_r = __shfl(val, (3 * l + rr) * RS + NS, 64);
                    nv += wt[32 + l * 9 + rr * 3 + cc] * mu_r;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) { nu[k] = read_lane(nv, k); nu2[k] = NL == 2 ? read_lane(nv, 3 + k) : 0.0; }
            }
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { S2.sol[k] = nu[k]; if (NL == 2) S2.sol[3 + k] = nu2[k]; }
                S2.sol[NS] = wt[0];
                S2.sol[NS + 1] = wt[1];
                S2.sol[NS + 2] = okS ? 1.0 : 0.0;
            }
            }
            // the loops' share of b^T H b, in parallel on another wave
            if (wave == (W > 1 ? 1 : 0) && lane < NL) S2.sol[NS + 3 + lane] = loop_quad(lane, S);
            IPC_ETICK(9)
            __syncthreads();
            IPC_ETICK(10)
            ++phase;
#pragma unroll
            for (int k = 0; k < 3; ++k) { nu[k] = S2.sol[k]; nu2[k] = NL == 2 ? S2.sol[3 + k] : 0.0; }
            bb = S2.sol[NS];
            bHb = S2.sol[NS + 1] + S2.sol[NS + 3] + (NL == 2 ? S2.sol[NS + 4] : 0.0);
            if (S2.sol[NS + 2] == 0.0) { flags |= 2; break; }
            alpha = bb / bHb;
            hsdNorm = sqrt(alpha * alpha * bb);
        }

        IPC_TICK(tmB)
        // ---- phase C: u, rho, prefix sums -> h_gn; b^T h, h^T H h ----
        double hgnNorm, bh, hHh;
        {
            double rx_[M], ry_[M], rth_[M];
#pragma unroll
            for (int s = 0; s < M; ++s) {
                rx_[s] = ry_[s] = rth_[s] = 0.0;
                if (!valid[s]) continue;
                // sum_l G^T mu = Phi_j^T (sum_l mask_l nu_l),  Phi^T v = (P^T v_t, -kappa . v_t + v_th)
                const int j = jbase + s * 64;
                const double m1 = (j > lo1 && j <= hi1) ? 1.0 : 0.0;
                double n0 = m1 * nu[0], n1 = m1 * nu[1], n2 = m1 * nu[2];
                if constexpr (NL == 2) {
                    const double m2 = (j > lo2 && j <= hi2) ? 1.0 : 0.0;
                    n0 += m2 * nu2[0]; n1 += m2 * nu2[1]; n2 += m2 * nu2[2];
                }
                const double c = cP[s], sn = sP[s];
                const double kx = -(X[s].y - gauge.y), ky = X[s].x - gauge.x;
                const double wx = c * n0 + sn * n1, wy = -sn * n0 + c * n1, wth = -(kx * n0 + ky * n1) + n2;
                double vx, vy, vth;
                ldsym(F_SG, s).mul(wx, wy, wth, vx, vy, vth);
                const double ux = -vx - ex[s], uy = -vy - ey[s], uth = -vth - eth[s];
                rx_[s] = cP[s] * ux - sP[s] * uy;
                ry_[s] = sP[s] * ux + cP[s] * uy;
                rth_[s] = uth;
            }
            // wave-local inclusive scans in pose order (slot after slot), theta first
            double lth[M], carry = 0.0;
#pragma unroll
            for (int s = 0; s < M; ++s) {
                lth[s] = wave_inclusive_scan(rth_[s]) + carry;
                carry = read_lane(lth[s], 63);
            }
            const double thTot = carry;
            // term = rho_t + J dt * h_theta(j-1), with the wave-local part of h_theta
            double lx_[M], ly_[M], cx = 0.0, cy = 0.0, cprev = 0.0;
#pragma unroll
            for (int s = 0; s < M; ++s) {
                const double thPrev = lane_prev(lth[s], cprev);      // wave-local h_theta(j-1)
                cprev = read_lane(lth[s], 63);
                double tx = 0.0, ty = 0.0;
                if (valid[s]) {
                    const double dx = X[s].x - A[s].x, dy = X[s].y - A[s].y;
                    tx = rx_[s] - dy * thPrev;
                    ty = ry_[s] + dx * thPrev;
                }
                lx_[s] = wave_inclusive_scan(tx) + cx;
                ly_[s] = wave_inclusive_scan(ty) + cy;
                cx = read_lane(lx_[s], 63); cy = read_lane(ly_[s], 63);
            }
            Se2Scratch<W, NL>& S = sh.scr[phase & 1];
            // cross-wave correction: h_t(j) += base_t(w) + J (t_j - t_start(w)) * base_theta(w)
            const double sx0 = edge.x, sy0 = edge.y;
            double lastx = sx0, lasty = sy0;          // last valid pose of the wave
#pragma unroll
            for (int s = 0; s < M; ++s) {
                const unsigned long long vm = __ballot(valid[s]);
                if (vm) {
                    const int ll = 63 - __clzll((long long)vm);
                    lastx = read_lane(X[s].x, ll); lasty = read_lane(X[s].y, ll);
                }
            }
            if (lane == 0) {
                S.scan[wave][0] = thTot; S.scan[wave][1] = cx; S.scan[wave][2] = cy;
                S.scan[wave][3] = lastx - sx0; S.scan[wave][4] = lasty - sy0;
            }
            __syncthreads();
            double bth_ = 0.0, bxx = 0.0, byy = 0.0;       // bases of this wave = h of its predecessor
#pragma unroll
            for (int w = 0; w < W; ++w) {
                if (w < wave) {
                    bxx += S.scan[w][1] - S.scan[w][4] * bth_;
                    byy += S.scan[w][2] + S.scan[w][3] * bth_;
                    bth_ += S.scan[w][0];
                }
            }
            bth_ = uni(bth_); bxx = uni(bxx); byy = uni(byy);
            ++phase;
            // |h|^2 and b.h; h^T H h equals b^T h because h solves H h = b (the same identity that
            // gives b^T H h = b^T b), so the second quadratic form is not evaluated.
            double p0 = 0.0, p1 = 0.0;
#pragma unroll
            for (int s = 0; s < M; ++s) {
                if (valid[s]) {
                    hth[s] = lth[s] + bth_;
                    hx[s] = lx_[s] + bxx - (X[s].y - sy0) * bth_;
                    hy[s] = ly_[s] + byy + (X[s].x - sx0) * bth_;
                    p0 += hx[s] * hx[s] + hy[s] * hy[s] + hth[s] * hth[s];
                    p1 += bx[s] * hx[s] + by[s] * hy[s] + bth[s] * hth[s];
                } else { hx[s] = hy[s] = hth[s] = 0.0; }
            }
            Se2Scratch<W, NL>& S2 = sh.scr[phase & 1];
            p0 = wave_sum(p0); p1 = wave_sum(p1);
            if (lane == 0) { S2.red[wave] = p0; S2.red[16 + wave] = p1; }
            __syncthreads();
            double tot[2];
            gather_totals<2>(S2.red, W, tot);
            ++phase;
            hgnNorm = sqrt(tot[0]);
            bh = tot[1];
            hHh = bh;
        }

        IPC_TICK(tmC)
        // converged (Se2View::term_eps): in the Newton regime (the last iteration took the full Gauss-Newton
        // step at its first trial) and one more such step cannot move any edge's chi2 by more than
        // 2 sqrt(term_eps) relative; g2o would still run its trial loop to Terminate
        if (lastGN && hgnNorm < delta && fabs(bh) < term_scale * currentChi) { it_done = it + 1; tries_total += maxTrials; flags |= 1; break; }
        // ---- trial loop ----
        const double deltaAtEntry = delta;
        bool goodStep = false;
        int numTries = 0;
        bool haveBlend = false;
        double blendC = 0.0, blendBma = 0.0;          // hsd.(hgn-hsd), |hgn-hsd|^2: independent of delta
        auto blend_sums = [&]() {
            double p0 = 0.0, p1 = 0.0;
#pragma unroll
            for (int s = 0; s < M; ++s) {
                if (!valid[s]) continue;
                const double sx = alpha * bx[s], sy = alpha * by[s], sth = alpha * bth[s];
                const double ax = hx[s] - sx, ay = hy[s] - sy, ath = hth[s] - sth;
                p0 += sx * ax + sy * ay + sth * ath;
                p1 += ax * ax + ay * ay + ath * ath;
            }
            p0 = wave_sum(p0); p1 = wave_sum(p1);
            Se2Scratch<W, NL>& S = sh.scr[phase & 1];
            if (lane == 0) { S.red[wave] = p0; S.red[16 + wave] = p1; }
            __syncthreads();
            double tot[2];
            gather_totals<2>(S.red, W, tot);
            ++phase;
            blendC = tot[0]; blendBma = tot[1];
            haveBlend = true;
        };
        do {
            ++numTries;
            int stepType;                             // 0 GN, 1 SD, 2 DL
            double beta = 0.0, sdScale = 0.0;
            if (hgnNorm < delta) stepType = 0;
            else if (hsdNorm > delta) { stepType = 1; sdScale = delta / hsdNorm; }
            else {
                stepType = 2;
                if (!haveBlend) blend_sums();         // c = hsd.(hgn-hsd), |hgn-hsd|^2, once per iteration
                const double c = blendC, bma = blendBma, hsdSq = alpha * alpha * bb;
                if (c <= 0.) beta = (-c + sqrt(c * c + bma * (delta * delta - hsdSq))) / bma;
                else beta = (delta * delta - hsdSq) / (c + sqrt(c * c + bma * (delta * delta - hsdSq)));
            }
            // trial step h_dl = pcoef * b + qcoef * h_gn, and its linear gain from the
            // per-iteration scalars
            double pcoef, qcoef, hdlNorm;
            if (stepType == 0) { pcoef = 0.0; qcoef = 1.0; hdlNorm = hgnNorm; }
            else if (stepType == 1) { pcoef = sdScale * alpha; qcoef = 0.0; hdlNorm = delta; }
            else { pcoef = alpha - beta * alpha; qcoef = beta; hdlNorm = delta; }
            const double hdlHhdl = pcoef * pcoef * bHb + 2 * pcoef * qcoef * bb + qcoef * qcoef * hHh;
            const double bhdl = pcoef * bb + qcoef * bh;
            double linearGain = -1 * hdlHhdl + 2 * bhdl;
            // h_dl = pcoef * b + qcoef * h_gn for every step type
            bool changed = false, big = false;
#pragma unroll
            for (int s = 0; s < M; ++s) {
                Xn[s] = X[s];
                if (!valid[s]) continue;
                Xn[s].x = X[s].x + fma(pcoef, bx[s], qcoef * hx[s]);
                Xn[s].y = X[s].y + fma(pcoef, by[s], qcoef * hy[s]);
                Xn[s].th = wrap_pi(X[s].th + fma(pcoef, bth[s], qcoef * hth[s]));
                big |= fabs(Xn[s].th - X[s].th) >= 0.015625;
            }
            // cos/sin of the trial angles: rotate the committed (c, s) by the (tiny) effective change
            // unless some lane of the wave moved by >= 2^-6 rad (then the full kernel, whole wave)
            if (__ballot(big) != 0ull) {
#pragma unroll
                for (int s = 0; s < M; ++s)
                    if (valid[s]) sincos_pi(Xn[s].th, Xn[s].s, Xn[s].c);
            } else {
#pragma unroll
                for (int s = 0; s < M; ++s)
                    if (valid[s]) rotate_small(X[s].c, X[s].s, Xn[s].th - X[s].th, Xn[s].c, Xn[s].s);
            }
            if (stepType == 1) {                      // only the steepest-descent no-op shortcut needs it
#pragma unroll
                for (int s = 0; s < M; ++s)
                    changed |= valid[s] && ((Xn[s].x != X[s].x) || (Xn[s].y != X[s].y) || (Xn[s].th != X[s].th));
            } else changed = true;
            const int trial = cur ^ 1;
            bool anyChanged;
            const double newChi = evaluate(Xn, std::integral_constant<int, 0>{}, trial, changed, anyChanged);
            ++evals;
            const double nonLinearGain = currentChi - newChi;
            if (fabs(linearGain) < 1e-12) linearGain = 1e-12;
            // rho = nonLinearGain / linearGain is only ever compared with 0, 0.25 and 0.75, so the
            // comparisons are done without the FP64 division (NaN still fails every test)
            const bool linPos = linearGain > 0;
            auto rho_gt = [&](double t) { return linPos ? nonLinearGain > t * linearGain : nonLinearGain < t * linearGain; };
            auto rho_lt = [&](double t) { return linPos ? nonLinearGain < t * linearGain : nonLinearGain > t * linearGain; };
            if (rho_gt(0.0)) {                        // rho > 0, discardTop: commit the trial
                goodStep = true;
                currentChi = newChi;
                cur = trial;
                edge = edgeN;
#pragma unroll
                for (int s = 0; s < M; ++s) {
                    X[s] = Xn[s];
                    // (hidden from CSE: otherwise the trial pass keeps half of its error arithmetic alive)
                    asm volatile("" : "+v"(X[s].x), "+v"(X[s].y), "+v"(X[s].th));
                }
                {   // errors of the new committed state (same arithmetic as the trial pass, no barrier)
                    Pose2 An[M];
                    prev_pose(X, edge, An);
#pragma unroll
                    for (int s = 0; s < M; ++s) {
                        if (!valid[s]) continue;
                        se2_error(An[s], X[s], ldc(F_TZX, s), ldc(F_TZY, s), ldc(F_CZ, s), ldc(F_SZ, s), ldc(F_THZ, s),
                                  ex[s], ey[s], eth[s]);
                    }
                }
            }
            if (rho_gt(0.75)) delta = fmax(delta, 3 * hdlNorm);
            else if (rho_lt(0.25)) delta *= 0.5;
            if (!goodStep) {
                if (nonLinearGain != nonLinearGain) {
                    numTries = maxTrials;       // NaN gain ratio: g2o leaves delta alone, so every retry is this same trial
                } else if (stepType == 0) {
                    // identical GN trial repeats while hgnNorm < delta: each halves delta
                    while (numTries < maxTrials && hgnNorm < delta) { ++numTries; delta *= 0.5; }
                } else if (stepType == 1 && !anyChanged) {
                    numTries = maxTrials;             // every later (halved) SD step is a no-op too
                }
            }
        } while (!goodStep && numTries < maxTrials);
        lastGN = goodStep && numTries == 1 && hgnNorm < deltaAtEntry;
        it_done = it + 1;
        tries_total += numTries;
        if (numTries == maxTrials || !goodStep) { flags |= 1; break; }
    }

#ifdef IPC_PHASE_TIMING
    IPC_TICK(tmT)
    if (tid == 0 && P.dbg) {
        atomicAdd(reinterpret_cast<unsigned long long*>(P.dbg) + 0, tmA);
        atomicAdd(reinterpret_cast<unsigned long long*>(P.dbg) + 1, tmB);
        atomicAdd(reinterpret_cast<unsigned long long*>(P.dbg) + 2, tmC);
        atomicAdd(reinterpret_cast<unsigned long long*>(P.dbg) + 3, tmT);
        atomicAdd(reinterpret_cast<unsigned long long*>(P.dbg) + 4, (unsigned long long)it_done);
        atomicAdd(reinterpret_cast<unsigned long long*>(P.dbg) + 5, (unsigned long long)evals);
        for (int k = 0; k < 6; ++k) atomicAdd(reinterpret_cast<unsigned long long*>(P.dbg) + 8 + k, te[k]);
        for (int k = 6; k < 11; ++k) atomicAdd(reinterpret_cast<unsigned long long*>(P.dbg) + 18 + k, te[k]);
    }
    if (tid == 64 * (W - 1) && P.dbg)
        for (int k = 0; k < 6; ++k) atomicAdd(reinterpret_cast<unsigned long long*>(P.dbg) + 16 + k, te[k]);
#endif
    // ---- per-edge chi2 (consensus_utils.cpp:15-19) ----
    double mx = 0.0;
    bool nan = false;
#pragma unroll
    for (int s = 0; s < M; ++s) {
        if (!valid[s]) continue;
        const double c = ldsym(F_OM, s).quad(ex[s], ey[s], eth[s]);
        if (c != c) nan = true;
        else mx = fmax(mx, c);
    }
    mx = wave_max(mx);
    {
        Se2Scratch<W, NL>& S = sh.scr[phase & 1];
        const unsigned long long nb = __ballot(nan);
        if (lane == 0) { S.red[wave] = mx; S.red[16 + wave] = nb ? 1.0 : 0.0; }
        __syncthreads();
        double m2 = S.red[0], nn = S.red[16];
#pragma unroll
        for (int w = 1; w < W; ++w) { m2 = fmax(m2, S.red[w]); nn += S.red[16 + w]; }
        mx = m2;
        nan = nn != 0.0;
        ++phase;
    }
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const double c = sh.ls[cur][l].chi;
        if (c != c) nan = true;
        else mx = fmax(mx, c);
    }
    // consensus_utils.cpp:17-19 rejects as soon as ONE edge has chi2 > th; a NaN chi2 is not "> th".  So the maximum is
    // taken over the edges that have a number, and NaN is reported only when none of them is positive (agrees either way).
    if (nan && !(mx > 0.0)) mx = __longlong_as_double(0x7ff8000000000000ll);
    res.max_chi2 = mx;
    res.chi2_total = currentChi;
    res.iterations = it_done;
    res.tries = tries_total;
    res.flags = flags;
    res.evals = evals;
}

}  // namespace ipc
