// SE(3) cell solver with the chain poses in LDS -- teams of W = 1 or 4 waves per cell (gfx950).
//
// Mathematics, dog-leg control flow and shortcuts are those of se3_cell.hpp (reference
// src/consensus_utils.cpp:7-22 on a gauge-fixed chain + one or two loop closures, g2o dog-leg,
// EdgeSE3 / VertexSE3 conventions, chain closed form + capacitance matrix).  What changed is where
// the state lives and how a cell is walked:
//   * the committed poses sit in LDS as (unit quaternion, translation) -- 56 bytes per pose, so a
//     workgroup holds a chain of 2560 poses (sphere2500 with every loop closure) in 143 KB.  A
//     lane reads its own pose and its chain neighbour (index - 1, another lane's pose) with
//     conflict-free ds_read_b128 / b64; nothing is handed over through DPP or scratch;
//   * per pose only b (6) and the Gauss-Newton step h (6) stay in registers: 24 VGPRs per pose,
//     M <= 10 poses per lane inside the 512 registers of a one-wave-per-SIMD kernel.  Trial
//     poses and odometry errors are never stored: a trial is one sweep that steps, evaluates
//     and sums chi2 on the fly, an accepted trial is re-swept once to commit (se2_wave_cell.hpp);
//   * consecutive lanes hold consecutive poses (pose = wave * 64 M + slot * 64 + lane + 1), so
//     the chain constants (54 doubles per edge, L2 resident) are read field-major with fully
//     coalesced 512-byte wave loads;
//   * the 27 capacitance partials of a pose (Psi_j, w_j) are accumulated UNMASKED in 27 registers
//     and reduced into per-class totals (class = set of loop ranges the pose lies in) whenever the
//     class changes along the wave's run of poses -- at most twice per sweep -- instead of 77
//     masked accumulators;
//   * persistent workgroups of four waves (one per SIMD) take cells from a work queue: four
//     independent cells (W = 1, no barrier at all) or one cell on the whole workgroup (W = 4,
//     s_barrier between phases).
// Rotations are composed as quaternions (q <- q * dq) where se3_cell.hpp multiplies matrices;
// both are the same update X <- X * fromVectorMQT(delta) up to rounding.
#pragma once
#include "se3_cell.hpp"

namespace ipc {

__device__ __forceinline__ void wave_sync3()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

struct PoseQ { double q[4]; double t[3]; };      // q = (x, y, z, w)

__device__ __forceinline__ void rot_of(const PoseQ& p, Pose3& X)
{
    R_from_quat(p.q[3], p.q[0], p.q[1], p.q[2], X.R);
    X.t[0] = p.t[0]; X.t[1] = p.t[1]; X.t[2] = p.t[2];
}

// LDS of one team
template <int W, int M, int NL>
struct Se3Lds {
    static constexpr int CAP = 64 * W * M + 1;    // pose index 0 is the gauge
    double2 q01[CAP];                             // (qx, qy)
    double2 q23[CAP];                             // (qz, qw)
    double t0[CAP], t1[CAP], t2[CAP];
    LoopConst3 lc[NL];
    LoopState3 ls[2][NL];
    double red[2][W][32];
    double tot[W][4][32];     // class totals of the capacitance partials: [wave][class][value]
    double hi_pose[2][W][8];  // trial sweep: (q, t) of each wave's last pose
    double hi_vec[2][W][6];
    double lo_vec[2][W][6];
    double lvec[2][NL][2][6];
    double scan[2][W][9];
    double sol[2][NL * 6 + 3];
    double w0tot[80];
    double w0gam[NL][36];
    double w0aug[NL * 6][NL * 6 + 1];
    double w0mu[NL * 6];
    double w0lq[2];           // b^T H b terms of the loop edges
    int cell;                 // the team's current cell (W > 1: broadcast from wave 0)
    int pad_;
};

// dst[k] += sum over the wave of v[k], k < 16 (one lane per value; dst is wave-private LDS)
__device__ __forceinline__ void wave_sum16_add(const double (&v)[16], double* dst)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double r0 = pair32(v[i], v[i + 8]);
        const double r1 = pair32(v[i + 4], v[i + 12]);
        double q = pair16(r0, r1);
        q = row_inclusive_scan(q);
        if ((lane & 15) == 15) dst[i + 4 * (lane >> 4)] += q;
    }
}

#define IPC3_FENCE() __builtin_amdgcn_sched_barrier(0)

template <int W, int M, int NL>
__device__ __forceinline__ void se3_lds_solve(const Se3View& P, int lo_abs, int L, const int (&cand)[2], int iterations,
                                              Se3Lds<W, M, NL>& sh, CellResult3& res)
{
    static_assert(W == 1 || W == 4, "one wave per cell, or the whole workgroup of four");
    constexpr int NS = NL * 6;
    const double term_scale = P.term_eps / (double)(L + NL);   // 0: the test is off
    bool lastGN = false;
    const int lane = threadIdx.x & 63;
    const int wave = W > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;  // wave of the team
    const int wbase = wave * 64 * M;                  // poses wbase+1 .. wbase+64M belong to this wave
    const int jbase = wbase + lane + 1;               // pose of slot 0

#ifdef IPC_PHASE_TIMING
    unsigned long long tmA = 0, tmB1 = 0, tmB2 = 0, tmC = 0, tmT = 0, tmK = 0, tmBar = 0, tm0 = __builtin_amdgcn_s_memtime();
#define IPC3_TICK(acc) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc += t_ - tm0; tm0 = t_; }
#define IPC3_SYNC() do { const unsigned long long b_ = __builtin_amdgcn_s_memtime(); __syncthreads(); tmBar += __builtin_amdgcn_s_memtime() - b_; } while (0)
#else
#define IPC3_TICK(acc)
#define IPC3_SYNC() __syncthreads()
#endif
    int phase = 0;
    auto tbar = [&]() {
        if constexpr (W > 1) IPC3_SYNC();
        else wave_sync3();
    };
    // team totals of K per-lane values, the same bits on every lane of the team
    auto team_sum2 = [&](double& a, double& b2) {
        a = wave_sum(a);
        b2 = wave_sum(b2);
        if constexpr (W > 1) {
            const int buf = phase & 1;
            if (lane == 0) { sh.red[buf][wave][0] = a; sh.red[buf][wave][1] = b2; }
            IPC3_SYNC();
            ++phase;
            double x = sh.red[buf][0][0], y = sh.red[buf][0][1];
#pragma unroll
            for (int w = 1; w < W; ++w) { x += sh.red[buf][w][0]; y += sh.red[buf][w][1]; }
            a = x; b2 = y;
        }
    };

    // ---------------- loop constants -> LDS ----------------
    if (wave == 0 && lane < NL) {
        const int l = lane, c = cand[l];
        LoopConst3& q = sh.lc[l];
        q.f = P.cand_from[c] - lo_abs;
        q.t = P.cand_to[c] - lo_abs;
        q.lo = min(q.f, q.t);
        q.hi = max(q.f, q.t);
        q.sigma = q.t > q.f ? 1.0 : -1.0;
        for (int k = 0; k < 9; ++k) q.Rz[k] = P.cand[(size_t)(G_RZ + k) * P.cstride + c];
        for (int k = 0; k < 3; ++k) q.tz[k] = P.cand[(size_t)(G_TZ + k) * P.cstride + c];
        for (int k = 0; k < 21; ++k) {
            q.om[k] = P.cand[(size_t)(G_OM + k) * P.cstride + c];
            q.sg[k] = P.cand[(size_t)(G_SG + k) * P.cstride + c];
        }
    }

    // ---------------- poses -> LDS (idle slots hold a copy of the gauge) ----------------
    Pose3 gauge;
#pragma unroll
    for (int k = 0; k < 9; ++k) gauge.R[k] = P.pose0[(size_t)k * P.V + lo_abs];
#pragma unroll
    for (int k = 0; k < 3; ++k) gauge.t[k] = P.pose0[(size_t)(9 + k) * P.V + lo_abs];
    auto st_pose = [&](int p, const PoseQ& y) {
        sh.q01[p] = make_double2(y.q[0], y.q[1]);
        sh.q23[p] = make_double2(y.q[2], y.q[3]);
        sh.t0[p] = y.t[0]; sh.t1[p] = y.t[1]; sh.t2[p] = y.t[2];
    };
    auto ld_pose = [&](int p) -> PoseQ {
        PoseQ y;
        const double2 a = sh.q01[p], c = sh.q23[p];
        y.q[0] = a.x; y.q[1] = a.y; y.q[2] = c.x; y.q[3] = c.y;
        y.t[0] = sh.t0[p]; y.t[1] = sh.t1[p]; y.t[2] = sh.t2[p];
        return y;
    };
    {
        PoseQ gq;
        quat_from_R(gauge.R, gq.q);
        gq.t[0] = gauge.t[0]; gq.t[1] = gauge.t[1]; gq.t[2] = gauge.t[2];
        if (wave == 0 && lane == 0) st_pose(0, gq);
#pragma unroll
        for (int s = 0; s < M; ++s) {
            const int j = jbase + s * 64;
            PoseQ y = gq;
            if (j <= L) {
                double R[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) R[k] = P.pose0[(size_t)k * P.V + lo_abs + j];
                quat_from_R(R, y.q);
#pragma unroll
                for (int k = 0; k < 3; ++k) y.t[k] = P.pose0[(size_t)(9 + k) * P.V + lo_abs + j];
            }
            st_pose(j, y);
        }
    }
    double b[M][6], h[M][6];
#pragma unroll
    for (int s = 0; s < M; ++s)
#pragma unroll
        for (int k = 0; k < 6; ++k) { b[s][k] = 0.0; h[s][k] = 0.0; }
    tbar();

    int lf[NL], lt[NL], rlo[NL], rhi[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        lf[l] = __builtin_amdgcn_readfirstlane(sh.lc[l].f);
        lt[l] = __builtin_amdgcn_readfirstlane(sh.lc[l].t);
        rlo[l] = min(lf[l], lt[l]);
        rhi[l] = max(lf[l], lt[l]);
    }
    if (wave == 0 && lane == 0) {                     // gauge end points never change
#pragma unroll
        for (int l = 0; l < NL; ++l)
#pragma unroll
            for (int bsel = 0; bsel < 2; ++bsel) {
                if (lf[l] == 0) sh.ls[bsel][l].pf = gauge;
                if (lt[l] == 0) sh.ls[bsel][l].pt = gauge;
            }
    }
    // slots of this wave that hold a loop end point (wave-uniform bit mask)
    unsigned ownSlots = 0;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        if (lf[l] > wbase && lf[l] <= wbase + 64 * M) ownSlots |= 1u << ((lf[l] - wbase - 1) >> 6);
        if (lt[l] > wbase && lt[l] <= wbase + 64 * M) ownSlots |= 1u << ((lt[l] - wbase - 1) >> 6);
    }
    // class boundaries along the chain (pair cells): poses <= bp[k] keep the class, later ones change
    int bp[2] = {1 << 30, 1 << 30};
    if constexpr (NL == 2) {
        int c4[4] = {rlo[0], rhi[0], rlo[1], rhi[1]};
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = a + 1; c < 4; ++c)
                if (c4[c] < c4[a]) { const int t = c4[a]; c4[a] = c4[c]; c4[c] = t; }
        bp[0] = __builtin_amdgcn_readfirstlane(c4[1]);    // c4[0] == 0 and c4[3] == L
        bp[1] = __builtin_amdgcn_readfirstlane(c4[2]);
    }
    auto class_of = [&](int j) -> int {
        int c = (j > rlo[0] && j <= rhi[0]) ? 1 : 0;
        if constexpr (NL == 2) c |= (j > rlo[1] && j <= rhi[1]) ? 2 : 0;
        return c;
    };

    // chain constants: blocked records (Se3View::chain_blk) -- block of 64 edges x 28 double2, so the
    // 64 lanes of a wave (consecutive edges) read 1 KB contiguous per global_load_dwordx4
    // The per-slot indices, validity and range predicates are loop invariant; left alone the compiler
    // hoists all of them out of the dog-leg loop (masks, 0/1 doubles, addresses for every slot) and then
    // spills them.  jv is jbase made opaque at every phase start, so they are recomputed where used.
    int jv = jbase;
    auto opaque = [&]() { asm volatile("" : "+v"(jv)); };
    auto rec_of = [&](int s) -> const double2* {
#ifdef IPC_DBG_SAMEREC   // timing experiment only (wrong results): every slot of every wave reads the same 64 records
        const unsigned e = (unsigned)(lo_abs + lane + 0 * (jv + s));
#else
        const unsigned e = (unsigned)(lo_abs + jv - 1 + s * 64);   // edge k joins pose k -> k+1
#endif
        return P.chain_blk + ((size_t)(e >> 6) * (kSe3BlkPairs * 64) + (e & 63u));
    };
    auto ld_rz = [&](const double2* pl, double* Rz, double* tz) {
        double v[12];
#pragma unroll
        for (int p = 0; p < 6; ++p) { const double2 t = pl[p * 64]; v[2 * p] = t.x; v[2 * p + 1] = t.y; }
#pragma unroll
        for (int k = 0; k < 9; ++k) Rz[k] = v[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) tz[k] = v[9 + k];
    };
    // p0 = 6: information, p0 = 17: covariance (21 values + one pad)
    auto ld_sym = [&](const double2* pl, int p0, double* S) {
#pragma unroll
        for (int p = 0; p < 11; ++p) {
            const double2 t = pl[(p0 + p) * 64];
            S[2 * p] = t.x;
            if (2 * p + 1 < 21) S[2 * p + 1] = t.y;
        }
    };

    auto loop_edge = [&](int l, int bsel, Edge3& E) {
        const LoopState3& st = sh.ls[bsel][l];
#pragma unroll
        for (int k = 0; k < 9; ++k) { E.Rab[k] = st.Rab[k]; E.RE[k] = st.RE[k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { E.tab[k] = st.tab[k]; E.qv[k] = st.qv[k]; }
        E.qw = st.qw;
#pragma unroll
        for (int k = 0; k < 6; ++k) E.e[k] = st.e[k];
    };
    // loop l at the end-point poses of buffer bsel; store: keep the state (one wave of the team does)
    auto loop_eval = [&](int l, int bsel, bool store) -> double {
        const LoopConst3& q = sh.lc[l];
        LoopState3& st = sh.ls[bsel][l];
        Pose3 a = st.pf, bb = st.pt;
        Edge3 E;
        se3_edge(a, bb, q.Rz, q.tz, E);
        double qo[6];
        sym6_mul(q.om, E.e, qo);
        double chi = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) chi += E.e[k] * qo[k];
        if (store) {
            double g[6], m[6];
            se3_Dt(E, qo, g);
            se3_Adt(E, g, m);
#pragma unroll
            for (int k = 0; k < 9; ++k) { st.Rab[k] = E.Rab[k]; st.RE[k] = E.RE[k]; }
#pragma unroll
            for (int k = 0; k < 3; ++k) { st.tab[k] = E.tab[k]; st.qv[k] = E.qv[k]; }
            st.qw = E.qw;
#pragma unroll
            for (int k = 0; k < 6; ++k) { st.e[k] = E.e[k]; st.g[k] = g[k]; st.m[k] = m[k]; }
            st.chi = chi;
        }
        return chi;
    };
    int cur = 0;
    auto loop_quad = [&](int l, int buf) -> double {
        Edge3 E;
        loop_edge(l, cur, E);
        double va[6], vb[6], w[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) { va[k] = sh.lvec[buf][l][0][k]; vb[k] = sh.lvec[buf][l][1][k]; }
        se3_apply_J(E, va, vb, w);
        return sym6_quad(sh.lc[l].om, w);
    };

    // X (+) delta: q <- q * (dq, sqrt(1 - |dq|^2)), t <- t + R dt   (VertexSE3::oplusImpl)
    auto oplus = [&](const PoseQ& x, const double* dl) -> PoseQ {
        PoseQ y;
        Pose3 Xr;
        rot_of(x, Xr);
        double rt3[3];
        m3_vec(Xr.R, dl, rt3);
#pragma unroll
        for (int k = 0; k < 3; ++k) y.t[k] = x.t[k] + rt3[k];
        double wq = 1.0 - (dl[3] * dl[3] + dl[4] * dl[4] + dl[5] * dl[5]);
        double dx = dl[3], dy = dl[4], dz = dl[5];
        if (wq < 0) { wq = 1.0; dx = 0.0; dy = 0.0; dz = 0.0; }
        else wq = sqrt(wq);
        const double ax = x.q[0], ay = x.q[1], az = x.q[2], aw = x.q[3];
        y.q[0] = aw * dx + ax * wq + ay * dz - az * dy;
        y.q[1] = aw * dy - ax * dz + ay * wq + az * dx;
        y.q[2] = aw * dz + ax * dy - ay * dx + az * wq;
        y.q[3] = aw * wq - ax * dx - ay * dy - az * dz;
        return y;
    };

    // One sweep over the chain.
    //   MODE 0: errors of the committed poses            -> chi2 (sum), loop state buffer `bsel`
    //   MODE 1: trial poses X (+) (p b + q h), not stored -> chi2 (sum), loop state buffer `bsel`, changed
    //   MODE 2: commit X <- X (+) (p b + q h)
    //   MODE 3: committed poses                          -> max per-edge chi2 (NaN flagged in sweepNan)
    bool sweepChanged = false, sweepNan = false;
    auto sweep = [&](auto mode_c, int stepType, double pc, double qc, int bsel) -> double {
        constexpr int MODE = decltype(mode_c)::value;
        opaque();
        asm volatile("" : "+v"(pc), "+v"(qc));
        auto stepped = [&](int s, bool v) -> PoseQ {
            const int j = jv + s * 64;
            const PoseQ x = ld_pose(j);
            if (!v) return x;
            double dl[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                if (stepType == 0) dl[k] = h[s][k];
                else dl[k] = fma(pc, b[s][k], qc * h[s][k]);
            }
            return oplus(x, dl);
        };
        if constexpr (MODE == 2) {
#pragma unroll
            for (int s = 0; s < M; ++s) {
                const int j = jv + s * 64;
                const bool v = j <= L;
                const PoseQ y = stepped(s, v);
                if (v) st_pose(j, y);
                IPC3_FENCE();
            }
            tbar();
            return 0.0;
        } else {
            PoseQ carry;                              // pose wbase (the wave's predecessor)
            PoseQ last;
            if constexpr (MODE == 1) {
                last = stepped(M - 1, jv + (M - 1) * 64 <= L);
                if constexpr (W > 1) {
                    const int buf = phase & 1;
                    if (lane == 63) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) sh.hi_pose[buf][wave][k] = last.q[k];
#pragma unroll
                        for (int k = 0; k < 3; ++k) sh.hi_pose[buf][wave][4 + k] = last.t[k];
                    }
                    IPC3_SYNC();
                    ++phase;
                    carry = ld_pose(0);
                    if (wave > 0) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) carry.q[k] = sh.hi_pose[buf][wave - 1][k];
#pragma unroll
                        for (int k = 0; k < 3; ++k) carry.t[k] = sh.hi_pose[buf][wave - 1][4 + k];
                    }
                } else carry = ld_pose(0);
            }
            double part = 0.0;
            bool changed = false, nan = false;
#pragma unroll
            for (int s = 0; s < M; ++s) {
                const int j = jv + s * 64;
                const bool v = j <= L;
                PoseQ y, a;
                if constexpr (MODE == 1) {
                    y = (s == M - 1) ? last : stepped(s, v);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { a.q[k] = lane_prev(y.q[k], carry.q[k]); carry.q[k] = read_lane(y.q[k], 63); }
#pragma unroll
                    for (int k = 0; k < 3; ++k) { a.t[k] = lane_prev(y.t[k], carry.t[k]); carry.t[k] = read_lane(y.t[k], 63); }
                    if (stepType == 1) {
                        const PoseQ x = ld_pose(j);
                        bool c = false;
#pragma unroll
                        for (int k = 0; k < 4; ++k) c |= y.q[k] != x.q[k];
#pragma unroll
                        for (int k = 0; k < 3; ++k) c |= y.t[k] != x.t[k];
                        changed |= v && c;
                    }
                } else {
                    y = ld_pose(j);
                    a = ld_pose(j - 1);
                }
                if (v) {
                    const double2* pl = rec_of(s);
                    Pose3 A, Y;
                    Edge3 E;
                    double om[21];
                    {
                        double Rz[9], tz[3];
                        ld_rz(pl, Rz, tz);
                        ld_sym(pl, 6, om);
                        rot_of(a, A);
                        rot_of(y, Y);
                        se3_edge(A, Y, Rz, tz, E);
                    }
                    IPC3_FENCE();
                    const double c2 = sym6_quad(om, E.e);
                    if constexpr (MODE == 3) {
                        if (c2 != c2) nan = true;
                        else part = fmax(part, c2);
                    } else part += c2;
                    if ((ownSlots >> s) & 1u) {
#pragma unroll
                        for (int l = 0; l < NL; ++l) {
                            if (j == lf[l]) sh.ls[bsel][l].pf = Y;
                            if (j == lt[l]) sh.ls[bsel][l].pt = Y;
                        }
                    }
                }
                asm volatile("" : "+v"(part) : : "memory");
                IPC3_FENCE();
            }
            if constexpr (MODE == 3) {
                part = wave_max(part);
                double nn = (__ballot(nan) != 0ull) ? 1.0 : 0.0;
                if constexpr (W > 1) {
                    const int buf = phase & 1;
                    if (lane == 0) { sh.red[buf][wave][0] = part; sh.red[buf][wave][1] = nn; }
                    IPC3_SYNC();
                    ++phase;
                    part = sh.red[buf][0][0]; nn = sh.red[buf][0][1];
#pragma unroll
                    for (int w = 1; w < W; ++w) { part = fmax(part, sh.red[buf][w][0]); nn += sh.red[buf][w][1]; }
                }
                sweepNan = nn != 0.0;
                return part;
            } else {
                // one exchange: partial sums (and that every end-point pose is in place); the loops
                // are then evaluated by every wave (same values), wave 0 keeps the state
                double chg = changed ? 1.0 : 0.0;
                wave_sync3();
                team_sum2(part, chg);
                const double lcv = lane < NL ? loop_eval(lane, bsel, wave == 0) : 0.0;
#pragma unroll
                for (int l = 0; l < NL; ++l) part += read_lane(lcv, l);
                sweepChanged = chg != 0.0;
                return part;
            }
        }
    };

    // ---------------- initial errors (consensus_utils.cpp:11) ----------------
    int evals = 1;
    double currentChi = sweep(std::integral_constant<int, 0>{}, 0, 0.0, 0.0, cur);
    IPC3_TICK(tmT)

    double delta = 1e4;
    const int maxTrials = 100;
    int it_done = 0, tries_total = 0, flags = 0;

    for (int it = 0; it < iterations; ++it) {
        // ---- phase A: g = D^T Om e, m = Ad^T g; b_j = m_{j+1} - g_j + loop terms ----
        opaque();
        tbar();                                       // loop state of `cur` visible to every wave
        double bbp = 0.0;
        {
            auto force = [&](int s, double* g, double* m) {
                const int j = jv + s * 64;
#pragma unroll
                for (int k = 0; k < 6; ++k) { g[k] = 0.0; m[k] = 0.0; }
                const PoseQ y = ld_pose(j), a = ld_pose(j - 1);
                if (j <= L) {
                    const double2* pl = rec_of(s);
                    Pose3 A, Y;
                    Edge3 E;
                    double om[21], qo[6];
                    {
                        double Rz[9], tz[3];
                        ld_rz(pl, Rz, tz);
                        ld_sym(pl, 6, om);
                        rot_of(a, A);
                        rot_of(y, Y);
                        se3_edge(A, Y, Rz, tz, E);
                    }
                    IPC3_FENCE();
                    sym6_mul(om, E.e, qo);
                    se3_Dt(E, qo, g);
                    se3_Adt(E, g, m);
                }
            };
            double g0[6], m0[6], nx[6];
            force(0, g0, m0);
#pragma unroll
            for (int k = 0; k < 6; ++k) nx[k] = 0.0;
            if constexpr (W > 1) {
                const int buf = phase & 1;
                if (lane == 0) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) sh.lo_vec[buf][wave][k] = m0[k];
                }
                IPC3_SYNC();
                ++phase;
                if (wave + 1 < W) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) nx[k] = sh.lo_vec[buf][wave + 1][k];
                }
            }
#pragma unroll
            for (int s = M - 1; s >= 0; --s) {
                double g[6], m[6];
                if (s == 0) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) { g[k] = g0[k]; m[k] = m0[k]; }
                } else force(s, g, m);
                const int j = jv + s * 64;
                const bool v = j <= L;
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const double u = lane_next(m[k], nx[k]);
                    nx[k] = read_lane(m[k], 0);
                    b[s][k] = v ? u - g[k] : 0.0;
                }
                if ((ownSlots >> s) & 1u) {
#pragma unroll
                    for (int l = 0; l < NL; ++l) {
                        const LoopState3& st = sh.ls[cur][l];
#pragma unroll
                        for (int k = 0; k < 6; ++k) {
                            if (j == lt[l]) b[s][k] -= st.g[k];
                            if (j == lf[l]) b[s][k] += st.m[k];
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 6; ++k) bbp += b[s][k] * b[s][k];
                asm volatile("" : "+v"(b[s][0]), "+v"(b[s][1]), "+v"(b[s][2]), "+v"(b[s][3]), "+v"(b[s][4]), "+v"(b[s][5]),
                             "+v"(bbp) : : "memory");
                IPC3_FENCE();
            }
        }
        IPC3_TICK(tmA)
        // ---- phase B: b^T b, b^T H b and the capacitance partials; solve on wave 0 ----
        // (formulas: se3_cell.hpp phase B)
        double bb, bHb, alpha, hsdNorm;
        double nu[NL][6];
        opaque();
        {
            const int bufB = phase & 1;
            // b at the loop end points and at the wave boundaries
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                if (wave == 0 && lane == 0) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        if (lf[l] == 0) sh.lvec[bufB][l][0][k] = 0.0;
                        if (lt[l] == 0) sh.lvec[bufB][l][1][k] = 0.0;
                    }
                }
#pragma unroll
                for (int s = 0; s < M; ++s) {
                    if (!((ownSlots >> s) & 1u)) continue;
                    const int j = jv + s * 64;
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        if (j == lf[l]) sh.lvec[bufB][l][0][k] = b[s][k];
                        if (j == lt[l]) sh.lvec[bufB][l][1][k] = b[s][k];
                    }
                }
            }
            if (lane == 63) {
#pragma unroll
                for (int k = 0; k < 6; ++k) sh.hi_vec[bufB][wave][k] = b[M - 1][k];
            }
            // zero this wave's class totals
#pragma unroll
            for (int q0 = 0; q0 < 128; q0 += 64) (&sh.tot[wave][0][0])[q0 + lane] = 0.0;
            tbar();
            ++phase;
            double cb[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) cb[k] = (W > 1 && wave > 0) ? sh.hi_vec[bufB][wave > 0 ? wave - 1 : 0][k] : 0.0;

            // value layout of a pose's partials: [0..6) w_j, [6..27) Psi_j, [27] b^T H b term, [28] b^T b term
            double acc[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) acc[k] = 0.0;
            // totals of class `cls` += wave sums of acc over the lanes with `take`; those lanes restart from zero
            auto flush = [&](int cls, bool take) {
                double v16[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) v16[k] = take ? acc[k] : 0.0;
                wave_sum16_add(v16, &sh.tot[wave][cls][0]);
#pragma unroll
                for (int k = 0; k < 16; ++k) v16[k] = take ? acc[16 + k] : 0.0;
                wave_sum16_add(v16, &sh.tot[wave][cls][16]);
#pragma unroll
                for (int k = 0; k < 32; ++k) acc[k] = take ? 0.0 : acc[k];
            };
            // nextb: the next class boundary at or after this wave's first pose (none: a huge index)
            int cls = class_of(wbase + 1), nextb = 1 << 30, afterb = 1 << 30;
            if constexpr (NL == 2) {
                if (bp[0] > wbase) { nextb = bp[0]; afterb = bp[1]; }
                else if (bp[1] > wbase) nextb = bp[1];
            }
#pragma unroll
            for (int s = 0; s < M; ++s) {
                const int j = jv + s * 64;
                const bool v = j <= L;
                double qb[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    qb[k] = lane_prev(b[s][k], cb[k]);
                    cb[k] = read_lane(b[s][k], 63);
                }
                bool cut = false;                             // a class boundary inside this block (rare)
                if constexpr (NL == 2) {
                    cut = nextb < wbase + 64 * s + 64;
                    if (cut) flush(cls, true);                // the blocks so far are of one class
                }
                const PoseQ yq = ld_pose(j), aq = ld_pose(j - 1);
                if (v) {
                    const double2* pl = rec_of(s);
                    Pose3 A, X;
                    Edge3 E;
                    double om[21];
                    {
                        double Rz[9], tz[3];
                        ld_rz(pl, Rz, tz);
                        ld_sym(pl, 6, om);
                        rot_of(aq, A);
                        rot_of(yq, X);
                        se3_edge(A, X, Rz, tz, E);
                    }
                    IPC3_FENCE();
                    double sg[21];
                    ld_sym(pl, 17, sg);
                    {
                        double w[6];
                        se3_apply_J(E, qb, b[s], w);
                        acc[27] += sym6_quad(om, w);
                        double bsq = 0.0;
#pragma unroll
                        for (int k = 0; k < 6; ++k) bsq += b[s][k] * b[s][k];
                        acc[28] += bsq;
                    }
                    asm volatile("" : "+v"(acc[27]), "+v"(acc[28]) : : "memory");
                    IPC3_FENCE();
                    // Phi = [[U, K],[0, Vq]]
                    double U[9], Vq[9], K[9];
                    m3_mult(X.R, E.RE, U);
                    {
                        double Qi[9];
                        const double iw = 1.0 / E.qw;
#pragma unroll
                        for (int i = 0; i < 3; ++i)
#pragma unroll
                            for (int k = 0; k < 3; ++k) Qi[3 * i + k] = E.qv[i] * E.qv[k] * iw + (i == k ? E.qw : 0.0);
                        Qi[1] += E.qv[2]; Qi[2] -= E.qv[1];
                        Qi[3] -= E.qv[2]; Qi[5] += E.qv[0];
                        Qi[6] += E.qv[1]; Qi[7] -= E.qv[0];
                        m3_mul(X.R, Qi, Vq);
                    }
                    {
                        const double tt[3] = {X.t[0] - gauge.t[0], X.t[1] - gauge.t[1], X.t[2] - gauge.t[2]};
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            K[0 + c] = 2 * (tt[1] * Vq[6 + c] - tt[2] * Vq[3 + c]);
                            K[3 + c] = 2 * (tt[2] * Vq[0 + c] - tt[0] * Vq[6 + c]);
                            K[6 + c] = 2 * (tt[0] * Vq[3 + c] - tt[1] * Vq[0 + c]);
                        }
                    }
                    // w_j = Phi e
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        acc[r] += U[3 * r] * E.e[0] + U[3 * r + 1] * E.e[1] + U[3 * r + 2] * E.e[2]
                                  + K[3 * r] * E.e[3] + K[3 * r + 1] * E.e[4] + K[3 * r + 2] * E.e[5];
                        acc[3 + r] += Vq[3 * r] * E.e[3] + Vq[3 * r + 1] * E.e[4] + Vq[3 * r + 2] * E.e[5];
                    }
                    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]) : : "memory");
                    IPC3_FENCE();
                    // Psi = Phi Sg Phi^T by 3x3 blocks (Sg = [[Stt, Stq],[Stq^T, Sqq]]):
                    //   T0 = U Stt + K Stq^T, T1 = U Stq + K Sqq, T2 = Vq Sqq
                    //   Psi_tt = T0 U^T + T1 K^T, Psi_tq = T1 Vq^T, Psi_qq = T2 Vq^T
                    auto S = [&](int a2, int c2) -> double { return sg[sym6_idx(a2, c2)]; };
                    {
                        double T1[9];
#pragma unroll
                        for (int r = 0; r < 3; ++r)
#pragma unroll
                            for (int c = 0; c < 3; ++c)
                                T1[3 * r + c] = U[3 * r] * S(0, 3 + c) + U[3 * r + 1] * S(1, 3 + c) + U[3 * r + 2] * S(2, 3 + c)
                                                + K[3 * r] * S(3, 3 + c) + K[3 * r + 1] * S(4, 3 + c) + K[3 * r + 2] * S(5, 3 + c);
                        // Psi_tq (rows 0..2, columns 3..5)
#pragma unroll
                        for (int r = 0; r < 3; ++r)
#pragma unroll
                            for (int c = 0; c < 3; ++c)
                                acc[6 + sym6_idx(r, 3 + c)] += T1[3 * r] * Vq[3 * c] + T1[3 * r + 1] * Vq[3 * c + 1] + T1[3 * r + 2] * Vq[3 * c + 2];
                        // T1 K^T part of Psi_tt
#pragma unroll
                        for (int r = 0; r < 3; ++r)
#pragma unroll
                            for (int c = r; c < 3; ++c)
                                acc[6 + sym6_idx(r, c)] += T1[3 * r] * K[3 * c] + T1[3 * r + 1] * K[3 * c + 1] + T1[3 * r + 2] * K[3 * c + 2];
                    }
                    asm volatile("" : "+v"(acc[6]), "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]),
                                 "+v"(acc[12]), "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15]), "+v"(acc[16]), "+v"(acc[17]),
                                 "+v"(acc[18]), "+v"(acc[19]), "+v"(acc[20]) : : "memory");
                    IPC3_FENCE();
                    {
                        double T0[9];
#pragma unroll
                        for (int r = 0; r < 3; ++r)
#pragma unroll
                            for (int c = 0; c < 3; ++c)
                                T0[3 * r + c] = U[3 * r] * S(0, c) + U[3 * r + 1] * S(1, c) + U[3 * r + 2] * S(2, c)
                                                + K[3 * r] * S(3, c) + K[3 * r + 1] * S(4, c) + K[3 * r + 2] * S(5, c);
#pragma unroll
                        for (int r = 0; r < 3; ++r)
#pragma unroll
                            for (int c = r; c < 3; ++c)
                                acc[6 + sym6_idx(r, c)] += T0[3 * r] * U[3 * c] + T0[3 * r + 1] * U[3 * c + 1] + T0[3 * r + 2] * U[3 * c + 2];
                    }
                    {
                        double T2[9];
#pragma unroll
                        for (int r = 0; r < 3; ++r)
#pragma unroll
                            for (int c = 0; c < 3; ++c)
                                T2[3 * r + c] = Vq[3 * r] * S(3, 3 + c) + Vq[3 * r + 1] * S(4, 3 + c) + Vq[3 * r + 2] * S(5, 3 + c);
#pragma unroll
                        for (int r = 0; r < 3; ++r)
#pragma unroll
                            for (int c = r; c < 3; ++c)
                                acc[6 + sym6_idx(3 + r, 3 + c)] += T2[3 * r] * Vq[3 * c] + T2[3 * r + 1] * Vq[3 * c + 1] + T2[3 * r + 2] * Vq[3 * c + 2];
                    }
                }
                if constexpr (NL == 2) {
                    if (cut) {                                // acc holds this block only: split it by class
                        const int bhi = wbase + 64 * s + 64;
                        while (nextb < bhi) {
                            flush(cls, j <= nextb);
                            cls = class_of(nextb + 1);
                            nextb = afterb;
                            afterb = 1 << 30;
                        }
                    }
                }
                asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]),
                             "+v"(acc[6]), "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]),
                             "+v"(acc[12]), "+v"(acc[13]), "+v"(acc[14]) : : "memory");
                asm volatile("" : "+v"(acc[15]), "+v"(acc[16]), "+v"(acc[17]), "+v"(acc[18]), "+v"(acc[19]), "+v"(acc[20]),
                             "+v"(acc[21]), "+v"(acc[22]), "+v"(acc[23]), "+v"(acc[24]), "+v"(acc[25]), "+v"(acc[26]),
                             "+v"(acc[27]), "+v"(acc[28]) : : "memory");
                IPC3_FENCE();
            }
            flush(cls, true);
            tbar();
            IPC3_TICK(tmB1)
            const int bufS = phase & 1;
            // The capacitance system is set up by the whole team (W > 1): wave 0 sums the class totals while wave 1
            // builds the Gamma_l, then every wave computes its share of the NS x (NS+1) augmented matrix; only the
            // Gauss-Jordan elimination (a serial chain) runs on wave 0 alone, next to the loops' own b^T H b terms
            // on wave 1.  With one wave doing all of it the other three idled for 10-17 % of an iteration.
            constexpr int RS = NS + 1;
            constexpr int kGamWave = W > 1 ? 1 : 0;
            if (wave == 0) {
                // totals: [0] b^T b, [1] b^T H b, [2..8) W_1, [8..29) M_11; pair: [29..35) W_2, [35..56) M_22, [56..77) M_12
                auto clsum = [&](int c, int k) -> double {
                    double t = sh.tot[0][c][k];
#pragma unroll
                    for (int w = 1; w < W; ++w) t += sh.tot[w][c][k];
                    return t;
                };
                // 27 values per target group, one lane each
                const int k = lane;
                if (k < 27) {
                    const double c1 = clsum(1, k), c3 = NL == 2 ? clsum(3, k) : 0.0, c2 = NL == 2 ? clsum(2, k) : 0.0;
                    sh.w0tot[2 + k] = c1 + c3;
                    if constexpr (NL == 2) { sh.w0tot[29 + k] = c2 + c3; if (k >= 6) sh.w0tot[56 + (k - 6)] = c3; }
                }
                if (k == 27) sh.w0tot[1] = (clsum(1, 27) + (NL == 2 ? clsum(2, 27) : 0.0)) + (NL == 2 ? clsum(3, 27) : 0.0);
                if (k == 28) sh.w0tot[0] = (clsum(1, 28) + (NL == 2 ? clsum(2, 28) : 0.0)) + (NL == 2 ? clsum(3, 28) : 0.0);
            }
            if (wave == kGamWave) {
                // Gamma_l entries (one lane per entry, one round per loop)
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    if (lane < 36) {
                        const int r = lane / 6, c = lane % 6;
                        const LoopConst3& q = sh.lc[l];
                        const LoopState3& st = sh.ls[cur][l];
                        const double tt[3] = {st.pt.t[0] - gauge.t[0], st.pt.t[1] - gauge.t[1], st.pt.t[2] - gauge.t[2]};
                        double g = 0.0;
                        if (r < 3) {
                            double Pr[3];
#pragma unroll
                            for (int k = 0; k < 3; ++k) Pr[k] = st.RE[3 * r] * st.pt.R[3 * k] + st.RE[3 * r + 1] * st.pt.R[3 * k + 1] + st.RE[3 * r + 2] * st.pt.R[3 * k + 2];
                            if (c < 3) g = Pr[c];
                            else {
                                const int cc = c - 3;
                                const double col[3] = {cc == 0 ? 0.0 : (cc == 1 ? -tt[2] : tt[1]),
                                                       cc == 0 ? tt[2] : (cc == 1 ? 0.0 : -tt[0]),
                                                       cc == 0 ? -tt[1] : (cc == 1 ? tt[0] : 0.0)};
                                g = -2 * (Pr[0] * col[0] + Pr[1] * col[1] + Pr[2] * col[2]);
                            }
                        } else if (c >= 3) {
                            const int rr = r - 3, cc = c - 3;
                            double Qr[3] = {rr == 0 ? st.qw : (rr == 1 ? st.qv[2] : -st.qv[1]),
                                            rr == 0 ? -st.qv[2] : (rr == 1 ? st.qw : st.qv[0]),
                                            rr == 0 ? st.qv[1] : (rr == 1 ? -st.qv[0] : st.qw)};
                            g = Qr[0] * st.pt.R[3 * cc] + Qr[1] * st.pt.R[3 * cc + 1] + Qr[2] * st.pt.R[3 * cc + 2];
                        }
                        sh.w0gam[l][lane] = q.sigma * g;
                    }
                }
            }
            tbar();
            // augmented matrix: entry idx = q0 + team lane, NS * RS entries over the team's 64 W lanes
            {
                constexpr int TL = 64 * W;
                const int tl = wave * 64 + lane;
#pragma unroll
                for (int q0 = 0; q0 < NS * RS; q0 += TL) {
                    const int idx = q0 + tl;
                    if (idx < NS * RS) {
                        const int r = idx / RS, c = idx % RS;
                        const int l1 = r / 6, i = r % 6;
                        const double* g1 = &sh.w0gam[l1][6 * i];
                        double vv;
                        if (c < NS) {
                            const int l2 = c / 6, k = c % 6;
                            const double* g2 = &sh.w0gam[l2][6 * k];
                            const int mb = l1 == l2 ? (l1 == 0 ? 8 : 35) : 56;
                            vv = 0.0;
#pragma unroll
                            for (int bq = 0; bq < 6; ++bq) {
                                double t = 0.0;
#pragma unroll
                                for (int a = 0; a < 6; ++a) t += g1[a] * sh.w0tot[mb + sym6_idx(a, bq)];
                                vv += t * g2[bq];
                            }
                            if (l1 == l2) vv += sh.lc[l1].sg[sym6_idx(i, k)];
                        } else {
                            const int wb = l1 == 0 ? 2 : 29;
                            double t = 0.0;
#pragma unroll
                            for (int a = 0; a < 6; ++a) t += g1[a] * sh.w0tot[wb + a];
                            vv = sh.ls[cur][l1].e[i] - t;
                        }
                        sh.w0aug[r][c] = vv;
                    }
                }
            }
            tbar();
            if (wave == kGamWave && lane < NL) sh.w0lq[lane] = loop_quad(lane, bufB);
            if (wave == 0) {
                // Gauss-Jordan, one row per lane
                double row[RS];
                const int rl = lane < NS ? lane : 0;
#pragma unroll
                for (int c = 0; c < RS; ++c) row[c] = sh.w0aug[rl][c];
                bool okS = true;
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const double piv = read_lane(row[k], k);
                    okS = okS && (piv > 0);
                    double inv = __builtin_amdgcn_rcp(piv);
                    inv = fma(fma(-piv, inv, 1.0), inv, inv);
                    inv = fma(fma(-piv, inv, 1.0), inv, inv);
                    const double f = row[k] * inv;
#pragma unroll
                    for (int c = k; c < RS; ++c) {
                        const double pr = read_lane(row[c], k);
                        row[c] = (lane == k) ? pr * inv : fma(-f, pr, row[c]);
                    }
                }
                if (lane < NS) sh.w0mu[lane] = row[NS];
                wave_sync3();
                if (lane < NS) {
                    const int l = lane / 6, c = lane % 6;
                    double t = 0.0;
#pragma unroll
                    for (int r = 0; r < 6; ++r) t += sh.w0gam[l][6 * r + c] * sh.w0mu[6 * l + r];
                    sh.sol[bufS][lane] = t;
                }
                if (lane == 0) {
                    sh.sol[bufS][NS] = sh.w0tot[0];
                    sh.sol[bufS][NS + 1] = sh.w0tot[1];
                    sh.sol[bufS][NS + 2] = okS ? 1.0 : 0.0;
                }
            }
            tbar();
            ++phase;
#pragma unroll
            for (int l = 0; l < NL; ++l)
#pragma unroll
                for (int k = 0; k < 6; ++k) nu[l][k] = sh.sol[bufS][6 * l + k];
            bb = sh.sol[bufS][NS];
            bHb = sh.sol[bufS][NS + 1] + sh.w0lq[0];
            if (NL == 2) bHb += sh.w0lq[1];
            if (sh.sol[bufS][NS + 2] == 0.0) { flags |= 2; break; }
            alpha = bb / bHb;
            hsdNorm = sqrt(alpha * alpha * bb);
        }

        IPC3_TICK(tmB2)
        // ---- phase C: u = -Sg Phi^T n - e, rho = D^-1 u, world-frame prefix sums -> h ----
        double hgnNorm, bh, hHh;
        opaque();
        {
            // pass 1: world-frame increments, wave-local prefix sums (slot after slot); h[s] holds
            // (tau_local, omega_local)
            double co[3] = {0, 0, 0}, ct[3] = {0, 0, 0};
#pragma unroll
            for (int s = 0; s < M; ++s) {
                const int j = jv + s * 64;
                const bool v = j <= L;
                double rq[3] = {0, 0, 0}, rt[3] = {0, 0, 0}, d3[3] = {0, 0, 0};
                const PoseQ yq = ld_pose(j), aq = ld_pose(j - 1);
                if (v) {
                    const double2* pl = rec_of(s);
                    Pose3 A, X;
                    Edge3 E;
                    double sg[21];
                    {
                        double Rz[9], tz[3];
                        ld_rz(pl, Rz, tz);
                        ld_sym(pl, 17, sg);
                        rot_of(aq, A);
                        rot_of(yq, X);
                        se3_edge(A, X, Rz, tz, E);
                    }
                    IPC3_FENCE();
                    double nn[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
                    for (int l = 0; l < NL; ++l) {
                        const double ml = (j > rlo[l] && j <= rhi[l]) ? 1.0 : 0.0;
#pragma unroll
                        for (int k = 0; k < 6; ++k) nn[k] += ml * nu[l][k];
                    }
                    double wv[6];
                    {
                        double U[9], Qi[9], Vq[9];
                        m3_mult(X.R, E.RE, U);
                        const double iw = 1.0 / E.qw;
#pragma unroll
                        for (int i = 0; i < 3; ++i)
#pragma unroll
                            for (int k = 0; k < 3; ++k) Qi[3 * i + k] = E.qv[i] * E.qv[k] * iw + (i == k ? E.qw : 0.0);
                        Qi[1] += E.qv[2]; Qi[2] -= E.qv[1];
                        Qi[3] -= E.qv[2]; Qi[5] += E.qv[0];
                        Qi[6] += E.qv[1]; Qi[7] -= E.qv[0];
                        m3_mul(X.R, Qi, Vq);
                        const double tt[3] = {X.t[0] - gauge.t[0], X.t[1] - gauge.t[1], X.t[2] - gauge.t[2]};
                        double cr[3], y3[3];
                        cross3(tt, nn, cr);
#pragma unroll
                        for (int k = 0; k < 3; ++k) y3[k] = nn[3 + k] - 2 * cr[k];
                        m3_tvec(U, nn, wv);
                        m3_tvec(Vq, y3, wv + 3);
                    }
                    double vv[6], u[6], rho[6];
                    sym6_mul(sg, wv, vv);
#pragma unroll
                    for (int k = 0; k < 6; ++k) u[k] = -vv[k] - E.e[k];
                    se3_Dinv(E, u, rho);
                    m3_vec(X.R, rho + 3, rq);
                    m3_vec(X.R, rho, rt);
#pragma unroll
                    for (int k = 0; k < 3; ++k) d3[k] = X.t[k] - A.t[k];
                }
                double lo3[3], op[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    lo3[k] = wave_inclusive_scan(rq[k]) + co[k];
                    op[k] = lane_prev(lo3[k], co[k]);          // wave-local omega of the previous pose
                    co[k] = read_lane(lo3[k], 63);
                }
                double term[3], c3[3];
                cross3(op, d3, c3);
#pragma unroll
                for (int k = 0; k < 3; ++k) term[k] = v ? rt[k] + 2 * c3[k] : 0.0;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double lt3 = wave_inclusive_scan(term[k]) + ct[k];
                    ct[k] = read_lane(lt3, 63);
                    h[s][k] = lt3;
                    h[s][3 + k] = lo3[k];
                }
                asm volatile("" : "+v"(h[s][0]), "+v"(h[s][1]), "+v"(h[s][2]), "+v"(h[s][3]), "+v"(h[s][4]), "+v"(h[s][5]) : : "memory");
                IPC3_FENCE();
            }
            // bases of this wave: omega, tau of its predecessor pose (index wbase)
            double bo[3] = {0, 0, 0}, bt[3] = {0, 0, 0};
            const PoseQ edge = ld_pose(wbase <= L ? wbase : 0);
            if constexpr (W > 1) {
                const int buf = phase & 1;
                if (lane == 0) {
                    const int pl = min(L, wbase + 64 * M);
                    const int pe = wbase <= L ? wbase : 0;
                    const int pq = wbase <= L ? pl : 0;
                    sh.scan[buf][wave][0] = co[0]; sh.scan[buf][wave][1] = co[1]; sh.scan[buf][wave][2] = co[2];
                    sh.scan[buf][wave][3] = ct[0]; sh.scan[buf][wave][4] = ct[1]; sh.scan[buf][wave][5] = ct[2];
                    sh.scan[buf][wave][6] = sh.t0[pq] - sh.t0[pe];
                    sh.scan[buf][wave][7] = sh.t1[pq] - sh.t1[pe];
                    sh.scan[buf][wave][8] = sh.t2[pq] - sh.t2[pe];
                }
                IPC3_SYNC();
                ++phase;
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    if (w < wave) {
                        double dT[3] = {sh.scan[buf][w][6], sh.scan[buf][w][7], sh.scan[buf][w][8]}, c[3];
                        cross3(bo, dT, c);
#pragma unroll
                        for (int k = 0; k < 3; ++k) bt[k] += sh.scan[buf][w][3 + k] + 2 * c[k];
#pragma unroll
                        for (int k = 0; k < 3; ++k) bo[k] += sh.scan[buf][w][k];
                    }
                }
            }
            // pass 2: h in the body frame, |h|^2, b.h  (h^T H h = b.h since h solves H h = b)
            double p0 = 0.0, p1 = 0.0;
#pragma unroll
            for (int s = 0; s < M; ++s) {
                const int j = jv + s * 64;
                const bool v = j <= L;
                const PoseQ yq = ld_pose(j);
                if (v) {
                    Pose3 X;
                    rot_of(yq, X);
                    double om3[3], ta3[3], d3[3] = {X.t[0] - edge.t[0], X.t[1] - edge.t[1], X.t[2] - edge.t[2]}, c[3];
                    cross3(bo, d3, c);
#pragma unroll
                    for (int k = 0; k < 3; ++k) { om3[k] = h[s][3 + k] + bo[k]; ta3[k] = h[s][k] + bt[k] + 2 * c[k]; }
                    m3_tvec(X.R, ta3, &h[s][0]);
                    m3_tvec(X.R, om3, &h[s][3]);
#pragma unroll
                    for (int k = 0; k < 6; ++k) { p0 += h[s][k] * h[s][k]; p1 += b[s][k] * h[s][k]; }
                } else {
#pragma unroll
                    for (int k = 0; k < 6; ++k) h[s][k] = 0.0;
                }
                asm volatile("" : "+v"(h[s][0]), "+v"(h[s][1]), "+v"(h[s][2]), "+v"(h[s][3]), "+v"(h[s][4]), "+v"(h[s][5]),
                             "+v"(p0), "+v"(p1) : : "memory");
                IPC3_FENCE();
            }
            team_sum2(p0, p1);
            hgnNorm = sqrt(p0);
            bh = p1;
            hHh = bh;
        }

        IPC3_TICK(tmC)
        // converged (Se2View::term_eps): in the Newton regime (the last iteration took the full Gauss-Newton
        // step at its first trial) and one more such step cannot move any edge's chi2 by more than
        // 2 sqrt(term_eps) relative; g2o would still run its trial loop to Terminate
        if (lastGN && hgnNorm < delta && fabs(bh) < term_scale * currentChi) { it_done = it + 1; tries_total += maxTrials; flags |= 1; break; }
        // ---- trial loop ----
        const double deltaAtEntry = delta;
        bool goodStep = false, dlReady = false;
        double dlC = 0.0, dlBma = 0.0;
        int numTries = 0;
        do {
            ++numTries;
            int stepType;
            double beta = 0.0, sdScale = 0.0;
            if (hgnNorm < delta) stepType = 0;
            else if (hsdNorm > delta) { stepType = 1; sdScale = delta / hsdNorm; }
            else {
                stepType = 2;
                // delta-independent: reduced once per iteration, reused by its later dog-leg trials (same bits)
                if (!dlReady) {
                    double p0 = 0.0, p1 = 0.0;
#pragma unroll
                    for (int s = 0; s < M; ++s) {
#pragma unroll
                        for (int k = 0; k < 6; ++k) {
                            const double sk = alpha * b[s][k], ak = h[s][k] - sk;
                            p0 += sk * ak;
                            p1 += ak * ak;
                        }
                    }
                    team_sum2(p0, p1);
                    dlC = p0; dlBma = p1;
                    dlReady = true;
                }
                const double c = dlC, bma = dlBma, hsdSq = alpha * alpha * bb;
                if (c <= 0.) beta = (-c + sqrt(c * c + bma * (delta * delta - hsdSq))) / bma;
                else beta = (delta * delta - hsdSq) / (c + sqrt(c * c + bma * (delta * delta - hsdSq)));
            }
            double pcoef, qcoef, hdlNorm;
            if (stepType == 0) { pcoef = 0.0; qcoef = 1.0; hdlNorm = hgnNorm; }
            else if (stepType == 1) { pcoef = sdScale * alpha; qcoef = 0.0; hdlNorm = delta; }
            else { pcoef = alpha - beta * alpha; qcoef = beta; hdlNorm = delta; }
            const double hdlHhdl = pcoef * pcoef * bHb + 2 * pcoef * qcoef * bb + qcoef * qcoef * hHh;
            const double bhdl = pcoef * bb + qcoef * bh;
            double linearGain = -1 * hdlHhdl + 2 * bhdl;
            const int trial = cur ^ 1;
            const double newChi = sweep(std::integral_constant<int, 1>{}, stepType, pcoef, qcoef, trial);
            const bool anyChanged = stepType == 1 ? sweepChanged : true;
            ++evals;
            const double nonLinearGain = currentChi - newChi;
            if (fabs(linearGain) < 1e-12) linearGain = 1e-12;
            const bool linPos = linearGain > 0;
            auto rho_gt = [&](double t) { return linPos ? nonLinearGain > t * linearGain : nonLinearGain < t * linearGain; };
            auto rho_lt = [&](double t) { return linPos ? nonLinearGain < t * linearGain : nonLinearGain > t * linearGain; };
            if (rho_gt(0.0)) {
                goodStep = true;
                currentChi = newChi;
                cur = trial;
                IPC3_TICK(tmT)
                sweep(std::integral_constant<int, 2>{}, stepType, pcoef, qcoef, trial);
                IPC3_TICK(tmK)
            }
            if (rho_gt(0.75)) delta = fmax(delta, 3 * hdlNorm);
            else if (rho_lt(0.25)) delta *= 0.5;
            if (!goodStep) {
                if (nonLinearGain != nonLinearGain) {
                    numTries = maxTrials;       // NaN gain ratio: g2o leaves delta alone, so every retry is this same trial
                } else if (stepType == 0) {
                    while (numTries < maxTrials && hgnNorm < delta) { ++numTries; delta *= 0.5; }
                } else if (stepType == 1 && !anyChanged) {
                    numTries = maxTrials;
                }
            }
        } while (!goodStep && numTries < maxTrials);
        lastGN = goodStep && numTries == 1 && hgnNorm < deltaAtEntry;
        it_done = it + 1;
        tries_total += numTries;
        IPC3_TICK(tmT)
        if (numTries == maxTrials || !goodStep) { flags |= 1; break; }
    }
#ifdef IPC_PHASE_TIMING
    if (lane == 0 && (wave == 0 || wave == W - 1) && P.dbg) {
        // slot: 1024 + 32 * (M + 16 * (W > 1)) + 16 * (wave != 0 || W == 1 ? 0 : 1) ... wave 0 first, last wave second
        unsigned long long* d = reinterpret_cast<unsigned long long*>(P.dbg) + 1024 + 32 * (M + 16 * (W > 1 ? 1 : 0)) + (wave == 0 ? 0 : 16);
        atomicAdd(d + 0, tmA); atomicAdd(d + 1, tmB1); atomicAdd(d + 2, tmB2); atomicAdd(d + 3, tmC);
        atomicAdd(d + 4, tmT); atomicAdd(d + 5, tmK); atomicAdd(d + 6, tmBar);
        atomicAdd(d + 7, (unsigned long long)it_done); atomicAdd(d + 8, (unsigned long long)evals);
        atomicAdd(d + 9, (unsigned long long)it_done * (unsigned long long)L); atomicAdd(d + 10, 1ull);
    }
#endif

    // ---- per-edge chi2 (consensus_utils.cpp:15-19) ----
    tbar();
    double mx = sweep(std::integral_constant<int, 3>{}, 0, 0.0, 0.0, cur);
    bool nan = sweepNan;
    tbar();
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
