// SE(2) cell solver, one WAVE (W = 1, chains up to 64 M poses) or a PAIR of waves (W = 2, up to
// 128 M poses) per cell.
//
// The block kernel (se2_cell.hpp) spreads one chain over W waves and pays a workgroup barrier
// plus an LDS round trip for every reduction / scan / neighbour hand-off, with one cell resident
// per CU.  Here:
//   * lane l owns the M CONSECUTIVE poses l*M+1 .. l*M+M (lane-major), so the chain neighbour
//     of a slot is the previous slot of the same lane (a register); only slot 0 takes one DPP
//     shift from the lane below.  Prefix sums are an in-lane serial pass + ONE wave scan,
//     reductions are in-lane accumulation + ONE wave reduction.  No workgroup barrier anywhere.
//   * the four waves of a workgroup (one per SIMD, up to 512 registers each) work on four
//     different cells (W = 1) or two (W = 2) and fetch the next one from a global counter when
//     done, so a CU always has independent dog-legs in flight.
//   * W = 2: the two waves of a cell exchange the boundary pose / force / step vectors, partial
//     sums and scan carries through a sequence-numbered, double-buffered LDS mailbox (PairBox).
//   * the chain constants of the residual pass (5 measurement + 6 information values per edge)
//     are staged ONCE per workgroup for the whole window of the launch, one record per edge, and
//     shared by every cell it solves; M is odd so the lane stride of M records is bank-conflict
//     free, and a slot's constants sit at compile-time offsets from one per-lane address.
//   * trial poses / trial errors are not stored: a trial is one sweep that steps, evaluates and
//     sums chi2 on the fly; an accepted trial is re-swept once to commit.  For M > 8 the committed
//     errors are not stored either (recomputed where used).
//   * with one wave per SIMD every instruction costs issue time: the per-slot code is pinned in
//     place (IPC_PIN*), loop end points are found by one bit test per slot, constants are fetched
//     one slot ahead where registers allow.
// Mathematics, dog-leg control flow and shortcuts are those of se2_cell.hpp.
#pragma once
#include "se2_cell.hpp"

namespace ipc {

template <int NL>
struct WaveScratch {          // per cell (one wave, or a pair of waves), LDS
    LoopConst lc[NL];
    LoopState ls[2][NL];
    double lvec[NL][2][3];
    double red[4][32];        // wide reduction, one row per wave of the cell
    double gam[4][NL * 9];    // Gamma_l of the capacitance assembly (each wave keeps its own copy)
};

// Mailbox of a wave pair that solves one cell together (W == 2).  The two waves run on different
// SIMDs of the same workgroup and execute the same sequence of exchanges; an exchange is "write
// my values, post my sequence number, wait for the partner's, read the partner's values", double
// buffered on the sequence parity.  LDS operations of one wave complete in order, so the data is
// visible before the flag.
//
// The payload is moved with RELAXED ATOMIC loads / stores (workgroup scope), not plain ones.  The instructions are the
// same ds_read / ds_write, but (i) the protocol is then race-free for every lane by the memory model -- with plain
// accesses only lane 0, which posts the flag, is ordered against the partner's next overwrite of the buffer; the other
// lanes' reads are ordered by the wave's lock-step execution, which the model knows nothing about -- and (ii) the
// compiler waits for a payload read where it is issued instead of leaving it in flight across the exec-mask juggling
// and the branches that follow.  With plain accesses the kernels whose errors are recomputed (M > 8) give wrong,
// run-to-run varying results when built with -mllvm -amdgpu-sched-strategy=max-ilp (round 4, DESIGN.md 7:
// tools/maxilp_repro.py; the read of the partner's chi2 partial sum in the trial sweep is the one that matters);
// with atomic accesses both scheduling strategies give bit-identical results.  -DIPC_MAILBOX_PLAIN restores the plain
// accesses for that reproducer.
struct PairBox {
    double data[4][2][8];     // [wave of the cell][parity][value]
    int flag[4];
};
__device__ __forceinline__ void mb_store(double* p, double v)
{
#ifdef IPC_MAILBOX_PLAIN
    *p = v;
#else
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}
__device__ __forceinline__ double mb_load(const double* p)
{
#ifdef IPC_MAILBOX_PLAIN
    return *p;
#else
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}

// what a wave does between two polls of its partner's mailbox flag
#ifndef IPC_SPIN_SLEEP
#define IPC_SPIN_SLEEP 1
#endif
#if IPC_SPIN_SLEEP > 0
#define IPC_SPIN_WAIT() __builtin_amdgcn_s_sleep(IPC_SPIN_SLEEP)
#else
#define IPC_SPIN_WAIT() asm volatile("s_nop 0")
#endif

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// SPD solve A x = b by LDL^T (N = 3 or 6) on wave-uniform values; reciprocals by v_rcp_f64 + two
// Newton steps.  Returns false when a pivot is not positive.
template <int N>
__device__ __forceinline__ bool ldl_solve(double (&A)[N][N], double (&b)[N])
{
    bool ok = true;
    double inv[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double d = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= A[j][k] * A[j][k] * A[k][k];
        ok = ok && (d > 0);
        double r = __builtin_amdgcn_rcp(d);
        r = fma(fma(-d, r, 1.0), r, r);
        r = fma(fma(-d, r, 1.0), r, r);
        inv[j] = r;
        A[j][j] = d;
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            double v = A[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) v -= A[i][k] * A[j][k] * A[k][k];
            A[i][j] = v * r;
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double v = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) v -= A[i][k] * b[k];
        b[i] = v;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) b[i] *= inv[i];
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        double v = b[i];
#pragma unroll
        for (int k = i + 1; k < N; ++k) v -= A[k][i] * b[k];
        b[i] = v;
    }
    return ok;
}

template <int V> struct IntC { static constexpr int value = V; };

// The per-slot loops are fully unrolled (the state lives in registers).  Memory operations keep
// their source order, pure arithmetic does not: left alone it sinks towards its final uses, every
// operand loaded for the M slots stays live and the kernel spills (> 1000 VGPRs at M = 13).  An
// empty volatile asm that consumes the results of a slot pins that slot's arithmetic in place;
// the scheduling barrier keeps the next slot's loads behind it (measured: without it the kernels
// use a few registers less but C2 runs 6 % slower).
#define IPC_SLOT_FENCE() __builtin_amdgcn_sched_barrier(0)
#define IPC_PIN1(a) asm volatile("" : "+v"(a) : : "memory")
#define IPC_PIN2(a, b) asm volatile("" : "+v"(a), "+v"(b) : : "memory")
#define IPC_PIN3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c) : : "memory")

// KEEP_E: the odometry errors of the committed state live in registers (3 doubles per pose).  For
// large M they are recomputed from the poses where needed (three times per iteration, ~20 flops
// each) to stay inside the register file; a commit is then a pure pose update.
#ifndef IPC_KEEPE_MAX
#define IPC_KEEPE_MAX 8
#endif
#ifndef IPC_PF_MAX
#define IPC_PF_MAX 9
#endif
template <int M, int NL, bool STAGED, int W = 1, bool KEEP_E = (M <= IPC_KEEPE_MAX)>
__device__ __forceinline__ void se2_wave_solve(const Se2View& P, int lo_abs, int L, const int (&cand)[2], int iterations,
                               WaveScratch<NL>& sh, const double* cst, int wlo, int wstride, CellResult& res,
                               PairBox* box = nullptr, int seq0 = 0, int* seq_out = nullptr)
{
    static_assert(W == 1 || W == 2 || W == 4, "one, two or four cooperating waves per cell");
    constexpr int NS = NL * 3;
    const double term_scale = P.term_eps / (double)(L + NL);   // 0: the test is off
    bool lastGN = false;
    const int lane = threadIdx.x & 63;
    const int wsub = W > 1 ? ((threadIdx.x >> 6) & (W - 1)) : 0;  // wave of the cell
    const int gl = wsub * 64 + lane;                 // lane index within the cell
    const int j0 = gl * M + 1;                       // pose index of slot 0

    // ---- exchange among the W waves of the cell: every wave posts up to 8 doubles and waits for
    // all others; peer(o, k) then reads wave o's value k.  Also the cell's barrier. ----
#ifdef IPC_PHASE_TIMING
    unsigned long long gnEvals = 0, gnRejected = 0, nBig = 0, tmBig = 0, tmSmall = 0;
    unsigned long long tmA = 0, tmB1 = 0, tmB2 = 0, tmC = 0, tmT = 0, tmW = 0, tmK = 0, tm0 = __builtin_amdgcn_s_memtime();
#define IPC_WTICK(acc) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc += t_ - tm0; tm0 = t_; }
#else
#define IPC_WTICK(acc)
#endif
    int seq = seq0;
    auto post_wait = [&](const double (&mine)[8], int n) {
        if constexpr (W > 1) {
            ++seq;
            double* my = box->data[wsub][seq & 1];
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) if (k < n) mb_store(&my[k], mine[k]);
            }
            wave_sync();
            if (lane == 0) __hip_atomic_store(&box->flag[wsub], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef IPC_PHASE_TIMING
            const unsigned long long tw0 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
            for (int o = 0; o < W; ++o) {
                if (o == wsub) continue;
                while (__hip_atomic_load(&box->flag[o], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) - seq < 0)
                    IPC_SPIN_WAIT();
            }
#ifdef IPC_PHASE_TIMING
            tmW += __builtin_amdgcn_s_memtime() - tw0;
#endif
#ifdef IPC_MB_DELAY_WAVE                               // (protocol stress: one wave of the cell dawdles behind every exchange)
            if (wsub == IPC_MB_DELAY_WAVE) { __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); }
#endif
            // Every lane polls the flag with an acquire load above, so every lane's payload reads are ordered behind the
            // partner's release; the fence states that once more for the reads that follow, whatever the compiler makes of
            // the loop (LDS only: a fence over global memory would wait for the constant prefetches in flight).
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            wave_sync();
        }
    };
    auto peer = [&](int o, int k) -> double { return mb_load(&box->data[o][seq & 1][k]); };
    auto pair_barrier = [&]() {
        if constexpr (W > 1) { double a[8] = {0, 0, 0, 0, 0, 0, 0, 0}; post_wait(a, 0); }
    };
    // sums over the cell of per-lane values: the waves' totals added in wave order on every wave
    auto cell_sum2 = [&](double v0, double v1, double& s0, double& s1) {
        const double t0 = wave_sum(v0), t1 = wave_sum(v1);
        if constexpr (W > 1) {
            double a[8] = {t0, t1, 0, 0, 0, 0, 0, 0};
            post_wait(a, 2);
#if defined(IPC_MAILBOX_PLAIN) && defined(IPC_DBG_SUM_SIDE)    // (tools/maxilp_repro.py: which wave's read goes wrong)
            auto rdx = [&](int o, int k) -> double {
                return ((IPC_DBG_SUM_SIDE >> o) & 1) ? __hip_atomic_load(&box->data[o][seq & 1][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : peer(o, k);
            };
#else
            auto rdx = [&](int o, int k) -> double { return peer(o, k); };
#endif
            s0 = wsub == 0 ? t0 : rdx(0, 0);
            s1 = wsub == 0 ? t1 : rdx(0, 1);
#pragma unroll
            for (int o = 1; o < W; ++o) { s0 += o == wsub ? t0 : rdx(o, 0); s1 += o == wsub ? t1 : rdx(o, 1); }
        } else { s0 = t0; s1 = t1; }
    };
    // exclusive prefix over the cell's lanes of a per-lane total
    auto cell_excl = [&](double run) -> double {
        const double inc = wave_inclusive_scan(run);
        double off = inc - run;
        if constexpr (W > 1) {
            double a[8] = {read_lane(inc, 63), 0, 0, 0, 0, 0, 0, 0};
            post_wait(a, 1);
            double lower = 0.0;
#pragma unroll
            for (int o = 0; o < W - 1; ++o) if (o < wsub) lower += peer(o, 0);
            off += lower;
        }
        return off;
    };
    auto cell_excl2 = [&](double r0, double r1, double& o0, double& o1) {
        const double i0 = wave_inclusive_scan(r0), i1 = wave_inclusive_scan(r1);
        o0 = i0 - r0; o1 = i1 - r1;
        if constexpr (W > 1) {
            double a[8] = {read_lane(i0, 63), read_lane(i1, 63), 0, 0, 0, 0, 0, 0};
            post_wait(a, 2);
            double l0 = 0.0, l1 = 0.0;
#pragma unroll
            for (int o = 0; o < W - 1; ++o) if (o < wsub) { l0 += peer(o, 0); l1 += peer(o, 1); }
            o0 += l0; o1 += l1;
        }
    };

    // ---------------- loop constants -> LDS ----------------
    if (lane < NL && wsub == 0) {
        const int l = lane, c = cand[l];
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
    Pose2 X[M];
    double ex[KEEP_E ? M : 1], ey[KEEP_E ? M : 1], eth[KEEP_E ? M : 1];
    double bx[M], by[M], bth[M];
    double hx[M], hy[M], hth[M];
    Pose2 gauge;
    gauge.x = P.pose0[lo_abs];
    gauge.y = P.pose0[(size_t)P.V + lo_abs];
    gauge.th = P.pose0[(size_t)2 * P.V + lo_abs];
    sincos_pi(gauge.th, gauge.s, gauge.c);
#pragma unroll
    for (int s = 0; s < M; ++s) {
        const int j = j0 + s;
        const int ja = j <= L ? lo_abs + j : lo_abs;
        X[s].x = P.pose0[ja];
        X[s].y = P.pose0[(size_t)P.V + ja];
        X[s].th = P.pose0[(size_t)2 * P.V + ja];
        sincos_pi(X[s].th, X[s].s, X[s].c);
        if (KEEP_E) ex[s] = ey[s] = eth[s] = 0.0;
        bx[s] = by[s] = bth[s] = 0.0;
        hx[s] = hy[s] = hth[s] = 0.0;
    }
    wave_sync();                                     // sh.lc visible
    pair_barrier();
    int lf[NL], lt[NL], of[NL], sf[NL], ot[NL], st_[NL], llo[NL], lhi[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        lf[l] = __builtin_amdgcn_readfirstlane(sh.lc[l].f);
        lt[l] = __builtin_amdgcn_readfirstlane(sh.lc[l].t);
        llo[l] = min(lf[l], lt[l]); lhi[l] = max(lf[l], lt[l]);
        of[l] = lf[l] > 0 ? (lf[l] - 1) / M : -1; sf[l] = lf[l] > 0 ? (lf[l] - 1) % M : -1;
        ot[l] = lt[l] > 0 ? (lt[l] - 1) / M : -1; st_[l] = lt[l] > 0 ? (lt[l] - 1) % M : -1;
    }
    // Slots of THIS wave that hold a loop end point (wave-uniform bit mask): the per-slot loops test
    // one bit instead of running the four owner checks in every slot.
    unsigned ownSlots = 0;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        if (of[l] >= 0 && (of[l] >> 6) == wsub) ownSlots |= 1u << sf[l];
        if (ot[l] >= 0 && (ot[l] >> 6) == wsub) ownSlots |= 1u << st_[l];
    }
    if (gl == 0) {                                   // gauge end points never change
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

    // chain constants: local edge index of slot s is eloc + s (idle lanes are parked on edge 0;
    // the partially filled lane reads up to M-1 records past the chain, which the window padding
    // and the record-array padding cover)
    // The constants are loop-invariant, so the compiler would hoist every load out of the dog-leg
    // loop and pin 17 doubles per slot in registers; opaque() makes the index look modified, which
    // keeps the loads where they are used.
    int eloc = gl * M < L ? gl * M : 0;
    auto opaque = [&]() { asm volatile("" : "+v"(eloc)); };
    auto ldc = [&](int field, int s) -> double {
        // staged window: one record of F_SG doubles per edge, so a slot's constants sit at
        // compile-time offsets from ONE per-lane address (no address arithmetic, paired reads);
        // with M odd the lane stride of M * 88 bytes is bank-conflict free
        if (STAGED && field < (int)F_SG) return cst[((lo_abs - wlo + eloc) + s) * (int)F_SG + field];
        // HBM / L2: record-major copy of the chain, so these loads too are compile-time offsets
        // from one per-lane address
        return P.chain_rec[((size_t)(lo_abs + eloc) + s) * (int)F_NFIELDS + field];
    };
    auto ldsym = [&](int field0, int s) -> Sym3 {
        Sym3 m;
        m.a00 = ldc(field0 + 0, s); m.a01 = ldc(field0 + 1, s); m.a02 = ldc(field0 + 2, s);
        m.a11 = ldc(field0 + 3, s); m.a12 = ldc(field0 + 4, s); m.a22 = ldc(field0 + 5, s);
        return m;
    };
    // Constants of one slot, fetched ONE SLOT AHEAD of their use: the per-slot fences keep a slot's
    // loads inside that slot, so without the look-ahead every slot starts by waiting for its own LDS
    // (or L2) reads.  WANT bits: 1 measurement translation + angle, 2 measurement rotation,
    // 4 information, 8 covariance.
    struct SlotConst { double tzx, tzy, thz, cz, sz; Sym3 om, sg; };
    auto ld_slot = [&](int s, SlotConst& c, auto want_c) {
        constexpr int WANT = decltype(want_c)::value;
        if (WANT & 1) { c.tzx = ldc(F_TZX, s); c.tzy = ldc(F_TZY, s); c.thz = ldc(F_THZ, s); }
        if (WANT & 2) { c.cz = ldc(F_CZ, s); c.sz = ldc(F_SZ, s); }
        if (WANT & 4) c.om = ldsym(F_OM, s);
        if (WANT & 8) c.sg = ldsym(F_SG, s);
    };
    constexpr int kErrWant = KEEP_E ? 0 : 3;         // what err_of needs when the errors are recomputed
    // the look-ahead costs one SlotConst of registers: only where the register file has room
    constexpr bool PF = M <= IPC_PF_MAX;

    // pose j-1 of slot 0: slot M-1 of the lane below (lane 0 of the cell: the gauge; lane 0 of the
    // pair's second wave: lane 63 of the first, through the mailbox)
    auto prev0 = [&](const Pose2& last) -> Pose2 {
        Pose2 cr = gauge;
        if constexpr (W > 1) {
            double a[8] = {read_lane(last.x, 63), read_lane(last.y, 63), read_lane(last.th, 63), read_lane(last.c, 63),
                           read_lane(last.s, 63), 0, 0, 0};
            post_wait(a, 5);
            if (wsub > 0) { cr.x = peer(wsub - 1, 0); cr.y = peer(wsub - 1, 1); cr.th = peer(wsub - 1, 2); cr.c = peer(wsub - 1, 3); cr.s = peer(wsub - 1, 4); }
        }
        Pose2 p;
        p.x = lane_prev(last.x, cr.x); p.y = lane_prev(last.y, cr.y); p.th = lane_prev(last.th, cr.th);
        p.c = lane_prev(last.c, cr.c); p.s = lane_prev(last.s, cr.s);
        return p;
    };
    // 3-vector of the lane below / above across the pair (0 at the ends of the cell)
    auto prev3 = [&](double vx, double vy, double vth, double& ox, double& oy, double& oth) {
        double c0 = 0.0, c1 = 0.0, c2 = 0.0;
        if constexpr (W > 1) {
            double a[8] = {read_lane(vx, 63), read_lane(vy, 63), read_lane(vth, 63), 0, 0, 0, 0, 0};
            post_wait(a, 3);
            if (wsub > 0) { c0 = peer(wsub - 1, 0); c1 = peer(wsub - 1, 1); c2 = peer(wsub - 1, 2); }
        }
        ox = lane_prev(vx, c0); oy = lane_prev(vy, c1); oth = lane_prev(vth, c2);
    };
    auto next3 = [&](double vx, double vy, double vth, double& ox, double& oy, double& oth) {
        double c0 = 0.0, c1 = 0.0, c2 = 0.0;
        if constexpr (W > 1) {
            double a[8] = {read_lane(vx, 0), read_lane(vy, 0), read_lane(vth, 0), 0, 0, 0, 0, 0};
            post_wait(a, 3);
            if (wsub < W - 1) { c0 = peer(wsub + 1, 0); c1 = peer(wsub + 1, 1); c2 = peer(wsub + 1, 2); }
        }
        ox = lane_next(vx, c0); oy = lane_next(vy, c1); oth = lane_next(vth, c2);
    };

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
    int cur = 0;
    auto loop_force = [&](int l) {
        const LoopConst& q = sh.lc[l];
        LoopState& st = sh.ls[cur][l];
        Sym3 om{q.om[0], q.om[1], q.om[2], q.om[3], q.om[4], q.om[5]};
        double q0, q1, q2;
        om.mul(st.e[0], st.e[1], st.e[2], q0, q1, q2);
        const double cP = st.pf[3] * q.cz - st.pf[4] * q.sz, sP = st.pf[4] * q.cz + st.pf[3] * q.sz;
        st.g[0] = cP * q0 - sP * q1; st.g[1] = sP * q0 + cP * q1; st.g[2] = q2;
    };
    auto loop_quad = [&](int l) -> double {
        const LoopConst& q = sh.lc[l];
        const LoopState& st = sh.ls[cur][l];
        Pose2 a{st.pf[0], st.pf[1], st.pf[2], st.pf[3], st.pf[4]};
        Pose2 b{st.pt[0], st.pt[1], st.pt[2], st.pt[3], st.pt[4]};
        double wx, wy, wth;
        se2_apply_J(a, b, q.cz, q.sz, sh.lvec[l][0][0], sh.lvec[l][0][1], sh.lvec[l][0][2], sh.lvec[l][1][0],
                    sh.lvec[l][1][1], sh.lvec[l][1][2], wx, wy, wth);
        Sym3 om{q.om[0], q.om[1], q.om[2], q.om[3], q.om[4], q.om[5]};
        return om.quad(wx, wy, wth);
    };

    // One sweep over the chain.
    //   MODE 0: errors of the committed poses X            -> e, chi2, loop state buffer `bsel`
    //   MODE 1: trial poses X (+) (p b + q h), not stored  -> chi2, loop state buffer `bsel`, changed
    //   MODE 2: commit X <- X (+) (p b + q h)              -> X, e
    // big (wave-uniform): some angle moves by >= 2^-6 rad (full sincos instead of the small
    // rotation).  A run-time flag on purpose: as two instantiations under one branch, the compiler
    // hoists the common half of every slot above the branch and keeps it alive.
    bool sweepChanged = false;
    // Identical sub-expressions recur in every phase (P_j, kappa_j, dt_j, range masks ...); the
    // compiler would keep them alive per slot from one phase to the next.  Laundering the state
    // at the phase boundaries (no instructions) makes it recompute them instead.
    auto launder = [&]() {
#pragma unroll
        for (int s = 0; s < M; ++s) {
            asm volatile("" : "+v"(X[s].x), "+v"(X[s].y), "+v"(X[s].c), "+v"(X[s].s));
            if (KEEP_E) asm volatile("" : "+v"(ex[s]), "+v"(ey[s]), "+v"(eth[s]));
        }
    };
    // error of edge j (slot s) at the committed poses: from registers, or recomputed
    auto err_of = [&](int s, const Pose2& a, const SlotConst& K, double& e0, double& e1, double& e2) {
        if (KEEP_E) { e0 = ex[s]; e1 = ey[s]; e2 = eth[s]; return; }
        se2_error(a, X[s], K.tzx, K.tzy, K.cz, K.sz, K.thz, e0, e1, e2);
        if (!(j0 + s <= L)) { e0 = 0.0; e1 = 0.0; e2 = 0.0; }
    };
    // gn (wave-uniform): the step is exactly h (Gauss-Newton trial, p = 0, q = 1): X + (0 b + 1 h) and
    // X + h are the same bits, so the b operands and two operations per component are skipped.
    // wantChanged: only a steepest-descent trial needs to know whether any pose moved.
    auto sweep = [&](auto mode_c, bool big, double p, double q, int bsel, bool gn = false, bool wantChanged = true) -> double {
        constexpr int MODE = decltype(mode_c)::value;
        opaque();
        launder();
        // the commit sweep repeats the trial sweep's arithmetic; hide the coefficients so the
        // compiler cannot keep the trial's poses and errors alive to reuse them
        asm volatile("" : "+v"(p), "+v"(q));
        auto stepped = [&](int s) -> Pose2 {
            Pose2 Y;
            double dx, dy, dth;
            if (gn) { dx = hx[s]; dy = hy[s]; dth = hth[s]; }
            else { dx = fma(p, bx[s], q * hx[s]); dy = fma(p, by[s], q * hy[s]); dth = fma(p, bth[s], q * hth[s]); }
            Y.x = X[s].x + dx;
            Y.y = X[s].y + dy;
            Y.th = wrap_pi(X[s].th + dth);
            if (big) sincos_pi(Y.th, Y.s, Y.c);
            else rotate_small(X[s].c, X[s].s, Y.th - X[s].th, Y.c, Y.s);
            return Y;
        };
        const Pose2 last = MODE == 0 ? X[M - 1] : stepped(M - 1);
        Pose2 prev = last;
        if (MODE != 2 || KEEP_E) prev = prev0(last);  // (a commit without stored errors needs no neighbour)
        double part = 0.0;
        bool changed = false;
        constexpr int kWant = (MODE != 2 || KEEP_E) ? (MODE != 2 ? 7 : 3) : 0;
        SlotConst Kn;
        if (PF) ld_slot(0, Kn, IntC<kWant>{});
#pragma unroll
        for (int s = 0; s < M; ++s) {
            SlotConst K;
            if (PF) { K = Kn; if (s + 1 < M) ld_slot(s + 1, Kn, IntC<kWant>{}); }
            else ld_slot(s, K, IntC<kWant>{});
            const Pose2 Y = MODE == 0 ? X[s] : (s == M - 1 ? last : stepped(s));
            const bool v = j0 + s <= L;
            if (MODE == 1 && wantChanged) changed |= v && ((Y.x != X[s].x) || (Y.y != X[s].y) || (Y.th != X[s].th));
            double e0 = 0.0, e1 = 0.0, e2 = 0.0;
            if (MODE != 2 || KEEP_E) se2_error(prev, Y, K.tzx, K.tzy, K.cz, K.sz, K.thz, e0, e1, e2);
            if (MODE != 2) {
                const double c2 = K.om.quad(e0, e1, e2);
                part += v ? c2 : 0.0;
            }
            if (MODE != 1 && KEEP_E) { ex[s] = v ? e0 : 0.0; ey[s] = v ? e1 : 0.0; eth[s] = v ? e2 : 0.0; }
            if (MODE != 2 && ((ownSlots >> s) & 1u)) {
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    if (sf[l] == s && gl == of[l]) {
                        double* w = sh.ls[bsel][l].pf;
                        w[0] = Y.x; w[1] = Y.y; w[2] = Y.th; w[3] = Y.c; w[4] = Y.s;
                    }
                    if (st_[l] == s && gl == ot[l]) {
                        double* w = sh.ls[bsel][l].pt;
                        w[0] = Y.x; w[1] = Y.y; w[2] = Y.th; w[3] = Y.c; w[4] = Y.s;
                    }
                }
            }
            if (MODE == 2) X[s] = Y;
            prev = Y;
            if (MODE != 2) IPC_PIN1(part);
            if (MODE != 1 && KEEP_E) IPC_PIN3(ex[s], ey[s], eth[s]);
            if (MODE == 2) IPC_PIN3(X[s].x, X[s].y, X[s].th);
            IPC_PIN3(prev.x, prev.y, prev.th);
            IPC_PIN2(prev.c, prev.s);
            IPC_SLOT_FENCE();
        }
        if (MODE == 2) return 0.0;
        wave_sync();
        if constexpr (W > 1) {
            // one exchange: the partial sums of both waves; it also tells each wave that the
            // partner's loop end points are in place, so the loops are evaluated after it (by both
            // waves, same values; the first wave's lanes store the loop state)
            double tot, chg;
            cell_sum2(part, (MODE == 1 && changed) ? 1.0 : 0.0, tot, chg);
            const double lc = lane < NL ? loop_eval(lane, bsel) : 0.0;
#pragma unroll
            for (int l = 0; l < NL; ++l) tot += read_lane(lc, l);
            if (MODE == 1) sweepChanged = chg != 0.0;
            return tot;
        } else {
            if (lane < NL) part += loop_eval(lane, bsel);
            double tot, chg;
            cell_sum2(part, (MODE == 1 && changed) ? 1.0 : 0.0, tot, chg);
            if (MODE == 1) sweepChanged = chg != 0.0;
            return tot;
        }
    };

    // ---------------- initial errors (consensus_utils.cpp:11) ----------------
    int evals = 1;
    double currentChi = sweep(IntC<0>{}, false, 0.0, 0.0, cur);

    // ---------------- dog-leg (g2o OptimizationAlgorithmDogleg::solve) ----------------
    double delta = 1e4;
    const int maxTrials = 100;
    int it_done = 0, tries_total = 0, flags = 0;

    IPC_WTICK(tmT)
    for (int it = 0; it < iterations; ++it) {
        // ---- phase A: forces g, hand-back m -> b ----
        opaque();
        launder();
        if (gl < NL) loop_force(gl);
        wave_sync();
        const Pose2 a0 = prev0(X[M - 1]);             // (also orders the loop forces before their readers)
        constexpr int kWantA = 2 | 4 | kErrWant;
        auto force = [&](int s, const Pose2& a, const SlotConst& K, double& gx, double& gy, double& gth, double& mx,
                         double& my, double& mth) {
            const double cz = K.cz, sz = K.sz;
            const double cP = a.c * cz - a.s * sz, sP = a.s * cz + a.c * sz;
            double e0, e1, e2, qx, qy, qth;
            err_of(s, a, K, e0, e1, e2);                                // zero on idle slots
            K.om.mul(e0, e1, e2, qx, qy, qth);
            gx = cP * qx - sP * qy;
            gy = sP * qx + cP * qy;
            gth = qth;
            const double dx = X[s].x - a.x, dy = X[s].y - a.y;
            mx = gx; my = gy;
            mth = gth + (-dy * gx + dx * gy);
        };
        double bbp = 0.0;
        {
            double g0x, g0y, g0th, m0x, m0y, m0th;
            SlotConst Kn;
            {
                SlotConst K0;
                ld_slot(0, K0, IntC<kWantA>{});
                if (PF && M > 1) ld_slot(M - 1, Kn, IntC<kWantA>{});
                force(0, a0, K0, g0x, g0y, g0th, m0x, m0y, m0th);
            }
            double nx, ny, nth;
            next3(m0x, m0y, m0th, nx, ny, nth);
#pragma unroll
            for (int s = M - 1; s >= 0; --s) {
                double gx, gy, gth, mx, my, mth;
                if (s == 0) { gx = g0x; gy = g0y; gth = g0th; mx = m0x; my = m0y; mth = m0th; }
                else {
                    SlotConst K;
                    if (PF) { K = Kn; if (s - 1 > 0) ld_slot(s - 1, Kn, IntC<kWantA>{}); }
                    else ld_slot(s, K, IntC<kWantA>{});
                    force(s, X[s - 1], K, gx, gy, gth, mx, my, mth);
                }
                const bool v = j0 + s <= L;
                double tx = nx - gx, ty = ny - gy, tth = nth - gth;
                if ((ownSlots >> s) & 1u) {
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    const LoopState& st = sh.ls[cur][l];
                    if (st_[l] == s && gl == ot[l]) { tx -= st.g[0]; ty -= st.g[1]; tth -= st.g[2]; }
                    if (sf[l] == s && gl == of[l]) {
                        const double dx = st.pt[0] - st.pf[0], dy = st.pt[1] - st.pf[1];
                        tx += st.g[0]; ty += st.g[1];
                        tth += st.g[2] + (-dy * st.g[0] + dx * st.g[1]);
                    }
                }
                }
                bx[s] = v ? tx : 0.0; by[s] = v ? ty : 0.0; bth[s] = v ? tth : 0.0;
                bbp += bx[s] * bx[s] + by[s] * by[s] + bth[s] * bth[s];
                nx = mx; ny = my; nth = mth;
                IPC_PIN3(bx[s], by[s], bth[s]);
                IPC_PIN3(nx, ny, nth);
                IPC_PIN1(bbp);
                IPC_SLOT_FENCE();
            }
        }
        IPC_WTICK(tmA)
        // ---- phase B: b^T b, b^T H b, capacitance partials, solve ----
        double bb, bHb, alpha, hsdNorm;
        double nu[NL][3];
        opaque();
        launder();
        {
            // b at the loop end points (the gauge end point carries zero)
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                if (gl == 0) {
                    if (lf[l] == 0) { sh.lvec[l][0][0] = 0.0; sh.lvec[l][0][1] = 0.0; sh.lvec[l][0][2] = 0.0; }
                    if (lt[l] == 0) { sh.lvec[l][1][0] = 0.0; sh.lvec[l][1][1] = 0.0; sh.lvec[l][1][2] = 0.0; }
                }
#pragma unroll
                for (int s = 0; s < M; ++s) {
                    if (!((ownSlots >> s) & 1u)) continue;
                    if (sf[l] == s && gl == of[l]) { sh.lvec[l][0][0] = bx[s]; sh.lvec[l][0][1] = by[s]; sh.lvec[l][0][2] = bth[s]; }
                    if (st_[l] == s && gl == ot[l]) { sh.lvec[l][1][0] = bx[s]; sh.lvec[l][1][1] = by[s]; sh.lvec[l][1][2] = bth[s]; }
                }
            }
            // group 1: b^T b, b^T H b, W_1 (3), M_11 (6); group 2: W_2 (3), M_22 (6), M_12 (6)
            double v1[16], v2[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) { v1[k] = 0.0; v2[k] = 0.0; }
            v1[0] = bbp;
            double qbx, qby, qbth;
            prev3(bx[M - 1], by[M - 1], bth[M - 1], qbx, qby, qbth);
            constexpr int kWantB = 2 | 4 | 8 | kErrWant;
            SlotConst Kn;
            if (PF) ld_slot(0, Kn, IntC<kWantB>{});
#pragma unroll
            for (int s = 0; s < M; ++s) {
                SlotConst K;
                if (PF) { K = Kn; if (s + 1 < M) ld_slot(s + 1, Kn, IntC<kWantB>{}); }
                else ld_slot(s, K, IntC<kWantB>{});
                const Pose2& a = s == 0 ? a0 : X[s - 1];
                const bool v = j0 + s <= L;
                const double cz = K.cz, sz = K.sz;
                double wx, wy, wth;
                se2_apply_J(a, X[s], cz, sz, qbx, qby, qbth, bx[s], by[s], bth[s], wx, wy, wth);
                const double hq = K.om.quad(wx, wy, wth);
                v1[1] += v ? hq : 0.0;
                const Sym3 sg = K.sg;
                const double c = a.c * cz - a.s * sz, sn = a.s * cz + a.c * sz;
                const double kx = -(X[s].y - gauge.y), ky = X[s].x - gauge.x;
                const double cc = c * c, ss = sn * sn, cs = c * sn;
                const double C00 = cc * sg.a00 - 2 * cs * sg.a01 + ss * sg.a11;
                const double C01 = cs * (sg.a00 - sg.a11) + (cc - ss) * sg.a01;
                const double C11 = ss * sg.a00 + 2 * cs * sg.a01 + cc * sg.a11;
                const double c0 = c * sg.a02 - sn * sg.a12, c1 = sn * sg.a02 + c * sg.a12;
                const double sth = sg.a22;
                double psi[6];
                psi[2] = c0 - sth * kx;
                psi[4] = c1 - sth * ky;
                psi[0] = C00 - kx * c0 - kx * psi[2];
                psi[1] = C01 - kx * c1 - ky * psi[2];
                psi[3] = C11 - ky * c1 - ky * psi[4];
                psi[5] = sth;
                double e0, e1, e2;
                err_of(s, a, K, e0, e1, e2);
                const double w0 = c * e0 - sn * e1 - kx * e2;
                const double w1 = sn * e0 + c * e1 - ky * e2;
                const double w2 = e2;
                const int j = j0 + s;
                const double m1 = (v && j > llo[0] && j <= lhi[0]) ? 1.0 : 0.0;
                v1[2] += m1 * w0; v1[3] += m1 * w1; v1[4] += m1 * w2;
#pragma unroll
                for (int k = 0; k < 6; ++k) v1[5 + k] += m1 * psi[k];
                if constexpr (NL == 2) {
                    const double m2 = (v && j > llo[1] && j <= lhi[1]) ? 1.0 : 0.0;
                    const double m12 = m1 * m2;
                    v2[0] += m2 * w0; v2[1] += m2 * w1; v2[2] += m2 * w2;
#pragma unroll
                    for (int k = 0; k < 6; ++k) { v2[3 + k] += m2 * psi[k]; v2[9 + k] += m12 * psi[k]; }
                }
                qbx = bx[s]; qby = by[s]; qbth = bth[s];
                asm volatile("" : "+v"(v1[0]), "+v"(v1[1]), "+v"(v1[2]), "+v"(v1[3]), "+v"(v1[4]), "+v"(v1[5]),
                             "+v"(v1[6]), "+v"(v1[7]), "+v"(v1[8]), "+v"(v1[9]), "+v"(v1[10]) : : "memory");
                if constexpr (NL == 2)
                    asm volatile("" : "+v"(v2[0]), "+v"(v2[1]), "+v"(v2[2]), "+v"(v2[3]), "+v"(v2[4]), "+v"(v2[5]),
                                 "+v"(v2[6]), "+v"(v2[7]), "+v"(v2[8]), "+v"(v2[9]), "+v"(v2[10]), "+v"(v2[11]),
                                 "+v"(v2[12]), "+v"(v2[13]), "+v"(v2[14]) : : "memory");
                IPC_SLOT_FENCE();
            }
            wave_sum16_store(v1, &sh.red[wsub][0]);
            if constexpr (NL == 2) wave_sum16_store(v2, &sh.red[wsub][16]);
            wave_sync();
            pair_barrier();                           // both waves' partial sums and end-point vectors are in place
            IPC_WTICK(tmB1)
            // ---- capacitance solve, lane-parallel (a uniform 6x6 solve in registers would pin ~150
            // VGPRs while the whole chain state is live):
            //   lane l*9+i*3+a   : Gamma_l[i][a]                              -> gam[.]
            //   lane r*(NS+1)+c  : S[r][c] (c < NS) / rhs d[r] (c == NS), then Gauss-Jordan in place
            auto wtv = [&](int k) -> double {
                double t = sh.red[0][k];
#pragma unroll
                for (int o = 1; o < W; ++o) t += sh.red[o][k];
                return t;
            };
            double* gamw = sh.gam[wsub];
            if (lane < NL * 9) {
                const int l = lane / 9, i = (lane % 9) / 3, a = lane % 3;
                const LoopConst& q = sh.lc[l];
                const LoopState& st = sh.ls[cur][l];
                const double Aq = st.pf[3] * q.cz - st.pf[4] * q.sz, Bq = st.pf[4] * q.cz + st.pf[3] * q.sz;
                const double Kx = -(st.pt[1] - gauge.y), Ky = st.pt[0] - gauge.x;
                double g;
                if (i == 0) g = a == 0 ? Aq : (a == 1 ? Bq : Aq * Kx + Bq * Ky);
                else if (i == 1) g = a == 0 ? -Bq : (a == 1 ? Aq : -Bq * Kx + Aq * Ky);
                else g = a == 2 ? 1.0 : 0.0;
                gamw[lane] = q.sigma * g;
            }
            wave_sync();
            constexpr int RS = NS + 1;                    // row stride of the augmented system
            double val = 0.0;
            const int r = lane / RS, c = lane % RS;
            if (lane < NS * RS) {
                const int l1 = r / 3, i = r % 3;
                const double g10 = gamw[l1 * 9 + i * 3], g11 = gamw[l1 * 9 + i * 3 + 1], g12 = gamw[l1 * 9 + i * 3 + 2];
                if (c < NS) {
                    const int l2 = c / 3, k = c % 3;
                    const int mb = l1 == l2 ? (l1 == 0 ? 5 : 19) : 25;
                    const double m00 = wtv(mb), m01 = wtv(mb + 1), m02 = wtv(mb + 2), m11 = wtv(mb + 3), m12 = wtv(mb + 4), m22 = wtv(mb + 5);
                    const double t0 = g10 * m00 + g11 * m01 + g12 * m02;
                    const double t1 = g10 * m01 + g11 * m11 + g12 * m12;
                    const double t2 = g10 * m02 + g11 * m12 + g12 * m22;
                    val = t0 * gamw[l2 * 9 + k * 3] + t1 * gamw[l2 * 9 + k * 3 + 1] + t2 * gamw[l2 * 9 + k * 3 + 2];
                    if (l1 == l2) {
                        const int lo_ = i < k ? i : k, hi_ = i < k ? k : i;
                        val += sh.lc[l1].sg[lo_ * 3 - lo_ * (lo_ - 1) / 2 + (hi_ - lo_)];
                    }
                } else {
                    const int wb = l1 == 0 ? 2 : 16;
                    val = sh.ls[cur][l1].e[i] - (g10 * wtv(wb) + g11 * wtv(wb + 1) + g12 * wtv(wb + 2));
                }
            }
            bool okS = true;
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const double piv = read_lane(val, k * RS + k);
                okS = okS && (piv > 0);
                double inv = __builtin_amdgcn_rcp(piv);
                inv = fma(fma(-piv, inv, 1.0), inv, inv);
                inv = fma(fma(-piv, inv, 1.0), inv, inv);
                const double rowk = __shfl(val, k * RS + c, 64);
                const double colk = __shfl(val, r * RS + k, 64);
                val = (r == k) ? rowk * inv : fma(-(colk * rowk), inv, val);
            }
            {   // nu_l[cc] = sum_rr Gamma_l[rr][cc] mu_{3l+rr}; mu_r sits in lane r*RS + NS
                const int l = (lane < NS) ? lane / 3 : 0, cc = lane % 3;
                double nv = 0.0;
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) {
                    const double mu_r = __shfl(val, (3 * l + rr) * RS + NS, 64);
                    nv += gamw[l * 9 + rr * 3 + cc] * mu_r;
                }
#pragma unroll
                for (int l2 = 0; l2 < NL; ++l2)
#pragma unroll
                    for (int k = 0; k < 3; ++k) nu[l2][k] = read_lane(nv, 3 * l2 + k);
            }
            bb = wtv(0);
            bHb = wtv(1);
            {
                const double lq = lane < NL ? loop_quad(lane) : 0.0;
#pragma unroll
                for (int l = 0; l < NL; ++l) bHb += read_lane(lq, l);
            }
            if (!okS) { flags |= 2; break; }
            alpha = bb / bHb;
            hsdNorm = sqrt(alpha * alpha * bb);
        }
        IPC_WTICK(tmB2)
        // ---- phase C: u, rho, prefix sums -> h_gn; |h|^2, b.h (h^T H h = b.h) ----
        double hgnNorm, bh, hHh;
        opaque();
        launder();
        {
            // rho into (hx, hy, hth); theta prefix in-lane
            double run = 0.0;
            constexpr int kWantC = 2 | 8 | kErrWant;
            SlotConst Kn;
            if (PF) ld_slot(0, Kn, IntC<kWantC>{});
#pragma unroll
            for (int s = 0; s < M; ++s) {
                SlotConst K;
                if (PF) { K = Kn; if (s + 1 < M) ld_slot(s + 1, Kn, IntC<kWantC>{}); }
                else ld_slot(s, K, IntC<kWantC>{});
                const Pose2& a = s == 0 ? a0 : X[s - 1];
                const int j = j0 + s;
                const bool v = j <= L;
                const double m1 = (v && j > llo[0] && j <= lhi[0]) ? 1.0 : 0.0;
                double n0 = m1 * nu[0][0], n1 = m1 * nu[0][1], n2 = m1 * nu[0][2];
                if constexpr (NL == 2) {
                    const double m2 = (v && j > llo[1] && j <= lhi[1]) ? 1.0 : 0.0;
                    n0 += m2 * nu[1][0]; n1 += m2 * nu[1][1]; n2 += m2 * nu[1][2];
                }
                const double cz = K.cz, sz = K.sz;
                const double c = a.c * cz - a.s * sz, sn = a.s * cz + a.c * sz;
                const double kx = -(X[s].y - gauge.y), ky = X[s].x - gauge.x;
                const double wx = c * n0 + sn * n1, wy = -sn * n0 + c * n1, wth = -(kx * n0 + ky * n1) + n2;
                double vx, vy, vth;
                K.sg.mul(wx, wy, wth, vx, vy, vth);
                double e0, e1, e2;
                err_of(s, a, K, e0, e1, e2);
                const double ux = -vx - e0, uy = -vy - e1, uth = -vth - e2;
                hx[s] = v ? c * ux - sn * uy : 0.0;
                hy[s] = v ? sn * ux + c * uy : 0.0;
                run += v ? uth : 0.0;
                hth[s] = run;                        // in-lane inclusive prefix of rho_theta
                IPC_PIN3(hx[s], hy[s], hth[s]);
                IPC_SLOT_FENCE();
            }
            const double offT = cell_excl(run);                        // h_theta of the lane's predecessor pose
            double rx = 0.0, ry = 0.0, thPrev = offT;
#pragma unroll
            for (int s = 0; s < M; ++s) {
                const Pose2& a = s == 0 ? a0 : X[s - 1];
                const bool v = j0 + s <= L;
                hth[s] += offT;
                const double dx = X[s].x - a.x, dy = X[s].y - a.y;
                rx += v ? hx[s] - dy * thPrev : 0.0;
                ry += v ? hy[s] + dx * thPrev : 0.0;
                hx[s] = rx; hy[s] = ry;
                thPrev = hth[s];
                IPC_PIN3(hx[s], hy[s], hth[s]);
                IPC_SLOT_FENCE();
            }
            double offX, offY;
            cell_excl2(rx, ry, offX, offY);
            double p0 = 0.0, p1 = 0.0;
#pragma unroll
            for (int s = 0; s < M; ++s) {
                const bool v = j0 + s <= L;
                hx[s] = v ? hx[s] + offX : 0.0;
                hy[s] = v ? hy[s] + offY : 0.0;
                hth[s] = v ? hth[s] : 0.0;
                p0 += hx[s] * hx[s] + hy[s] * hy[s] + hth[s] * hth[s];
                p1 += bx[s] * hx[s] + by[s] * hy[s] + bth[s] * hth[s];
            }
            double hh;
            cell_sum2(p0, p1, hh, bh);
            hgnNorm = sqrt(hh);
            hHh = bh;
        }
        IPC_WTICK(tmC)
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
            int stepType;                             // 0 GN, 1 SD, 2 DL
            double beta = 0.0, sdScale = 0.0;
            if (hgnNorm < delta) stepType = 0;
            else if (hsdNorm > delta) { stepType = 1; sdScale = delta / hsdNorm; }
            else {
                stepType = 2;
                // c = hsd.(hgn-hsd) and |hgn-hsd|^2 do not depend on delta: reduced once per iteration, reused by
                // every later dog-leg trial of the same iteration (same bits)
                if (!dlReady) {
                    double p0 = 0.0, p1 = 0.0;
#pragma unroll
                    for (int s = 0; s < M; ++s) {
                        const double sx = alpha * bx[s], sy = alpha * by[s], sth = alpha * bth[s];
                        const double ax = hx[s] - sx, ay = hy[s] - sy, ath = hth[s] - sth;
                        p0 += sx * ax + sy * ay + sth * ath;
                        p1 += ax * ax + ay * ay + ath * ath;
                    }
                    cell_sum2(p0, p1, dlC, dlBma);
                    dlReady = true;
                }
                const double c = dlC, bma = dlBma;
                const double hsdSq = alpha * alpha * bb;
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
            bool big = false;
            double pc2 = pcoef, qc2 = qcoef;
            asm volatile("" : "+v"(pc2), "+v"(qc2));
#pragma unroll
            for (int s = 0; s < M; ++s) {
                const double thn = wrap_pi(X[s].th + (stepType == 0 ? hth[s] : fma(pc2, bth[s], qc2 * hth[s])));
                big |= fabs(thn - X[s].th) >= 0.015625;
            }
            const bool anyBig = __ballot(big) != 0ull;
            const int trial = cur ^ 1;
#ifdef IPC_PHASE_TIMING
            const unsigned long long tb0 = __builtin_amdgcn_s_memtime();
#endif
            const double newChi = sweep(IntC<1>{}, anyBig, pcoef, qcoef, trial, stepType == 0, stepType == 1);
#ifdef IPC_PHASE_TIMING
            if (anyBig) { ++nBig; tmBig += __builtin_amdgcn_s_memtime() - tb0; } else { tmSmall += __builtin_amdgcn_s_memtime() - tb0; }
#endif
            const bool anyChanged = stepType == 1 ? sweepChanged : true;
            ++evals;
            const double nonLinearGain = currentChi - newChi;
#ifdef IPC_PHASE_TIMING
            if (stepType == 0) { ++gnEvals; if (!(linearGain > 0 ? nonLinearGain > 0 : nonLinearGain < 0)) ++gnRejected; }
#endif
            if (fabs(linearGain) < 1e-12) linearGain = 1e-12;
            const bool linPos = linearGain > 0;
            auto rho_gt = [&](double t) { return linPos ? nonLinearGain > t * linearGain : nonLinearGain < t * linearGain; };
            auto rho_lt = [&](double t) { return linPos ? nonLinearGain < t * linearGain : nonLinearGain > t * linearGain; };
            if (rho_gt(0.0)) {                        // rho > 0, discardTop: commit the trial
                goodStep = true;
                currentChi = newChi;
                cur = trial;
                IPC_WTICK(tmT)
                sweep(IntC<2>{}, anyBig, pcoef, qcoef, trial, stepType == 0, false);
                IPC_WTICK(tmK)
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
        IPC_WTICK(tmT)
        if (numTries == maxTrials || !goodStep) { flags |= 1; break; }
    }
#ifdef IPC_PHASE_TIMING
    if (lane == 0 && wsub == 0 && P.dbg) {
        unsigned long long* d = reinterpret_cast<unsigned long long*>(P.dbg) + 64 + 16 * (M + 16 * (W - 1));
        atomicAdd(d + 0, tmA); atomicAdd(d + 1, tmB1); atomicAdd(d + 2, tmB2); atomicAdd(d + 3, tmC);
        atomicAdd(d + 4, tmT); atomicAdd(d + 5, tmW);
        atomicAdd(d + 6, (unsigned long long)it_done); atomicAdd(d + 7, (unsigned long long)evals);
        atomicAdd(d + 8, (unsigned long long)it_done * (unsigned long long)L);
        atomicAdd(d + 9, 1ull); atomicAdd(d + 10, tmK); atomicAdd(d + 11, (unsigned long long)(evals - 1 - it_done)); atomicAdd(d + 12, gnEvals); atomicAdd(d + 13, gnRejected); atomicAdd(d + 14, nBig); atomicAdd(d + 15, tmBig); atomicAdd(d + 1024, tmSmall);
    }
#endif

    // ---- per-edge chi2 (consensus_utils.cpp:15-19) ----
    double mx = 0.0;
    bool nan = false;
    opaque();
    {
        const Pose2 a0 = prev0(X[M - 1]);
#pragma unroll
        for (int s = 0; s < M; ++s) {
            SlotConst K;
            ld_slot(s, K, IntC<(4 | kErrWant)>{});
            double e0, e1, e2;
            err_of(s, s == 0 ? a0 : X[s - 1], K, e0, e1, e2);
            if (j0 + s > L) continue;
            const double c = K.om.quad(e0, e1, e2);
            if (c != c) nan = true;
            else mx = fmax(mx, c);
        }
    }
    mx = wave_max(mx);
    nan = __ballot(nan) != 0ull;
    if constexpr (W > 1) {
        double a[8] = {mx, nan ? 1.0 : 0.0, 0, 0, 0, 0, 0, 0};
        post_wait(a, 2);
#pragma unroll
        for (int o = 0; o < W; ++o) {
            if (o == wsub) continue;
            mx = fmax(mx, peer(o, 0));
            nan = nan || peer(o, 1) != 0.0;
        }
    }
    wave_sync();
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
    if (seq_out) *seq_out = seq;
}

}  // namespace ipc
