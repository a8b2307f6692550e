// Device-resident cluster solve: the whole dog-leg of one cluster problem (IPC::agreementCheck's
// isAgreeingWithCurrentState, reference src/consensus_utils.cpp:7-22, or the final map,
// src/simulation.cpp:50-65) in ONE persistent launch, no host round trip until the result record is written.
//
// Same arithmetic as the one-kernel-per-phase solvers (cluster_se2.hpp / cluster_se3.hpp, whose per-index bodies
// `gk*_at` it calls, and dense_chol.hpp, whose factorisation it re-orchestrates): the reductions add in the same
// order, so the two paths agree bit for bit -- tests/test_gpu_persistent.py holds them against each other.
//
//   workgroup 0 ("leader", 512 threads) runs the dog-leg control flow (g2o OptimizationAlgorithmDogleg::solve as
//     restated in cluster_common.hpp::cluster_dogleg) and every chain phase: a phase is a loop over the poses /
//     loops followed by __syncthreads(), its scalars are reduced in LDS -- no launch, no barrier across CUs;
//   workgroups 1 .. G-1 ("helpers") sleep on a grid barrier until the leader asks for a factorisation, then share
//     the assembly of the capacitance matrix and its blocked Cholesky (64 x 64 tiles, two per workgroup at a
//     time).  The leader runs the critical path of the factorisation one step ahead: while the helpers apply block
//     column k to the trailing matrix, it brings the next diagonal block up to date itself and factors it, so the
//     32 dependent pivots of a block column never wait for the bulk update.
//
// Data that crosses workgroups inside the launch (capacitance matrix, factor, pivots) moves with sc1 loads / stores
// (ld_shared / st_shared) and needs no cache maintenance; the leader's chain arrays that the assembly reads (prefix
// sums, Gamma_l, loop errors) are published once per iteration with an agent-scope release / acquire pair.
// G is sized by the capacitance system (1 for systems of one tile: then there is no cross-workgroup traffic at all).
#pragma once
#include <mutex>

#include "cluster_se2.hpp"
#include "cluster_se3.hpp"

namespace ipc {

constexpr int kPT = 512;                     // threads per workgroup: 8 waves = 2 per SIMD = 256 VGPRs per lane, which the panel solve and
                                             // the tile update need (768 threads / 168 VGPRs: the leader's phases take two passes instead of three
                                             // over a 1 500-pose chain, but the panel solve spills and a block column costs 2.5x -- measured, dropped)
constexpr int kPSG = kPT / 256;              // 256-thread sub-groups (one 64 x 64 tile each)
constexpr unsigned kSpinLimit = 1u << 21;    // polls before a barrier gives up (seconds): a lost workgroup must not hang the GPU

struct PersistCtl {                          // device memory, zeroed before every launch
    unsigned bar;                            // grid barrier: monotonic arrival counter
    int cmd;                                 // leader -> helpers: 1 = factor, 2 = exit
    int le_sel;                              // 1: the committed loop errors sit in the second buffer (commit swaps them)
    int error;                               // 1: a barrier timed out
    unsigned team_bar[2];                    // band kernel, split factorisation: the two halves' own barriers
    unsigned long long last_start;           // IPC_PERSIST_PROF: (clock << 8 | block) of the workgroup that started last
};

struct PersistOut {                          // result record (device, copied to pinned host memory behind the kernel)
    double max_chi2, chi2_total, chi2_initial;
    int iterations, tries, flags, evals;
    int x_sel;                               // 1: the optimised poses sit in the second pose buffer
    int error;                               // 1: a barrier timed out, 2: aborted by the host (speculative solve no longer needed)
    int device_ticks;                        // the leader's wall clock from its first to its last instruction, 100 MHz ticks
    int pad1;
};

struct PersistArgs {
    const double* src; int src_ld;           // committed poses of the whole trajectory (global indexing)
    int iterations; double term_eps;
    PersistCtl* ctl; PersistOut* out;
    double* dinv;                            // [n] reciprocal pivots of the factor
    unsigned long long* prof;                // optional [16] phase clocks of the leader (IPC_PERSIST_PROF=1), 100 MHz ticks
    const int* abort_word; int launch_id;    // optional: host-mapped word; the solve gives up when it has reached launch_id (ids only grow)
};

// ---- LDS carve-up (doubles) -------------------------------------------------------------------------------
constexpr int kLdsD = 0;                                   // [32][33] diagonal block of the current block column
constexpr int kLdsDinv = kCB * (kCB + 1);                  // [32]
constexpr int kLdsR = kLdsDinv + kCB;                      // phase-private region
constexpr int kLdsPanel = kCB * (64 + 1);                  // one [32][65] panel
constexpr int kLdsTotal = kLdsR + kPSG * 2 * kLdsPanel;    // 17 728 doubles = 141 824 bytes
// leader's use of the region: reduction staging [kLeadMaxBlk runs][4 waves][2], results [8], scan partials [9][32]
constexpr int kLeadMaxBlk = 700;            // runs of 256 indices the reduction staging holds (L + nl < 179 200)
constexpr int kLdsRed = kLdsR, kLdsRes = kLdsRed + kLeadMaxBlk * 4 * 2, kLdsWsum = kLdsRes + 8, kLdsMisc = kLdsWsum + 9 * 32;
static_assert(kLdsMisc + 64 <= kLdsTotal, "LDS carve-up");

struct GridBar { unsigned* ctr; unsigned target; int G; int* error; unsigned long long* prof; };

// phase clocks (thread 0 of workgroup 0 only; prof == nullptr: off)
enum { kProfTotal = 0, kProfPre, kProfHandoff, kProfAssemble, kProfFactor, kProfFactorWork, kProfFactorWait, kProfBacksolve,
       kProfPost, kProfTrial, kProfRest, kProfIterations, kProfSteps,
       kProfHelpDT, kProfHelpSolve, kProfHelpUpdate, kProfHelpWait, kProfLookLoad, kProfLookSolve, kProfLookFill, kProfLookPotrf, kProfLookPub, kProfBsDots, kProfBsPrefetch, kProfBsSync, kProfBsTri,
       kProfStartSkew, kProfStartSkewXcd01, kProfStartLaunches,             // band kernel: workgroup 0 start -> start of the LAST workgroup of the launch; the part of it in launches where that is over 1 ms (their number: "handoff"); launches
       kProfTileBlock, kProfTileSelect, kProfTileTrsm, kProfTileStore,      // band kernel, first tile workgroup: until the column's block has arrived / operands selected / panel solve / stores drained
       kProfN };
__device__ __forceinline__ unsigned long long prof_now() { return wall_clock64(); }
__device__ __forceinline__ void prof_add(unsigned long long* prof, int slot, unsigned long long t0)
{
    if (prof && threadIdx.x == 0 && blockIdx.x == 0) prof[slot] += prof_now() - t0;
}
// the first helper's clocks (slots kProfHelp*)
__device__ __forceinline__ void prof_add1(unsigned long long* prof, int slot, unsigned long long t0)
{
    if (prof && threadIdx.x == 0 && blockIdx.x == 1) prof[slot] += prof_now() - t0;
}

// Every workgroup arrives once; leaves when all G have.  Monotonic counter, relaxed sc1 poll with s_sleep.  The
// caller's cross-workgroup stores are sc1 (write-through) and are drained by the s_waitcnt before the arrival.
__device__ __forceinline__ bool grid_barrier(GridBar& gb)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (gb.G == 1) return true;
    gb.target += (unsigned)gb.G;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(gb.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (a few fast polls for the barriers inside the factorisation, then long naps: the helpers sit here through the
        // leader's chain phases, and a chip full of pollers -- several solves in flight -- starves the memory fabric)
        unsigned spins = 0;
        while (__hip_atomic_load(gb.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gb.target) {
            if (spins < 48) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(64);
            if (++spins > kSpinLimit) { __hip_atomic_store(gb.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
    }
    __syncthreads();
    return __hip_atomic_load(gb.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
}

// ---- leader phases ----------------------------------------------------------------------------------------
template <class F>
__device__ __forceinline__ void lead_for(int n, F f)
{
    for (int i = threadIdx.x; i < n; i += kPT) f(i);
    __syncthreads();
}

// Sum of K per-index values over indices 0 .. nblk*256-1, added exactly as gk_block_reduce_store + gk_sum add them:
// a DPP scan inside each run of 64 indices, the four wave totals of a run of 256 in order, the runs in ascending order.
// One __syncthreads() per phase (plus one before the staging area is reused).
template <int K, class F>
__device__ __forceinline__ void lead_reduce(int nblk, double* lds, double (&tot)[K], F f)
{
    double* wpart = lds + kLdsRed;                            // [nblk][4][K]
    const int tid = threadIdx.x, q = tid >> 8, t = tid & 255;
    for (int vb0 = 0; vb0 < nblk; vb0 += kPSG) {
        const int vb = vb0 + q;
        double v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = 0.0;
        if (vb < nblk) f(vb * 256 + t, v);
        double ws[K];
#pragma unroll
        for (int k = 0; k < K; ++k) ws[k] = wave_sum(v[k]);
        if (vb < nblk) {                                      // (wave-uniform; every lane stores the same value)
#pragma unroll
            for (int k = 0; k < K; ++k) wpart[(vb * 4 + (t >> 6)) * K + k] = ws[k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double acc = 0.0;
        for (int vb = 0; vb < nblk; ++vb) {
            const double* w = wpart + (size_t)vb * 4 * K + k;
            acc += ((w[0] + w[K]) + w[2 * K]) + w[3 * K];
        }
        tot[k] = acc;
    }
    __syncthreads();
}

// In-place inclusive prefix sums over indices 1..L of K arrays (row length ld), added as gk_scan adds them: runs of
// 1024 indices = 16 wave scans whose totals are added in wave order (here: kPT / 64 waves, 1024 / kPT passes per run).
template <int K>
__device__ __forceinline__ void lead_scan_k(double* arr, int L, int ld, double* lds)
{
    constexpr int NP = (1024 + kPT - 1) / kPT;               // passes of the workgroup over a run of 1024 indices
    double* wsum = lds + kLdsWsum;
    const int tid = threadIdx.x, lane = tid & 63;
    double carry[K];
#pragma unroll
    for (int k = 0; k < K; ++k) carry[k] = 0.0;
    for (int base = 1; base <= L; base += 1024) {
        double v[NP][K];
#pragma unroll
        for (int hp = 0; hp < NP; ++hp) {
            const int off = hp * kPT + tid;                   // position in the run; its wave of 64 is off >> 6
            const int i = base + off;
            const bool in = off < 1024 && i <= L;
            const int ic = in ? i : 0;                        // (index 0 is a valid, unused slot of every row: no predicated loads)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const double a = gptr(arr)[(size_t)k * ld + ic];
                v[hp][k] = wave_inclusive_scan(in ? a : 0.0);
                wsum[k * 32 + (off >> 6)] = read_lane(v[hp][k], 63);      // (uniform value, every lane stores it; slots >= 16 are never summed)
            }
        }
        __syncthreads();
        double res[NP][K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double tt = carry[k];
            for (int w = 0; w < 16; ++w) tt += wsum[k * 32 + w];
#pragma unroll
            for (int hp = 0; hp < NP; ++hp) {
                const int vw = (hp * kPT + tid) >> 6;
                double off = carry[k];
                for (int w = 0; w < vw && w < 16; ++w) off += wsum[k * 32 + w];
                res[hp][k] = v[hp][k] + off;
            }
            carry[k] = tt;
        }
#pragma unroll
        for (int hp = 0; hp < NP; ++hp) {
            const int off = hp * kPT + tid, i = base + off;
            if (off < 1024 && i <= L) {                       // (one branch around the K stores)
#pragma unroll
                for (int k = 0; k < K; ++k) gptr(arr)[(size_t)k * ld + i] = res[hp][k];
            }
        }
        __syncthreads();
    }
}
__device__ __forceinline__ void lead_scan(double* arr, int K, int L, int ld, double* lds)
{
    while (K >= 9) { lead_scan_k<9>(arr, L, ld, lds); arr += 9 * (size_t)ld; K -= 9; }
    while (K >= 3) { lead_scan_k<3>(arr, L, ld, lds); arr += 3 * (size_t)ld; K -= 3; }
    while (K >= 1) { lead_scan_k<1>(arr, L, ld, lds); arr += (size_t)ld; K -= 1; }
}

// ---- blocked Cholesky across the workgroups of the launch (dense_chol.hpp's arithmetic) ---------------------
// first wave: factor the 32 x 32 block in Dn (identity padded), reciprocal pivots to Dninv; false on a non-positive pivot
__device__ __noinline__ bool potrf32_wave(double (*Dn)[kCB + 1], double* Dninv)
{
    const int lane = threadIdx.x & 63, r = lane & 31;
    bool ok = true;
    double row[kCB];
#pragma unroll
    for (int c = 0; c < kCB; ++c) row[c] = Dn[r][c];
    // Lane r holds row r; column c reaches the other rows through v_readlane.  The 32 pivots are one dependent
    // chain (pivot -> rsq -> two Newton steps -> column -> next pivot); the rank-one update of the columns beyond
    // c + 1 does not feed it, so it is written BEHIND the start of the next pivot's chain and fills its latency.
    double piv = read_lane(row[0], 0);
    if (!(piv > 0)) ok = false;
    double inv = rsqrt_newton(piv);
#pragma unroll
    for (int c = 0; c < kCB; ++c) {
        // (lane c: row[c] IS the pivot, so one product serves the diagonal and the column; rows above pick up
        // garbage that nothing reads: lrc is forced to 0 for r < c, only the lower triangle is written back)
        const double lrc = r >= c ? row[c] * inv : 0.0;
        row[c] = lrc;
        Dninv[c] = inv;                                       // (every lane stores the same value: no exec juggling)
        if (c + 1 < kCB) {
            row[c + 1] = fma(-lrc, read_lane(lrc, c + 1), row[c + 1]);
            piv = read_lane(row[c + 1], c + 1);
            if (!(piv > 0)) ok = false;
            inv = rsqrt_newton(piv);
        }
#pragma unroll
        for (int cc = c + 2; cc < kCB; ++cc) row[cc] = fma(-lrc, read_lane(lrc, cc), row[cc]);
    }
    if (lane < kCB) {
#pragma unroll
        for (int c = 0; c < kCB; ++c) Dn[r][c] = row[c];
    }
    return ok;
}

// whole workgroup: write the factored block Dn / Dninv of block column kb0 (width nbb) to the factor
__device__ __forceinline__ void publish_block(double (*Dn)[kCB + 1], const double* Dninv, double* Lf, double* dinv, int ld, int kb0, int nbb)
{
    for (int idx = threadIdx.x; idx < kCB * kCB; idx += kPT) {
        const int r = idx % kCB, c = idx / kCB;
        if (r < nbb && c < nbb && r >= c) st_shared(&Lf[(unsigned)(kb0 + c) * (unsigned)ld + (unsigned)(kb0 + r)], Dn[r][c]);
    }
    if (threadIdx.x < nbb) st_shared(&dinv[kb0 + threadIdx.x], Dninv[threadIdx.x]);
}

// Panel solve x L^T = a against the factored 32 x 32 block of the column, one row per lane, x in registers.
// DT is the block as the tiles use it: DT[p][c] = L[c][p] for c > p (a straight copy of the factor's column p, which
// is stored contiguously), DT[p][p] = 1 / L[p][p]; columns beyond a short block are identity.  Column by column: once
// x[p] is final it is taken out of every later column -- the subtractions reach each x[c] in the order p = 0, 1, ...
// (a dot-product form's order, without its dependent chain); the block's columns come from LDS at uniform addresses
// (broadcast reads).
typedef __attribute__((address_space(3))) const double lds_cdouble;
template <bool DOUBLE_BUFFER, class Store>
__device__ __forceinline__ void trsm32(double (&x)[kCB], const double* DT_generic, Store store)
{
    // The block's LDS address goes through a register the compiler cannot see into: with a link-time constant base
    // it materialises every one of the 500 element addresses in its own SGPR (spilled through v_writelane) instead
    // of using the instruction's immediate offset.
    unsigned dt_off = (unsigned)(size_t)(lds_cdouble*)DT_generic;
    asm volatile("" : "+v"(dt_off));
    lds_cdouble* DT = (lds_cdouble*)(size_t)dt_off;
    double cur[kCB];
#pragma unroll
    for (int c = 0; c < kCB; ++c) cur[c] = DT[c];
#pragma unroll
    for (int p = 0; p < kCB; ++p) {
        // The next column of the block is requested as ONE batch (left alone, the scheduler sinks every read next to its
        // use: read, wait, two FMAs, read, wait ...).  DOUBLE_BUFFER: before this column's products, into a second set of
        // registers -- its LDS latency hides behind them (workgroup 0's look-ahead, which has the registers); otherwise
        // right behind the products into the registers of the column just applied (the tiles: a second buffer spilled).
        double nxt[kCB];
        if (DOUBLE_BUFFER) {
#pragma unroll
            for (int c = p + 1; c < kCB; ++c) nxt[c] = DT[(p + 1) * kCB + c];
            __builtin_amdgcn_sched_barrier(0);
        }
        x[p] = x[p] * cur[p];
#pragma unroll
        for (int c = p + 1; c < kCB; ++c) x[c] -= x[p] * cur[c];
        store(p, x[p]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = p + 1; c < kCB; ++c) cur[c] = DOUBLE_BUFFER ? nxt[c] : DT[(p + 1) * kCB + c];
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The same solve with LPR (2 or 4) adjacent lanes per row: lane q of a row's group holds the columns s * LPR + q in
// x[s].  Column p is scaled by its owner (lane p % LPR), reaches the row's other lanes through a quad-permute DPP move
// and is taken out of the later columns by whichever lane holds them -- per element the operations and their order are
// those of trsm32 (x[p] * d, then x[c] -= x[p] * DT[p][c] for p ascending), so the results are the same bits; per lane it
// is 1 / LPR of the products and of the LDS reads, on LPR times as many waves.  The next column of the block is
// always fetched ahead (the registers are there: 32 / LPR values per lane).
template <int LPR, int OWNER>
__device__ __forceinline__ double trsm_bcast_from(double v)
{
    static_assert(LPR == 2 || LPR == 4, "two or four lanes per row");
    constexpr int CTRL = LPR == 4 ? OWNER * 0x55 : (OWNER | (OWNER << 2) | ((2 + OWNER) << 4) | ((2 + OWNER) << 6));
    return dpp_mov<CTRL, 0xf, 0xf>(v, v);
}
template <int LPR, int P>
struct TrsmStep {
    static constexpr int NS = kCB / LPR, SP = P / LPR, QO = P % LPR;
    template <class DTp>
    __device__ __forceinline__ static void run(double (&x)[NS], double (&cur)[NS], double& dpp, DTp DT, int q)
    {
        // the next column's entries for this lane, and its diagonal entry, before this column's products
        double nxt[NS], dnx = 0.0;
        if (P + 1 < kCB) {
#pragma unroll
            for (int s = (P + 1) / LPR; s < NS; ++s) nxt[s] = DT[(P + 1) * kCB + s * LPR + q];
            dnx = DT[(P + 1) * kCB + (P + 1)];
        }
        __builtin_amdgcn_sched_barrier(0);
        const double t = x[SP] * dpp;                          // (the owner's is column P; the other lanes' products are not used)
        const double xp = trsm_bcast_from<LPR, QO>(t);
        // slot SP: the lanes behind the owner hold later columns, the owner takes the final value, the lanes before it
        // hold finished columns
        if (QO + 1 < LPR) {
            const double upd = x[SP] - xp * cur[SP];
            x[SP] = q > QO ? upd : (q == QO ? xp : x[SP]);
        } else {
            x[SP] = q == QO ? xp : x[SP];
        }
#pragma unroll
        for (int s = SP + 1; s < NS; ++s) x[s] -= xp * cur[s];
        __builtin_amdgcn_sched_barrier(0);
        if (P + 1 < kCB) {
#pragma unroll
            for (int s = (P + 1) / LPR; s < NS; ++s) cur[s] = nxt[s];
            dpp = dnx;
        }
        if constexpr (P + 1 < kCB) TrsmStep<LPR, P + 1>::run(x, cur, dpp, DT, q);
    }
};
template <int LPR>
__device__ __forceinline__ void trsm32_lanes(double (&x)[kCB / LPR], const double* DT_generic, int q)
{
    constexpr int NS = kCB / LPR;
    unsigned dt_off = (unsigned)(size_t)(lds_cdouble*)DT_generic;      // (opaque base: see trsm32)
    asm volatile("" : "+v"(dt_off));
    lds_cdouble* DT = (lds_cdouble*)(size_t)dt_off;
    double cur[NS], dpp = DT[0];
#pragma unroll
    for (int s = 0; s < NS; ++s) cur[s] = DT[s * LPR + q];
    TrsmStep<LPR, 0>::run(x, cur, dpp, DT, q);
}

// One 64 x 64 tile (bx >= by) of the trailing update of block column k0, by one 256-thread sub-group; the two
// __syncthreads() are workgroup-wide, so every sub-group of the workgroup calls this the same number of times
// (has = false: no tile this round).  DT: the factored diagonal block of the column as trsm32 reads it, in LDS.
// skip_next_diag: leave rows / columns k1 .. k1+32 (the next diagonal block, inside tile (0, 0)) untouched.
// after_loads(): called by every thread once the tile's loads (panel rows, old values) are on their way -- the helpers
// write the block of the column they fetched into DT there and synchronise, so that fetch and these loads share one
// trip to memory instead of taking two in a row.
template <class AfterLoads>
__device__ __forceinline__ void chol_tile(double* A, double* Lf, int n, int ld, int k0, int nb, bool has, int bx, int by,
                                          const double* DT, double* panel, bool skip_next_diag, unsigned long long* prof,
                                          AfterLoads after_loads)
{
    const int t = threadIdx.x & 255, wv = t >> 6, lane = t & 63;
    const int k1 = k0 + nb;
    const int i0 = k1 + bx * 64, j0 = k1 + by * 64;
    double (*Ai)[64 + 1] = reinterpret_cast<double (*)[64 + 1]>(panel);
    double (*Aj)[64 + 1] = reinterpret_cast<double (*)[64 + 1]>(panel + kLdsPanel);
    const unsigned long long ts0 = prof_now();
    // panel rows of the tile: 64 of Ai and 64 of Aj, two lanes per row on the sub-group's four waves
    constexpr int LPR = 2, NS = kCB / LPR;
    const int pr = t >> 1, q = t & 1, lrow = pr & 63;
    const bool first = pr < 64;
    const int prow = first ? i0 + lrow : j0 + lrow;
    const bool pvalid = has && (first ? prow <= n : prow < n);      // row n (rhs) only ever is a tile ROW
    double x[NS];
    // (address clamped instead of a predicated load: all loads in flight, not one round trip each)
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) {
        const int c = s2 * LPR + q;
        x[s2] = ld_shared(&A[(c < nb && pvalid) ? (unsigned)(k0 + c) * (unsigned)ld + (unsigned)prow : 0u]);      // (32-bit: a system holds < 2^32 doubles)
    }
    after_loads();
    if (has) {
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) x[s2] = ((s2 * LPR + q) < nb && pvalid) ? x[s2] : 0.0;
        trsm32_lanes<LPR>(x, DT, q);
        double (*P)[64 + 1] = first ? Ai : Aj;
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) P[s2 * LPR + q][lrow] = (s2 * LPR + q) < nb ? x[s2] : 0.0;
        // (one branch around all the stores of the factor: a conditional store per column splits the unrolled code)
        if (first && by == 0 && pvalid) {
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2)
                if (s2 * LPR + q < nb) st_shared(&Lf[(unsigned)(k0 + s2 * LPR + q) * (unsigned)ld + (unsigned)prow], x[s2]);
        }
    }
    __syncthreads();
    prof_add1(prof, kProfHelpSolve, ts0);
    const unsigned long long tu0 = prof_now();
    if (has && j0 < n) {
        // the tile's old values: all sixteen loads in flight behind the product below (sc1: straight from memory;
        // one load - subtract - store per element would be sixteen round trips)
        const TileOwn own{wv, lane};
        double old[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = i0 + own.i_of(e), j = j0 + own.j_of(e);
            const bool in = j < n && i <= n && i >= j;
            old[e] = ld_shared(&A[in ? (unsigned)j * (unsigned)ld + (unsigned)i : 0u]);         // (always a valid address: no branch per element)
        }
        double acc[16];
        tile_product(Ai, Aj, own, acc);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = i0 + own.i_of(e), j = j0 + own.j_of(e);
            if (j >= n) continue;
            // (the next diagonal block belongs to workgroup 0, which reads its old values while this tile runs)
            if (skip_next_diag && i < min(k1 + kCB, n)) continue;       // (row n is the right-hand side, never part of it)
            if (i <= n && i >= j) st_shared(&A[(unsigned)j * (unsigned)ld + (unsigned)i], old[e] - acc[e]);
        }
    }
    __syncthreads();
    prof_add1(prof, kProfHelpUpdate, tu0);
}

// Factor the (n+1) x n system A (row n = right-hand side) into Lf / dinv; all G workgroups call it together.
// Returns (on workgroup 0) 0 or 1 + the first block column with a non-positive pivot.
__device__ __noinline__ int pchol_factor(double* A, double* Lf, double* dinv, int n, GridBar& gb, double* lds, bool& alive)
{
    const int ld = n + 1, g = blockIdx.x, G = gb.G, tid = threadIdx.x, sg = tid >> 8;
    double* DT = lds + kLdsD;                                  // [32][32] factored block of the current column (trsm32 layout)
    // workgroup 0's private blocks: panel rows of the next diagonal block, the block itself, its reciprocal pivots
    double (*Lrow)[64 + 1] = reinterpret_cast<double (*)[64 + 1]>(lds + kLdsR);          // (a slot per lane of the wave: stores need no predicate)
    double (*Dn)[kCB + 1] = reinterpret_cast<double (*)[kCB + 1]>(lds + kLdsR + kLdsPanel);
    double* Dninv = lds + kLdsR + kLdsPanel + kCB * (kCB + 1);
    int info = 0;
    // the factored block Dn / Dninv -> DT (workgroup 0 keeps the block it has just factored: no trip through memory)
    auto dn_to_dt = [&](int nbb) {
        for (int idx = tid; idx < kCB * kCB; idx += kPT) {
            const int c = idx >> 5, r = idx & 31;              // DT[c][r] = L[r][c]
            DT[idx] = (r < nbb && c < nbb) ? (r > c ? Dn[r][c] : (r == c ? Dninv[c] : 0.0)) : (r == c ? 1.0 : 0.0);
        }
    };
    if (g == 0) {                                             // diagonal block 0 straight from the system
        const int nb0 = min(kCB, n);
        for (int idx = tid; idx < kCB * kCB; idx += kPT) {
            const int r = idx % kCB, c = idx / kCB;
            Dn[r][c] = (r < nb0 && c < nb0 && r >= c) ? ld_shared(&A[(unsigned)c * (unsigned)ld + (unsigned)r]) : (r == c ? 1.0 : 0.0);
        }
        __syncthreads();
        bool ok = true;
        if (tid < 64) ok = potrf32_wave(Dn, Dninv);
        if (tid == 0) lds[kLdsMisc] = ok ? 0.0 : 1.0;
        __syncthreads();
        if (lds[kLdsMisc] != 0.0) info = 1;
        publish_block(Dn, Dninv, Lf, dinv, ld, 0, nb0);
        dn_to_dt(nb0);
    }
    alive = grid_barrier(gb);
    for (int k0 = 0; k0 < n && alive; k0 += kCB) {
        const unsigned long long tw0 = prof_now();
        const int nb = min(kCB, n - k0), k1 = k0 + nb;
        constexpr int kDtPass = kCB * kCB / kPT;
        double dtv[kDtPass];
        if (g > 0) {                                          // helpers fetch the block workgroup 0 published: requested here,
#pragma unroll                                                 // written to DT behind the first tile's own loads
            for (int qd = 0; qd < kDtPass; ++qd) {
                const int idx = tid + qd * kPT, c = idx >> 5, r = idx & 31;
                const bool in = r < nb && c < nb;
                dtv[qd] = ld_shared(in ? (r == c ? &dinv[k0 + c] : &Lf[(unsigned)(k0 + c) * (unsigned)ld + (unsigned)(k0 + r)]) : &dinv[k0]);
            }
        }
        auto write_dt = [&]() {
            if (g > 0) {
#pragma unroll
                for (int qd = 0; qd < kDtPass; ++qd) {
                    const int idx = tid + qd * kPT, c = idx >> 5, r = idx & 31;
                    const bool in = r < nb && c < nb;
                    DT[idx] = in ? (r >= c ? dtv[qd] : 0.0) : (r == c ? 1.0 : 0.0);
                }
                __syncthreads();
            }
        };
        const bool tiles_here = G == 1 || g > 0;
        if (tiles_here) {
            const int nti = (n + 1 - k1 + 63) / 64, ntj = max((n - k1 + 63) / 64, 1);
            int total = 0;
            for (int by = 0; by < ntj; ++by) total += max(nti - by, 0);
            const int nslots = G == 1 ? kPSG : kPSG * (G - 1);
            const int slot = G == 1 ? sg : kPSG * (g - 1) + sg;
            for (int base = 0; base < total; base += nslots) {
                const int tt = base + slot;
                const bool has = tt < total;
                int by = 0, rem = tt;
                if (has) { while (rem >= nti - by) { rem -= nti - by; ++by; } }
                if (base == 0)
                    chol_tile(A, Lf, n, ld, k0, nb, has, by + rem, by, DT, lds + kLdsR + sg * 2 * kLdsPanel,
                              G > 1 && has && by == 0 && rem == 0, gb.prof, write_dt);
                else
                    chol_tile(A, Lf, n, ld, k0, nb, has, by + rem, by, DT, lds + kLdsR + sg * 2 * kLdsPanel,
                              G > 1 && has && by == 0 && rem == 0, gb.prof, [] {});
            }
        }
        if (g == 0 && k1 < n) {
            const int nb2 = min(kCB, n - k1);
            if (G == 1) {                                     // the tiles above have updated the block in A
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                for (int idx = tid; idx < kCB * kCB; idx += kPT) {
                    const int r = idx % kCB, c = idx / kCB;
                    Dn[r][c] = (r < nb2 && c < nb2 && r >= c) ? ld_shared(&A[(unsigned)(k1 + c) * (unsigned)ld + (unsigned)(k1 + r)]) : (r == c ? 1.0 : 0.0);
                }
            } else {
                // One step ahead of the helpers: rows k1 .. k1+nb2 of block column k0 against the block, then the
                // update of the next diagonal block with them -- the same operations, in the same order, as chol_tile
                // applies.  The old values of that block are requested together with the panel rows (one round trip).
                unsigned long long tl0 = prof_now();
                constexpr int kTri = kCB * (kCB + 1) / 2, kTriPass = (kTri + kPT - 1) / kPT;    // lower-triangle elements
                double oldv[kTriPass];
                int er[kTriPass], es[kTriPass];
#pragma unroll
                for (int q = 0; q < kTriPass; ++q) {
                    const int idx = tid + q * kPT;
                    int r = (int)((sqrtf(8.0f * idx + 1.0f) - 1.0f) * 0.5f);
                    if (r * (r + 1) / 2 > idx) --r;
                    if ((r + 1) * (r + 2) / 2 <= idx) ++r;
                    er[q] = r; es[q] = idx - r * (r + 1) / 2;
                    const bool in = idx < kTri && r < nb2;                                      // (sc <= r < nb2)
                    oldv[q] = ld_shared(&A[in ? (unsigned)(k1 + es[q]) * (unsigned)ld + (unsigned)(k1 + r) : 0u]);
                }
                if (tid < 128) {
                    constexpr int LPR = 4, NS = kCB / LPR;
                    const int lrow = tid >> 2, q = tid & 3, prow = k1 + lrow;
                    const bool pvalid = lrow < nb2;
                    double x[NS];
#pragma unroll
                    for (int s2 = 0; s2 < NS; ++s2) {
                        const int c = s2 * LPR + q;
                        x[s2] = ld_shared(&A[(c < nb && pvalid) ? (unsigned)(k0 + c) * (unsigned)ld + (unsigned)prow : 0u]);
                    }
#pragma unroll
                    for (int s2 = 0; s2 < NS; ++s2) x[s2] = ((s2 * LPR + q) < nb && pvalid) ? x[s2] : 0.0;
                    prof_add(gb.prof, kProfLookLoad, tl0); tl0 = prof_now();
                    trsm32_lanes<LPR>(x, DT, q);
#pragma unroll
                    for (int s2 = 0; s2 < NS; ++s2) Lrow[s2 * LPR + q][lrow] = (s2 * LPR + q) < nb ? x[s2] : 0.0;
                }
                // identity padding / zeros above the diagonal, then the lower triangle on top
                for (int idx = tid; idx < kCB * kCB; idx += kPT) Dn[idx >> 5][idx & 31] = (idx >> 5) == (idx & 31) ? 1.0 : 0.0;
                __syncthreads();
                prof_add(gb.prof, kProfLookSolve, tl0); tl0 = prof_now();
#pragma unroll
                for (int q = 0; q < kTriPass; ++q) {
                    const int idx = tid + q * kPT, r = er[q], sc = es[q];
                    if (idx < kTri && r < nb2) {
                        double acc = 0.0;
#pragma unroll 8
                        for (int p = 0; p < kCB; ++p) acc = fma(Lrow[p][r], Lrow[p][sc], acc);
                        Dn[r][sc] = oldv[q] - acc;
                    }
                }
                prof_add(gb.prof, kProfLookFill, tl0);
            }
            __syncthreads();
            unsigned long long tp0 = prof_now();
            bool ok = true;
            if (tid < 64) ok = potrf32_wave(Dn, Dninv);
            if (tid == 0) lds[kLdsMisc] = ok ? 0.0 : 1.0;
            __syncthreads();
            prof_add(gb.prof, kProfLookPotrf, tp0); tp0 = prof_now();
            if (lds[kLdsMisc] != 0.0 && info == 0) info = k1 + 1;
            publish_block(Dn, Dninv, Lf, dinv, ld, k1, nb2);
            dn_to_dt(nb2);                                    // (the tiles / the panel solve of this step are done with DT)
            prof_add(gb.prof, kProfLookPub, tp0);
        }
        prof_add(gb.prof, kProfFactorWork, tw0);
        const unsigned long long tb0 = prof_now();
        alive = grid_barrier(gb);
        prof_add(gb.prof, kProfFactorWait, tb0);
        prof_add1(gb.prof, kProfHelpWait, tb0);
        if (gb.prof && threadIdx.x == 0 && blockIdx.x == 0) gb.prof[kProfSteps] += 1;
    }
    return info;
}

// L^T x = y (y = row n of the factor), workgroup 0 only: dense_chol.hpp::chol_backsolve's sums in its order, but the
// factor's entries of block column kb-1 (4 columns per wave, the first 768 rows below the block; its diagonal block;
// its right-hand side) are requested BEFORE the triangle of block column kb is solved -- they do not depend on x --
// so a step costs its arithmetic, not a round trip to memory per phase.  x also lives in LDS while it fits.
__device__ __noinline__ void pchol_backsolve(const double* Lf, int n, double* x, double* lds, unsigned long long* prof = nullptr)
{
    constexpr int NW = kPT / 64, CPW = (kCB + NW - 1) / NW, MAXM = 12, DPT = (kCB * kCB + kPT - 1) / kPT;
    double (*D)[kCB + 1] = reinterpret_cast<double (*)[kCB + 1]>(lds + kLdsD);
    double* t = lds + kLdsDinv;
    double* xs = lds + kLdsR;
    const bool x_in_lds = n <= kPSG * 2 * kLdsPanel;
    const int ld = n + 1;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l = lane & 31;
    const int nblk = (n + kCB - 1) / kCB;
    double pre[CPW][MAXM], dpre[DPT], ypre = 0.0;
    auto prefetch = [&](int kb) {
        const int k0 = kb * kCB, nb = min(kCB, n - k0), k1 = k0 + nb;
        const double* colp[CPW];
#pragma unroll
        for (int q = 0; q < CPW; ++q) {
            const int c = min(wave + q * NW, nb - 1);         // (columns beyond a short block: a valid one, result unused)
            colp[q] = Lf + (size_t)(k0 + c) * ld + k1 + lane;
        }
#pragma unroll
        for (int m = 0; m < MAXM; ++m) {
            if (k1 + 64 * m < n) {                            // (wave-uniform: slices of 64 rows that exist)
                const bool in = k1 + lane + 64 * m < n;
#pragma unroll
                for (int q = 0; q < CPW; ++q) pre[q][m] = ld_shared(in ? colp[q] + 64 * m : Lf);
            }
        }
#pragma unroll
        for (int q = 0; q < DPT; ++q) {
            const int idx = tid + q * kPT, r = idx % kCB, c = idx / kCB;
            dpre[q] = ld_shared(&Lf[(idx < kCB * kCB && r < nb && c < nb && r >= c) ? (unsigned)(k0 + c) * (unsigned)ld + (unsigned)(k0 + r) : 0u]);
        }
        ypre = ld_shared(&Lf[l < nb ? (unsigned)(k0 + l) * (unsigned)ld + (unsigned)n : 0u]);
    };
    prefetch(nblk - 1);
    for (int kb = nblk - 1; kb >= 0; --kb) {
        const int k0 = kb * kCB, nb = min(kCB, n - k0), k1 = k0 + nb;
        unsigned long long tq = prof_now();
        // x of the rows below the block: one batch of LDS reads (the same rows for every column of the wave), then the
        // products -- read next to its use, each of them is a round trip of its own
        double xr[MAXM];
#pragma unroll
        for (int m = 0; m < MAXM; ++m) {
            const int r = k1 + lane + 64 * m;
            xr[m] = x_in_lds ? xs[r < n ? r : 0] : gptr(x)[r < n ? r : 0];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < CPW; ++q) {
            const int c = wave + q * NW;
            if (c < nb) {                                     // (wave-uniform)
                double acc = 0.0;
#pragma unroll
                for (int m = 0; m < MAXM; ++m) {
                    const int r = k1 + lane + 64 * m;
                    if (r < n) acc += pre[q][m] * xr[m];
                }
                for (int r = k1 + lane + 64 * MAXM; r < n; r += 64) acc += ld_shared(&Lf[(size_t)(k0 + c) * ld + r]) * gptr(x)[r];
                acc = wave_sum(acc);
                t[c] = acc;                                   // (uniform: every lane stores it)
            }
        }
#pragma unroll
        for (int q = 0; q < DPT; ++q) {
            const int idx = tid + q * kPT, r = idx % kCB, c = idx / kCB;
            if (idx < kCB * kCB) D[r][c] = (r < nb && c < nb && r >= c) ? dpre[q] : (r == c ? 1.0 : 0.0);
        }
        const double ycur = ypre;
        prof_add(prof, kProfBsDots, tq); tq = prof_now();
        if (kb > 0) prefetch(kb - 1);
        prof_add(prof, kProfBsPrefetch, tq); tq = prof_now();
        __syncthreads();
        prof_add(prof, kProfBsSync, tq); tq = prof_now();
        if (wave == 0) {
            double v = l < nb ? ycur - t[l] : 0.0;
            const double dinv = 1.0 / D[l][l];
            double col[kCB];
#pragma unroll
            for (int r = 0; r < kCB; ++r) col[r] = D[r][l];
#pragma unroll
            for (int r = kCB - 1; r >= 0; --r) {
                const double xr = read_lane(v, r) * read_lane(dinv, r);
                v = l == r ? xr : (l < r ? fma(-col[r], xr, v) : v);
            }
            if (lane < nb) {
                gptr(x)[k0 + lane] = v;
                if (x_in_lds) xs[k0 + lane] = v;
            }
        }
        __syncthreads();
        prof_add(prof, kProfBsTri, tq);
    }
}

// The factorisation + back substitution alone (ipc_debug_dense_solve: parity of the persistent orchestration with
// dense_chol.hpp::chol_solve_device on arbitrary systems)
__global__ __launch_bounds__(kPT, 1) void pchol_test_kernel(double* A, double* Lf, double* dinv, int n, double* x, PersistCtl* ctl, int* info)
{
    extern __shared__ double lds[];
    GridBar gb{&ctl->bar, 0u, (int)gridDim.x, &ctl->error, nullptr};
    bool alive = true;
    const int r = pchol_factor(A, Lf, dinv, n, gb, lds, alive);
    if (blockIdx.x == 0) {
        pchol_backsolve(Lf, n, x, lds);
        if (threadIdx.x == 0) *info = alive ? r : -1;
    }
}

// ---- pose-type bindings -------------------------------------------------------------------------------------
struct PersistSe2 {
    using Dev = ClusterDev;
    static constexpr int kD = 3, kNPS = 9, kNND = 3, kSC1 = 1, kSC2 = 2;
    static constexpr int kEdgeDoubles = 43, kLoopDoubles = 27;          // per ld / per loop, + (ld + nl) of chi_edges
    __device__ static void load_initial(const Dev& D, const double* src, int V, int i)
    {
        const size_t s = (size_t)D.lo + i;
        auto g = gptr(src);
        gptr(D.X.x)[i] = g[s]; gptr(D.X.y)[i] = g[(size_t)V + s]; gptr(D.X.th)[i] = g[2 * (size_t)V + s];
        gptr(D.X.c)[i] = g[3 * (size_t)V + s]; gptr(D.X.s)[i] = g[4 * (size_t)V + s];
    }
    __device__ static void eval(const Dev& D, bool trial, int i, double (&v)[1])
    {
        if (trial) gk_eval_at(D, D.Xn, D.en, D.len, i, v); else gk_eval_at(D, D.X, D.e, D.le, i, v);
    }
    __device__ static void chi_edges(const Dev& D, int i) { gk_chi_edges_at(D, i); }
    __device__ static void force(const Dev& D, int i) { gk_force_at(D, i); }
    __device__ static void b(const Dev& D, int i, double (&v)[1]) { gk_b_at(D, i, v); }
    __device__ static void bHb_psi(const Dev& D, int i, double (&v)[1]) { gk_bHb_psi_at(D, i, v); }
    __device__ static void assemble(const Dev& D, int l1, int l2) { gk_assemble_at(D, l1, l2); }
    template <class Put, class PutRhs>
    __device__ static void assemble_core(const Dev& D, int l1, int l2, Put put, PutRhs put_rhs) { gk_assemble_core(D, l1, l2, put, put_rhs); }
    // the banded kernel's unit of assembly work: the whole 3 x 3 block
    static constexpr int kAsmRows = 1;
    template <class Put, class PutRhs>
    __device__ static void assemble_row(const Dev& D, int l1, int l2, int, Put put, PutRhs put_rhs) { gk_assemble_core(D, l1, l2, put, put_rhs); }
    __device__ static void nu(const Dev& D, int l) { gk_nu_at(D, l); }
    __device__ static void events(const Dev& D, int j) { gk_events_at(D, j); }
    __device__ static void rho(const Dev& D, int i) { gk_rho_at(D, i); }
    __device__ static void term(const Dev& D, int i) { gk_term_at(D, i); }
    __device__ static void h(const Dev& D, int i, double (&v)[2]) { gk_h_at(D, i, v); }
    __device__ static void blend(const Dev& D, double alpha, int i, double (&v)[2]) { gk_blend_at(D, alpha, i, v); }
    __device__ static void update(const Dev& D, double p, double q, int i, double (&v)[1]) { gk_update_at(D, p, q, i, v); }
    static void exchange(Dev& D)                 // the view after a commit: committed <-> trial buffers
    {
        PoseArr tx = D.X; D.X = D.Xn; D.Xn = tx;
        double* t = D.e; D.e = D.en; D.en = t;
        t = D.le; D.le = D.len; D.len = t;
    }
    static void carve(Dev& D, double* edge, double* loop, int ld, int nl)
    {
        double* p = edge;
        auto take = [&](size_t n) { double* q = p; p += n; return q; };
        D.X = PoseArr{take(ld), take(ld), take(ld), take(ld), take(ld)};
        D.Xn = PoseArr{take(ld), take(ld), take(ld), take(ld), take(ld)};
        D.e = take(3 * (size_t)ld); D.en = take(3 * (size_t)ld); D.g = take(3 * (size_t)ld); D.m = take(3 * (size_t)ld);
        D.b = take(3 * (size_t)ld); D.h = take(3 * (size_t)ld); D.ps = take(9 * (size_t)ld); D.nd = take(3 * (size_t)ld);
        D.sc = take(3 * (size_t)ld);
        D.chi_edges = take((size_t)ld + nl);
        double* q = loop;
        auto takel = [&](size_t n) { double* r = q; q += n; return r; };
        D.le = takel(3 * (size_t)nl); D.len = takel(3 * (size_t)nl); D.lg = takel(3 * (size_t)nl); D.lm = takel(3 * (size_t)nl);
        D.gam = takel(9 * (size_t)nl); D.nu = takel(3 * (size_t)nl); D.rhs = takel(3 * (size_t)nl);
    }
};

struct PersistSe3 {
    using Dev = ClusterDev3;
    static constexpr int kD = 6, kNPS = 27, kNND = 6, kSC1 = 3, kSC2 = 3;
    static constexpr int kEdgeDoubles = 99, kLoopDoubles = 72;
    __device__ static void load_initial(const Dev& D, const double* src, int src_ld, int i)
    {
#pragma unroll
        for (int k = 0; k < 12; ++k) D.X[(size_t)k * D.ld + i] = src[(size_t)k * src_ld + D.lo + i];
    }
    __device__ static void eval(const Dev& D, bool trial, int i, double (&v)[1])
    {
        if (trial) gk3_eval_at(D, D.Xn, D.en, D.len, i, v); else gk3_eval_at(D, D.X, D.e, D.le, i, v);
    }
    __device__ static void chi_edges(const Dev& D, int i) { gk3_chi_edges_at(D, i); }
    __device__ static void force(const Dev& D, int i) { gk3_force_at(D, i); }
    __device__ static void b(const Dev& D, int i, double (&v)[1]) { gk3_b_at(D, i, v); }
    __device__ static void bHb_psi(const Dev& D, int i, double (&v)[1]) { gk3_bHb_psi_at(D, i, v); }
    __device__ static void assemble(const Dev& D, int l1, int l2) { gk3_assemble_at(D, l1, l2); }
    template <class Put, class PutRhs>
    __device__ static void assemble_core(const Dev& D, int l1, int l2, Put put, PutRhs put_rhs) { gk3_assemble_core(D, l1, l2, put, put_rhs); }
    // the banded kernel's unit of assembly work: one row of a 6 x 6 block
    static constexpr int kAsmRows = 6;
    template <class Put, class PutRhs>
    __device__ static void assemble_row(const Dev& D, int l1, int l2, int r, Put put, PutRhs put_rhs) { gk3_assemble_row(D, l1, l2, r, put, put_rhs); }
    __device__ static void nu(const Dev& D, int l) { gk3_nu_at(D, l); }
    __device__ static void events(const Dev& D, int j) { gk3_events_at(D, j); }
    __device__ static void rho(const Dev& D, int i) { gk3_rho_at(D, i); }
    __device__ static void term(const Dev& D, int i) { gk3_term_at(D, i); }
    __device__ static void h(const Dev& D, int i, double (&v)[2]) { gk3_h_at(D, i, v); }
    __device__ static void blend(const Dev& D, double alpha, int i, double (&v)[2]) { gk3_blend_at(D, alpha, i, v); }
    __device__ static void update(const Dev& D, double p, double q, int i, double (&v)[1]) { gk3_update_at(D, p, q, i, v); }
    static void exchange(Dev& D)                 // the view after a commit: committed <-> trial buffers
    {
        double* t = D.X; D.X = D.Xn; D.Xn = t;
        t = D.e; D.e = D.en; D.en = t;
        t = D.le; D.le = D.len; D.len = t;
    }
    static void carve(Dev& D, double* edge, double* loop, int ld, int nl)
    {
        double* p = edge;
        auto take = [&](size_t n) { double* q = p; p += n; return q; };
        D.X = take(12 * (size_t)ld); D.Xn = take(12 * (size_t)ld);
        D.e = take(6 * (size_t)ld); D.en = take(6 * (size_t)ld); D.g = take(6 * (size_t)ld); D.m = take(6 * (size_t)ld);
        D.b = take(6 * (size_t)ld); D.h = take(6 * (size_t)ld); D.ps = take(27 * (size_t)ld);
        D.nd = take(6 * (size_t)ld); D.sc = take(6 * (size_t)ld);
        D.chi_edges = take((size_t)ld + nl);
        double* q = loop;
        auto takel = [&](size_t n) { double* r = q; q += n; return r; };
        D.le = takel(6 * (size_t)nl); D.len = takel(6 * (size_t)nl); D.lg = takel(6 * (size_t)nl); D.lm = takel(6 * (size_t)nl);
        D.gam = takel(36 * (size_t)nl); D.nu = takel(6 * (size_t)nl); D.rhs = takel(6 * (size_t)nl);
    }
};

// ---- the kernel ---------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(kPT, 1) void cluster_persist_kernel(typename T::Dev D0, typename T::Dev D1, PersistArgs P)
{
    // D0: the problem as carved by the host; D1: the same with the committed / trial buffers exchanged (a commit of
    // the dog-leg is a switch between the two views).  Both are read where they are, in the kernel-argument segment:
    // taking the address of the by-value arguments makes the compiler copy them to scratch, and every D->member of every
    // phase body then starts with a scratch load (one more dependent memory round trip per phase; 608 of them in the
    // kernel).  Through the segment pointer they are scalar loads of the constant address space.
    using Dev = typename T::Dev;
    constexpr size_t kView1 = (sizeof(Dev) + alignof(Dev) - 1) / alignof(Dev) * alignof(Dev);     // offset of D1 behind D0
    const __attribute__((address_space(4))) char* kargs = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    int vsel = 0;                                             // which view is the committed one (uniform)
    // a phase's copy of the view: scalar loads of the members the phase uses, nothing of it lives across phases
    auto view = [&](int sel) {
        Dev d;
        __builtin_memcpy(&d, kargs + (__builtin_amdgcn_readfirstlane(sel) ? kView1 : 0), sizeof(Dev));
        return d;
    };
    (void)D0; (void)D1;
    extern __shared__ double lds[];
    const int tid = threadIdx.x, g = blockIdx.x, G = gridDim.x;
    GridBar gb{&P.ctl->bar, 0u, G, &P.ctl->error, P.prof};
    const int L = D0.L, nl = D0.nl, ld = D0.ld;
    const int n = T::kD * nl;
    double* A = D0.S;
    double* Lf = D0.S + (size_t)(n + 1) * n;
    bool alive = true;

    auto assemble_share = [&](const typename T::Dev& Dv) {
        const long total = (long)nl * nl;
        for (long q = (long)g * kPT + tid; q < total; q += (long)G * kPT) {
            const int l1 = (int)(q / nl), l2 = (int)(q - (long)l1 * nl);
            if (l2 <= l1) T::assemble(Dv, l1, l2);
        }
    };

    if (g != 0) {                                             // ---- helpers ----
        for (;;) {
            if (!grid_barrier(gb)) return;                                                   // B1: a command is posted
            if (__hip_atomic_load(&P.ctl->cmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1) return;
            if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                  // the leader's ps / Gamma / loop errors
            __syncthreads();
            { const Dev Dv = view(__hip_atomic_load(&P.ctl->le_sel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); assemble_share(Dv); }
            if (!grid_barrier(gb)) return;                                                   // B2: the system is assembled
            pchol_factor(A, Lf, P.dinv, n, gb, lds, alive);
            if (!alive) return;
        }
    }

    // ---- leader ----
    const unsigned long long tk0 = prof_now();
    const int nidx = L + nl + 1, nblk = (nidx + 255) / 256;
    int n_commit = 0;
    { const Dev Dv = view(0); lead_for(L + 1, [&](int i) { T::load_initial(Dv, P.src, P.src_ld, i); }); }

    auto evaluate = [&](bool trial) {
        double tot[1];
        const Dev Dv = view(vsel);
        lead_reduce<1>(nblk, lds, tot, [&](int i, double (&v)[1]) { T::eval(Dv, trial, i, v); });
        return tot[0];
    };
    // H h_gn = b through the capacitance system; returns the solver's info word
    auto linearize = [&](double& bb, double& bHb, double& hh, double& bh) {
        unsigned long long t0 = prof_now();
        { const Dev Dv = view(vsel); lead_for(nidx, [&](int i) { T::force(Dv, i); }); }
        { const Dev Dv = view(vsel); double tot[1]; lead_reduce<1>(nblk, lds, tot, [&](int i, double (&v)[1]) { T::b(Dv, i, v); }); bb = tot[0]; }
        { const Dev Dv = view(vsel); double tot[1]; lead_reduce<1>(nblk, lds, tot, [&](int i, double (&v)[1]) { T::bHb_psi(Dv, i, v); }); bHb = tot[0]; }
        { const Dev Dv = view(vsel); lead_scan(Dv.ps, T::kNPS, L, ld, lds); }
        prof_add(P.prof, kProfPre, t0); t0 = prof_now();
        // hand the assembly's inputs to the helpers, factor together
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (G > 1 && tid == 0) {
            __hip_atomic_store(&P.ctl->le_sel, n_commit & 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&P.ctl->cmd, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        alive = grid_barrier(gb);                                                            // B1
        prof_add(P.prof, kProfHandoff, t0); t0 = prof_now();
        { const Dev Dv = view(vsel); assemble_share(Dv); }
        if (alive) alive = grid_barrier(gb);                                                 // B2
        prof_add(P.prof, kProfAssemble, t0); t0 = prof_now();
        int info = 0;
        if (alive) info = pchol_factor(A, Lf, P.dinv, n, gb, lds, alive);
        prof_add(P.prof, kProfFactor, t0); t0 = prof_now();
        { const Dev Dv = view(vsel); pchol_backsolve(Lf, n, Dv.rhs, lds, P.prof); }
        prof_add(P.prof, kProfBacksolve, t0); t0 = prof_now();
        { const Dev Dv = view(vsel); lead_for(nl, [&](int l) { T::nu(Dv, l); }); }
        { const Dev Dv = view(vsel); lead_for(L + 2, [&](int j) { T::events(Dv, j); }); }
        { const Dev Dv = view(vsel); lead_scan(Dv.nd, T::kNND, L, ld, lds); }
        { const Dev Dv = view(vsel); lead_for(nidx, [&](int i) { T::rho(Dv, i); }); }
        { const Dev Dv = view(vsel); lead_scan(Dv.sc, T::kSC1, L, ld, lds); }
        { const Dev Dv = view(vsel); lead_for(nidx, [&](int i) { T::term(Dv, i); }); }
        { const Dev Dv = view(vsel); lead_scan(Dv.sc + (size_t)T::kSC1 * ld, T::kSC2, L, ld, lds); }
        { const Dev Dv = view(vsel); double tot[2]; lead_reduce<2>(nblk, lds, tot, [&](int i, double (&v)[2]) { T::h(Dv, i, v); }); hh = tot[0]; bh = tot[1]; }
        prof_add(P.prof, kProfPost, t0);
        if (P.prof && tid == 0) P.prof[kProfIterations] += 1;
        return info;
    };

    // g2o OptimizationAlgorithmDogleg::solve / SparseOptimizer::optimize, as cluster_dogleg() runs it from the host.
    // The scalar control arithmetic is kept unfused: cluster_dogleg() computes it on the CPU (x86-64, no FMA).
    PersistOut o{};
    bool aborted = false;
    double currentChi = evaluate(false);
    o.chi2_initial = currentChi;
    {
#pragma clang fp contract(off)
    double delta = 1e4;
    const int maxTrials = 100;
    const int n_edges = L + nl;
    bool lastGN = false;
    for (int it = 0; it < P.iterations && alive; ++it) {
        // a speculative solve whose starting state has been overtaken (the host committed an earlier candidate) stops here
        if (P.abort_word && __hip_atomic_load(P.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= P.launch_id) { aborted = true; break; }
        double bb, bHb, hh, bh;
        const int info = linearize(bb, bHb, hh, bh);
        if (!alive) break;
        if (info != 0) { o.flags |= 2; o.iterations = it + 1; break; }
        const double hHh = bh;
        const double alpha = bb / bHb, hsdNorm = sqrt(alpha * alpha * bb), hgnNorm = sqrt(hh);
        if (lastGN && hgnNorm < delta && fabs(bh) * n_edges < P.term_eps * currentChi) {
            o.iterations = it + 1; o.tries += maxTrials; o.flags |= 1;
            break;
        }
        const double deltaAtEntry = delta;
        bool goodStep = false;
        int numTries = 0;
        const unsigned long long tt0 = prof_now();
        do {
            ++numTries;
            int stepType;
            double beta = 0.0, sdScale = 0.0;
            if (hgnNorm < delta) stepType = 0;
            else if (hsdNorm > delta) { stepType = 1; sdScale = delta / hsdNorm; }
            else {
                stepType = 2;
                double tot[2];
                { const Dev Dv = view(vsel); lead_reduce<2>(nblk, lds, tot, [&](int i, double (&v)[2]) { T::blend(Dv, alpha, i, v); }); }
                const double c = tot[0], bma = tot[1];
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
            double changed[1];
            { const Dev Dv = view(vsel); lead_reduce<1>(nblk, lds, changed, [&](int i, double (&v)[1]) { T::update(Dv, pcoef, qcoef, i, v); }); }
            const bool anyChanged = changed[0] != 0.0;
            const double newChi = evaluate(true);
            ++o.evals;
            const double nonLinearGain = currentChi - newChi;
            if (fabs(linearGain) < 1e-12) linearGain = 1e-12;
            const double rho = nonLinearGain / linearGain;
            if (rho > 0) {
                goodStep = true;
                currentChi = newChi;
                ++n_commit;
                vsel = n_commit & 1;
            }
            if (rho > 0.75) delta = fmax(delta, 3 * hdlNorm);
            else if (rho < 0.25) delta *= 0.5;
            if (!goodStep) {
                if (rho != rho) {
                    numTries = maxTrials;
                } else if (stepType == 0) {
                    while (numTries < maxTrials && hgnNorm < delta) { ++numTries; delta *= 0.5; }
                } else if (stepType == 1 && !anyChanged) {
                    numTries = maxTrials;
                }
            }
        } while (!goodStep && numTries < maxTrials);
        prof_add(P.prof, kProfTrial, tt0);
        lastGN = goodStep && numTries == 1 && hgnNorm < deltaAtEntry;
        o.iterations = it + 1;
        o.tries += numTries;
        if (numTries == maxTrials || !goodStep) { o.flags |= 1; break; }
    }
    }
    // release the helpers
    if (G > 1) {
        if (tid == 0) __hip_atomic_store(&P.ctl->cmd, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        grid_barrier(gb);
    }
    // per-edge chi2 of the committed state and their maximum (over the edges that have a number; NaN only if none of them is positive)
    { const Dev Dv = view(vsel); lead_for(nidx, [&](int i) { T::chi_edges(Dv, i); }); }
    {
        double mx = 0.0;
        bool nan = false;
        const Dev Dv = view(vsel);
        for (int i = tid; i < L + nl; i += kPT) {
            const double c = gptr(Dv.chi_edges)[i];
            if (c != c) nan = true; else mx = fmax(mx, c);
        }
        mx = wave_max(mx);
        const bool wnan = __any(nan);
        double* red = lds + kLdsRed;
        if ((tid & 63) == 0) { red[tid >> 6] = mx; red[16 + (tid >> 6)] = wnan ? 1.0 : 0.0; }
        __syncthreads();
        if (tid == 0) {
            double m = 0.0, f = 0.0;
            for (int w = 0; w < kPT / 64; ++w) { m = fmax(m, red[w]); f += red[16 + w]; }
            o.max_chi2 = (f != 0.0 && !(m > 0.0)) ? __builtin_nan("") : m;
            o.chi2_total = currentChi;
            o.x_sel = n_commit & 1;
            o.error = !alive ? 1 : (aborted ? 2 : 0);
            o.device_ticks = (int)(prof_now() - tk0);
            *P.out = o;
            prof_add(P.prof, kProfTotal, tk0);
        }
    }
}

}  // namespace ipc
#include "cluster_band.hpp"     // the same solve for clusters of thousands of loops (banded capacitance system, every workgroup in every phase)
namespace ipc {

// ---- host side ----------------------------------------------------------------------------------------------
// One solver instance = one set of workspaces + one pinned result record; launch() enqueues a solve on a stream and
// returns, wait() blocks for its record.  Several instances on different streams run concurrently (speculative
// candidate window of the incremental mode).
// The environment knobs of a solver, read ONCE per engine (ipc_create) and handed to every instance that engine makes:
// the pipeline's slot solvers are created at the first check, long after the caller's environment may have changed
// (round 6: until then an engine created under IPC_BAND_MIN_N=0 ran its one-at-a-time checks through the band kernel
// and its pipeline through the dense one).
struct PersistKnobs {
    int band_min_n = 1024, band_split_min = 8, band_team_wgs = 0, band_team_reject = 0, fault_every = 0;
    static PersistKnobs from_env()
    {
        PersistKnobs k;
        if (const char* e = getenv("IPC_BAND_MIN_N")) { if (*e) k.band_min_n = atoi(e); }
        if (const char* e = getenv("IPC_BAND_SPLIT")) { if (*e) k.band_split_min = std::max(0, atoi(e)); }
        if (const char* e = getenv("IPC_BAND_TEAM")) { if (*e) k.band_team_wgs = std::max(0, atoi(e)); }
        if (const char* e = getenv("IPC_BAND_TEAM_REJECT")) { if (*e) k.band_team_reject = std::max(0, atoi(e)); }
        if (const char* e = getenv("IPC_PERSIST_FAULT_EVERY")) { if (*e) k.fault_every = std::max(0, atoi(e)); }
        return k;
    }
};
template <class T>
class PersistSolver {
public:
    using Dev = typename T::Dev;
    double term_eps = 0.0;
    unsigned long long* d_prof = nullptr;       // optional device buffer [kProfN] the leader accumulates its phase clocks into
    const int* d_abort_word = nullptr;          // optional host-mapped word (device address); the solve stops once it is >= launch_id
    int launch_id = 0;
    int max_helpers = 39;                       // workgroups besides the leader (kLdsTotal = 141 824 bytes of LDS each: one per CU)
    // clusters of at least this many capacitance unknowns whose loops form a band go to cluster_band_kernel
    // (cluster_band.hpp); IPC_BAND_MIN_N, negative = never.  Below it (C1's 759, C2's 480 unknowns): the dense kernel, bit for bit as in rounds 3-4.
    int band_split_min = 8;                     // IPC_BAND_SPLIT: split the factorisation of a band of at least this many half-widths of loops; 0: never
    int band_team_reject = 0;                   // IPC_BAND_TEAM_REJECT (experiments; 0: off): workgroups per team of a banded solve the caller expects to reject
                                                // (`economy`).  Measured with the per-XCD budget of spec_pump: C4 prefix 23.1 s without, 24.8 / 27.8 / 29.5 s
                                                // with 5 / 6 / 7 -- a reject on fewer workgroups holds its slot longer, and the run is bound by the accept chain
    bool economy = false;                       // set per launch by the pipeline (scheduling only: results do not depend on the workgroup count)
    int band_team_wgs = 0;                      // IPC_BAND_TEAM (experiments): workgroups per team of a split factorisation; 0: the rule in launch()
    int band_min_n = 1024;                      // (2 048 in the first round-5 runs: C4's first 700 candidates 5.6 s -> 3.9 s; at 1 000 unknowns the dense
                                                // trailing update is already several rounds of tiles per block column, the band's is one)

    explicit PersistSolver(const PersistKnobs& k = PersistKnobs::from_env())
    {
        band_min_n = k.band_min_n; band_split_min = k.band_split_min; band_team_wgs = k.band_team_wgs;
        band_team_reject = k.band_team_reject; fault_every_ = k.fault_every;
    }
    ~PersistSolver() { release(); }
    bool last_was_band() const { return last_band_; }
    const BandLayout& last_band_layout() const { return band_; }

    hipError_t launch(hipStream_t st, const double* chain, int estride, const double* cand, int cstride, const double* src,
                      int src_ld, int lo, int hi, const std::vector<int>& members, const int* from, const int* to,
                      int iterations)
    {
        const int L = hi - lo, nl = (int)members.size(), n = T::kD * nl;
        // the band structure of the cluster's loops, if it is large enough to look for one
        BandPlan plan;
        if (band_min_n >= 0 && n >= band_min_n) {
            std::vector<int> a(nl), b(nl);
            for (int l = 0; l < nl; ++l) { a[l] = std::min(from[members[l]], to[members[l]]); b[l] = std::max(from[members[l]], to[members[l]]); }
            plan = band_plan(T::kD, a, b, band_min_n);
        }
        last_band_ = plan.use;
        std::vector<int> reordered;
        if (plan.use) {
            reordered.resize(nl);
            for (int q = 0; q < nl; ++q) reordered[q] = members[plan.order[q]];
            band_.nb = T::kD * plan.nlb; band_.m = T::kD * (nl - plan.nlb) + 1; band_.W = T::kD * (plan.bwb + 1);
            band_.ldb = band_.W + band_.m; band_.n = n;
        }
        const std::vector<int>& mem = plan.use ? reordered : members;
        // Split factorisation (cluster_band.hpp, BandArgs::split_s): worth it from ~8 band widths of loops on
        int split_s = -1;
        if (plan.use && band_split_min > 0 && plan.nlb >= band_split_min * (plan.bwb + 1)) {   // (NOT a function of the helper count: results must not depend on it)
            split_s = (plan.nlb - (plan.bwb + 1)) / 2;
            const int m = band_.m, W = band_.W;
            band_.nb = T::kD * (split_s + plan.bwb + 1); band_.n = band_.nb + m - 1;
            band2_.nb = T::kD * (plan.nlb - split_s); band2_.m = m; band2_.W = W; band2_.ldb = W + m; band2_.n = band2_.nb + m - 1;
        }
        last_split_ = split_s >= 0;
        const size_t band_doubles = split_s >= 0 ? 2 * (band_.doubles() + band2_.doubles()) : 2 * band_.doubles();
        // The factorisation and the back substitution index one system with 32-bit products (j * ld + i, BandLayout::at32):
        // a system of 2^31 doubles and more (16 GB; ~11 000 SE3 loops in ONE non-banded cluster) would wrap silently.
        {
            const size_t one = plan.use ? std::max(band_.doubles(), split_s >= 0 ? band2_.doubles() : (size_t)0) : ((size_t)n + 1) * n;
            if (one >= ((size_t)1 << 31)) return hipErrorInvalidValue;
        }
        IPC_CL_CHK(ensure(L, nl, plan.use ? band_doubles : 2 * ((size_t)n + 1) * n));
        const int ld = L + 2;
        Dev& D = dev_;
        D.chain = chain; D.estride = estride; D.lo = lo; D.L = L; D.nl = nl; D.ld = ld;
        D.cand = cand; D.cstride = cstride;
        T::carve(D, d_edge_, d_loop_, ld, nl);
        D.S = d_S_; D.ldS = n + 1;
        D.partial = nullptr; D.scal = nullptr;
        tab_.build(lo, hi, mem, from, to);
        // (the staging buffer of this launch must outlive its copy: launches are enqueued without waiting for the
        // previous one of this instance -- speculative solves get aborted and replaced -- so the buffers rotate)
        tab_slot_ = (tab_slot_ + 1) % kTabSlots;
        if (tab_used_[tab_slot_]) IPC_CL_CHK(hipEventSynchronize(ev_tab_[tab_slot_]));
        std::memcpy(h_tab_[tab_slot_], tab_.host.data(), sizeof(int) * tab_.size());
        IPC_CL_CHK(hipMemcpyAsync(d_int_, h_tab_[tab_slot_], sizeof(int) * tab_.size(), hipMemcpyHostToDevice, st));
        IPC_CL_CHK(hipEventRecord(ev_tab_[tab_slot_], st));
        tab_used_[tab_slot_] = true;
        D.lfrom = tab_.lfrom(d_int_); D.lto = tab_.lto(d_int_); D.lcand = tab_.lcand(d_int_);
        D.adj_ptr = tab_.adj_ptr(d_int_); D.adj_item = tab_.adj_item(d_int_);
        D.ev_ptr = tab_.ev_ptr(d_int_); D.ev_item = tab_.ev_item(d_int_);
        IPC_CL_CHK(hipMemsetAsync(d_ctl_, 0, sizeof(PersistCtl), st));
        // workgroups: the leader + one helper per four 64 x 64 tiles of the first trailing update
        const int nt = (n + 63) / 64;
        const int tiles0 = n > 64 ? nt * (nt + 1) / 2 : 0;
        int G = 1 + std::min(max_helpers, (tiles0 + kPSG - 1) / kPSG);
        if (tiles0 == 0) G = 1;
        PersistArgs P{src, src_ld, iterations, term_eps, d_ctl_, d_out_, d_dinv_, d_prof, d_abort_word, launch_id};
        // (once per process and kernel, whichever thread comes first: engines of ipc_run_sharded live on worker threads)
        static std::once_flag attr_once;
        static hipError_t attr_rc = hipSuccess;
        static int resident_limit = 1 << 30;
        std::call_once(attr_once, [] {
            attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&cluster_persist_kernel<T>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * kLdsTotal));
            // every workgroup of a launch must be resident (grid barriers): never more than the device can hold at once
            int per_cu = 0, dev = 0;
            hipDeviceProp_t prop;
            if (attr_rc == hipSuccess && hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&cluster_persist_kernel<T>), kPT,
                                                             sizeof(double) * kLdsTotal) == hipSuccess && per_cu > 0)
                resident_limit = per_cu * prop.multiProcessorCount;
        });
        IPC_CL_CHK(attr_rc);
        Dev D1 = D;
        T::exchange(D1);
        if (plan.use) {
            static std::once_flag band_once;
            static hipError_t band_rc = hipSuccess;
            std::call_once(band_once, [] {
                band_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&cluster_band_kernel<T>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * kLdsTotal));
            });
            IPC_CL_CHK(band_rc);
            // workgroups: one per two tiles of a block column's trailing update ((W + m) / 64 tile rows), and enough of
            // them that a chain phase is a handful of poses per thread
            const int R = band_.W + band_.m, nti = (R + 63) / 64, tiles = nti * (nti + 1) / 2;
            // (more workgroups than the tiles need make every barrier slower: C4's first 1 500 candidates 17.5 s with 9, 21.6 s
            // with 20, 33.0 s with 40 workgroups per solve -- same digest)
            const int want_tiles = (tiles + kPSG - 1) / kPSG, want_chain = std::min(24, (L + nl) / (4 * kPT));
            const int want = std::max(want_tiles, want_chain);
            G = 1 + std::min(max_helpers, want);
            G = std::max(1, std::min(G, resident_limit));
            if (economy && band_team_reject > 0) G = std::min(G, 2 * band_team_reject);
            if (split_s >= 0) {                               // two teams of equal size
                // (a team = the workgroups its tiles need; the chain phases run on both teams together.  Prefixes of C4 / C5 with 3 / 5 / 7 / 9 / 12
                // workgroups per team: 18.5 / 13.9 / 15.4 / 13.3 / 14.1 s and 17.1 / 14.2 / 16.6 / 18.6 / 19.3 s, same digests -- C5 keeps 16 solves in
                // flight and is bound by the CUs they hold, its expected rejects run on the reject helper limit: tools/band_team_sweep.sh)
                const int team_rule = std::max(1 + want_tiles, (2 + want_chain) / 2);
                int Gt = std::max(1, std::min(band_team_wgs > 0 ? band_team_wgs : team_rule, std::min((max_helpers + 1) / 2, resident_limit / 2)));
                if (economy && band_team_reject > 0) Gt = std::min(Gt, band_team_reject);   // (two workgroups at least: one per team)
                G = 2 * Gt;
            }
            static const bool band_debug = getenv("IPC_BAND_DEBUG") != nullptr;
            if (band_debug)
                fprintf(stderr, "[band] L %d loops %d (band %d, half-width %d blocks) n %d W %d m %d workgroups %d split at loop %d\n", L, nl, plan.nlb,
                        plan.bwb, n, band_.W, band_.m, G, split_s);
            double* A2 = d_S_ + 2 * band_.doubles();
            BandArgs Q{band_, d_S_, d_S_ + band_.doubles(), d_dinv_, d_gpart_, d_gscan_, plan.nlb, plan.bwb, d_abort_seen_, reinterpret_cast<const double*>(d_abort_seen_ + 8),
                       split_s, band2_, A2, A2 + band2_.doubles(), d_dinv_ + band_.n + kCB, d_ctl_->team_bar};
            hipLaunchKernelGGL(cluster_band_kernel<T>, dim3(G), dim3(kPT), sizeof(double) * kLdsTotal, st, D, D1, P, Q);
        } else {
            G = std::max(1, std::min(G, resident_limit));
            hipLaunchKernelGGL(cluster_persist_kernel<T>, dim3(G), dim3(kPT), sizeof(double) * kLdsTotal, st, D, D1, P);
        }
        IPC_CL_CHK(hipGetLastError());
        IPC_CL_CHK(hipMemcpyAsync(h_out_, d_out_, sizeof(PersistOut), hipMemcpyDeviceToHost, st));
        st_ = st;
        last_G_ = G;
        return hipSuccess;
    }

    hipError_t wait(ClusterOut& out)
    {
        IPC_CL_CHK(hipStreamSynchronize(st_));
        return fetch(out);
    }
    // the result record of a solve known to have ended (an event behind launch() has completed)
    hipError_t fetch(ClusterOut& out)
    {
        out = ClusterOut{};
        out.max_chi2 = h_out_->max_chi2; out.chi2_total = h_out_->chi2_total; out.chi2_initial = h_out_->chi2_initial;
        out.iterations = h_out_->iterations; out.tries = h_out_->tries; out.flags = h_out_->flags; out.evals = h_out_->evals;
        x_sel_ = h_out_->x_sel;
        device_us_ = 0.01 * h_out_->device_ticks;
        aborted_ = h_out_->error == 2;
        timed_out_ = h_out_->error == 1;
        // IPC_PERSIST_FAULT_EVERY=k (tests): every k-th solve of this instance that was not aborted is reported LOST, as if a
        // grid barrier had given up -- what a foreign tenant on the GPU does to a launch whose workgroups must all be resident
        if (fault_every_ > 0 && !aborted_ && ++fault_count_ % fault_every_ == 0) timed_out_ = true;
        return hipSuccess;
    }
    bool aborted() const { return aborted_; }
    double device_us() const { return device_us_; }           // of the last fetched solve (IPC_SPEC_STATS)
    // a grid barrier gave up (some workgroup of the launch never became resident beside foreign work on the GPU): the
    // result is void and the caller redoes the solve with the host-driven kernels, which need no co-residency
    bool timed_out() const { return timed_out_; }

    // workspaces for chains up to L poses and nl loops, up front
    // nl: loops the per-loop arrays and tables hold (cheap: ~100 doubles per loop); nl_dense: loops the DENSE capacitance
    // system + factor are reserved for (2 (d nl)^2 doubles; larger clusters are banded and grow the buffer as they come)
    hipError_t reserve(int L, int nl, int nl_dense = -1)
    {
        if (nl_dense < 0) nl_dense = nl;
        return ensure(L, nl, 2 * ((size_t)T::kD * nl_dense + 1) * ((size_t)T::kD * nl_dense));
    }
    const Dev& dev() const { return dev_; }
    bool result_in_second() const { return x_sel_ != 0; }
    // problems the leader's LDS staging cannot hold go to the host-driven solver
    static bool fits(int L, int nl) { return (L + nl + 1 + 255) / 256 <= kLeadMaxBlk; }
    int ld() const { return dev_.ld; }
    int workgroups() const { return last_G_; }

private:
    Dev dev_{};
    hipStream_t st_ = nullptr;
    int capL_ = 0, capNl_ = 0, x_sel_ = 0, last_G_ = 1;
    size_t capS_ = 0;                           // doubles of d_S_ (system + factor: dense 2 (n + 1) n, banded 2 n (W + m))
    bool aborted_ = false, timed_out_ = false, last_band_ = false, last_split_ = false;
    int fault_every_ = 0; long fault_count_ = 0;
    BandLayout band_{}, band2_{};
    double device_us_ = 0.0;
    double *d_edge_ = nullptr, *d_loop_ = nullptr, *d_S_ = nullptr, *d_dinv_ = nullptr;
    double *d_gpart_ = nullptr, *d_gscan_ = nullptr;       // reduction partials / scan run totals of the band kernel
    int* d_abort_seen_ = nullptr;
    int* d_int_ = nullptr;
    static constexpr int kTabSlots = 8;
    int* h_tab_[kTabSlots] = {};
    hipEvent_t ev_tab_[kTabSlots] = {};
    bool tab_used_[kTabSlots] = {};
    int tab_slot_ = 0;
    PersistCtl* d_ctl_ = nullptr;
    PersistOut *d_out_ = nullptr, *h_out_ = nullptr;
    LoopTables tab_;

    void release()
    {
        hipFree(d_edge_); hipFree(d_loop_); hipFree(d_S_); hipFree(d_dinv_); hipFree(d_int_); hipFree(d_ctl_); hipFree(d_out_);
        hipFree(d_gpart_); hipFree(d_gscan_); hipFree(d_abort_seen_);
        d_gpart_ = d_gscan_ = nullptr; d_abort_seen_ = nullptr; capS_ = 0;
        if (st_) hipStreamSynchronize(st_);
        if (h_out_) hipHostFree(h_out_);
        for (int k = 0; k < kTabSlots; ++k) {
            if (h_tab_[k]) hipHostFree(h_tab_[k]);
            if (ev_tab_[k]) hipEventDestroy(ev_tab_[k]);
            h_tab_[k] = nullptr; ev_tab_[k] = nullptr; tab_used_[k] = false;
        }
        d_edge_ = d_loop_ = d_S_ = d_dinv_ = nullptr; d_int_ = nullptr; d_ctl_ = nullptr; d_out_ = h_out_ = nullptr;
        capL_ = capNl_ = 0;
    }
    hipError_t ensure(int L, int nl, size_t sdoubles)
    {
        if (!h_out_) {
            IPC_CL_CHK(hipHostMalloc(&h_out_, sizeof(PersistOut)));
            IPC_CL_CHK(hipMalloc(&d_out_, sizeof(PersistOut)));
            IPC_CL_CHK(hipMalloc(&d_ctl_, sizeof(PersistCtl)));
            IPC_CL_CHK(hipMalloc(&d_abort_seen_, sizeof(int) * 16));       // [0]: abort word as workgroup 0 saw it; [8..9]: a double 0.0
            IPC_CL_CHK(hipMemset(d_abort_seen_, 0, sizeof(int) * 16));
            IPC_CL_CHK(hipStreamSynchronize(nullptr));             // (NULL-stream memsets are not ordered against the non-blocking streams the solves run on)
            for (int k = 0; k < kTabSlots; ++k) IPC_CL_CHK(hipEventCreateWithFlags(&ev_tab_[k], hipEventDisableTiming));
        }
        if (L > capL_ || nl > capNl_) {
            // (hipFree waits for the whole device, i.e. for every other solve in flight: grow in big steps -- the clusters
            // of a run grow by one loop per accept)
            const int nL = L > capL_ ? std::max(L, capL_ + capL_ / 2) : capL_;
            const int nN = nl > capNl_ ? std::max(std::max(nl, 48), capNl_ + capNl_ / 2) : capNl_;
            if (st_) IPC_CL_CHK(hipStreamSynchronize(st_));        // (a launch in flight still uses the old workspaces)
            hipFree(d_edge_); hipFree(d_loop_); hipFree(d_dinv_); hipFree(d_int_); hipFree(d_gpart_); hipFree(d_gscan_);
            for (int k = 0; k < kTabSlots; ++k) { if (h_tab_[k]) hipHostFree(h_tab_[k]); h_tab_[k] = nullptr; tab_used_[k] = false; }
            d_edge_ = d_loop_ = d_dinv_ = d_gpart_ = d_gscan_ = nullptr; d_int_ = nullptr;
            capL_ = capNl_ = 0;
            const size_t ld = (size_t)nL + 2, n = (size_t)T::kD * nN;
            IPC_CL_CHK(hipMalloc(&d_edge_, sizeof(double) * (T::kEdgeDoubles * ld + ld + nN)));
            IPC_CL_CHK(hipMalloc(&d_loop_, sizeof(double) * (T::kLoopDoubles * (size_t)nN + 8)));
            IPC_CL_CHK(hipMalloc(&d_dinv_, sizeof(double) * (2 * n + 4 * kCB)));      // (a split factorisation keeps two systems' pivots)
            IPC_CL_CHK(hipMalloc(&d_int_, sizeof(int) * LoopTables::capacity(nL, nN)));
            IPC_CL_CHK(hipMalloc(&d_gpart_, sizeof(double) * 2 * 8 * ((ld + nN + 255) / 256 + 2)));
            IPC_CL_CHK(hipMalloc(&d_gscan_, sizeof(double) * 27 * ((ld + 1023) / 1024 + 2)));
            for (int k = 0; k < kTabSlots; ++k) IPC_CL_CHK(hipHostMalloc(&h_tab_[k], sizeof(int) * LoopTables::capacity(nL, nN)));
            capL_ = nL; capNl_ = nN;
        }
        if (sdoubles > capS_) {
            const size_t want = std::max(sdoubles + sdoubles / 4, 2 * capS_);      // (hipFree waits for the whole device: few, large steps)
            if (st_) IPC_CL_CHK(hipStreamSynchronize(st_));
            hipFree(d_S_); d_S_ = nullptr; capS_ = 0;
            IPC_CL_CHK(hipMalloc(&d_S_, sizeof(double) * want));
            capS_ = want;
        }
        return hipSuccess;
    }
};

}  // namespace ipc
#include "cluster_literal_band.hpp"   // Levenberg retry of the literal normal equations in the banded layout
