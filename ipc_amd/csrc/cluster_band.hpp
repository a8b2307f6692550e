// Cluster solve for LARGE clusters (thousands of accepted loops, chains of tens of thousands of poses): the faithful
// incremental mode on BASELINE configs[3] / [4] (IPC::agreementCheck, reference src/consensus.cpp:43-75, whatever
// computeIndependentSubgraph :124-171 grows the cluster to; the reference solves it with g2o's variable-block solver +
// Eigen sparse LLT, src/utils.cpp:104-105).  Included by cluster_persist.hpp (it builds on that file's primitives:
// grid barrier, potrf32_wave, trsm32_lanes, tile_product, the pose-type bindings PersistSe2 / PersistSe3).
//
// Same mathematics as the persistent kernel (capacitance system of the cluster's loops, cluster_se2.hpp), two changes:
//
//  * the capacitance matrix is BANDED with a dense border.  S_ll' = Gamma_l (P[min hi] - P[max lo]) Gamma_l'^T is zero
//    unless the vertex ranges of loops l and l' overlap; with the loops ordered by their first vertex the overlapping
//    partners of a loop are a contiguous run behind it, so S is block-banded with half-bandwidth = the longest such run
//    (sphere2500: loops of span 50 -> 49 blocks; C5: spans <= 200 at 0.1 loops per pose -> ~40 blocks).  The few loops
//    that span far more than the others (the candidate itself when it is an outlier, falsely accepted outliers) would
//    widen the band for everyone: they go LAST, as dense rows -- an arrowhead.  Storage per column j (BandLayout): the W
//    band rows j .. j+W-1, then the m dense rows (the wide loops' unknowns and the right-hand side, which rides along as
//    the last row exactly as in dense_chol.hpp).  The factorisation is the blocked right-looking Cholesky of
//    cluster_persist.hpp::pchol_factor on that layout: the trailing update of a block column touches the (W + m)^2 / 2
//    entries inside the profile instead of (n - k)^2 / 2 -- n * W^2 operations and n * (W + m) doubles instead of n^3 / 3
//    and n^2 (C4: 14 400 unknowns, W = 300; C5: 30 000 unknowns, W ~ 250);
//
//  * EVERY workgroup of the launch runs every chain phase (poses / loops interleaved over the workgroups in runs of 512)
//    and the dog-leg control flow redundantly, from scalars that are reduced in a fixed order that does not depend on the
//    number of workgroups: per run of 256 indices the four wave totals go to memory, every workgroup adds all of them in
//    run order.  A leader looping over 50 000 poses alone (cluster_persist.hpp) would spend milliseconds per phase.
//    Phases are separated by a grid barrier with an agent-scope release / acquire pair (the phase bodies use plain
//    loads and stores; per-XCD L2s are not coherent with each other) -- about 5 us each, ~25 per iteration, against
//    a factorisation of milliseconds at these sizes.  The results do not depend on the number of workgroups.
//
// Clusters below IPC_BAND_MIN_N unknowns (default 1 024) keep the dense persistent kernel, bit for bit as before.
#pragma once

namespace ipc {

// ---- layout -----------------------------------------------------------------------------------------------------
// (struct BandLayout: cluster_common.hpp -- the literal normal equations of the Levenberg retry use the same layout)
// the rows below a block column that ends in front of column k1 and can hold a non-zero of it: band rows k1 .. and every
// dense row from max(k1, nb) on, numbered 0 .. R-1 ("virtual rows", the last one is the right-hand side)
struct BandRows {
    int k1, nbr, dstart, R;
    __device__ __forceinline__ BandRows(const BandLayout& B, int k1_)
    {
        k1 = k1_;
        nbr = k1 < B.nb ? min(B.nb - k1, B.W - 1) : 0;
        dstart = max(k1, B.nb);
        R = nbr + (B.n + 1 - dstart);
    }
    __device__ __forceinline__ int row(int v) const { return v < nbr ? k1 + v : dstart + (v - nbr); }
};

struct BandArgs {
    BandLayout B;
    double* A;               // the system, B.doubles()
    double* Lf;              // the factor, same layout
    double* dinv;            // [n] reciprocal pivots
    double* gpart;           // [2][runs of 256][4 waves][2] reduction partials (two buffers, see band_reduce)
    double* gscan;           // [27][runs of 1024] run totals of the prefix sums
    int nlb;                 // band loops 0 .. nlb-1 (sorted by first vertex), wide loops nlb .. nl-1
    int bwb;                 // block half-bandwidth of the band loops
    int* abort_seen;         // device word: workgroup 0 publishes what it read from the host's abort word
    const double* zero;      // a word that holds 0.0 (entries outside the profile are read from it)
    // Split ("twisted") factorisation: the band loops are cut into T = [0, s), M = [s, s + bwb + 1) and the rest; system 1
    // (B / A / Lf / dinv above) holds T and M in their order + the wide loops, system 2 the rest and M in REVERSE order + the
    // wide loops.  Both eliminate their own loops at the same time (two teams of workgroups), the Schur complement system 2
    // leaves on M and the wide loops is added to system 1's, and system 1 finishes.  split_s < 0: one system, no split.
    int split_s;
    BandLayout B2;
    double *A2, *Lf2, *dinv2;
    unsigned* team_bar;      // [2] arrival counters of the two teams
};

// ---- grid-wide phase helpers --------------------------------------------------------------------------------------
// Every workgroup arrives once.  fence: the phase in front of the barrier wrote data with plain stores that other
// workgroups read behind it (release before the arrival, acquire after the last one has arrived; one lane each).
__device__ __forceinline__ bool band_barrier(GridBar& gb, bool fence)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (gb.G == 1) return true;
    gb.target += (unsigned)gb.G;
    if (threadIdx.x == 0) {
        if (fence) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the compiler may drop the wait behind buffer_wbl2: restated where it cannot)
        }
        __hip_atomic_fetch_add(gb.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(gb.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gb.target) {
            if (spins < 48) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(32);
            if (++spins > kSpinLimit) { __hip_atomic_store(gb.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            // (a workgroup that gave up has left the launch: nobody will complete this barrier)
            if ((spins & 255) == 0 && __hip_atomic_load(gb.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
        }
        if (fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return __hip_atomic_load(gb.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
}

// indices 0 .. n-1 in runs of kPT, run r on workgroup r % G (the same mapping as band_reduce's runs of 256)
template <class F>
__device__ __forceinline__ void band_for(int n, int G, F f)
{
    if ((int)blockIdx.x >= G) return;                          // (G: the workgroups that share the chain phases)
    for (int i = blockIdx.x * kPT + threadIdx.x; i < n; i += G * kPT) f(i);
}

// Sum of K per-index values over the indices 0 .. nblk*256-1; tot is the same bit pattern on every thread of every
// workgroup and does not depend on G: wave totals of each run of 256 (DPP scan), ((w0 + w1) + w2) + w3 per run, the runs
// in ascending order.  Ends the phase: contains its grid barrier (fenced: the phase's other outputs are published too).
// gpart0 holds TWO buffers of 8 doubles per run: consecutive reductions alternate between them (parity), because a fast
// workgroup starts writing the partials of the next reduction while a slow one still reads those of this one -- the
// barrier of the reduction in between is what separates two uses of the same buffer.
template <int K, class F>
__device__ __forceinline__ bool band_reduce(int nblk, double* lds, double* gpart0, unsigned& parity, GridBar& gb, int Gc, double (&tot)[K], F f)
{
    static_assert(K <= 2, "gpart holds two values per wave");
    double* gpart = gpart0 + (size_t)(parity & 1u) * 8 * (size_t)nblk;
    ++parity;
    const int tid = threadIdx.x, q = tid >> 8, t = tid & 255, G = Gc;
    double* rtot = lds + kLdsRed;                              // [nblk][K] run totals
    for (int vb0 = (int)blockIdx.x < G ? (int)blockIdx.x * kPSG : nblk; vb0 < nblk; vb0 += G * kPSG) {
        const int vb = vb0 + q;
        double v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = 0.0;
        if (vb < nblk) f(vb * 256 + t, v);
        double ws[K];
#pragma unroll
        for (int k = 0; k < K; ++k) ws[k] = wave_sum(v[k]);
        if (vb < nblk && (t & 63) == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) st_shared(&gpart[((size_t)vb * 4 + (t >> 6)) * 2 + k], ws[k]);
        }
    }
    const bool alive = band_barrier(gb, true);
    for (int r = tid; r < nblk; r += kPT) {
        double w[4][K];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int k = 0; k < K; ++k) w[a][k] = ld_shared(&gpart[((size_t)r * 4 + a) * 2 + k]);
#pragma unroll
        for (int k = 0; k < K; ++k) rtot[r * K + k] = ((w[0][k] + w[1][k]) + w[2][k]) + w[3][k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double acc = 0.0;
        for (int r = 0; r < nblk; ++r) acc += rtot[r * K + k];
        tot[k] = acc;
    }
    __syncthreads();
    return alive;
}

// In-place inclusive prefix sums over indices 1 .. L of K arrays (row length ld).  Runs of 1024 indices, run r on
// workgroup r % G: inside a run as cluster_persist.hpp::lead_scan_k (wave scans, wave totals added in wave order); the
// run totals go to memory, and after a barrier every run adds the sum of the totals in front of it (in run order).
// Independent of G.  Ends with a fenced barrier.
template <int K>
__device__ __forceinline__ bool band_scan_k(double* arr, int L, int ld, double* lds, double* gscan, GridBar& gb, int Gc)
{
    constexpr int NP = (1024 + kPT - 1) / kPT;
    double* wsum = lds + kLdsWsum;                             // [K][32]
    double* carry = lds + kLdsMisc + 8;                        // [K] (K <= 27; kLdsMisc + 64 is reserved)
    const int tid = threadIdx.x, G = Gc;
    const int nruns = (L + 1023) / 1024;
    for (int r = (int)blockIdx.x < G ? (int)blockIdx.x : nruns; r < nruns; r += G) {
        const int base = 1 + 1024 * r;
        double v[NP][K];
#pragma unroll
        for (int hp = 0; hp < NP; ++hp) {
            const int off = hp * kPT + tid, i = base + off;
            const bool in = off < 1024 && i <= L;
            const int ic = in ? i : 0;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const double a = gptr(arr)[(size_t)k * ld + ic];
                v[hp][k] = wave_inclusive_scan(in ? a : 0.0);
                wsum[k * 32 + (off >> 6)] = read_lane(v[hp][k], 63);
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double tt = 0.0;
            for (int w = 0; w < 16; ++w) tt += wsum[k * 32 + w];
            if (tid == k) st_shared(&gscan[(size_t)k * nruns + r], tt);
#pragma unroll
            for (int hp = 0; hp < NP; ++hp) {
                const int vw = (hp * kPT + tid) >> 6;
                double off = 0.0;
                for (int w = 0; w < vw && w < 16; ++w) off += wsum[k * 32 + w];
                v[hp][k] += off;
            }
        }
#pragma unroll
        for (int hp = 0; hp < NP; ++hp) {
            const int off = hp * kPT + tid, i = base + off;
            if (off < 1024 && i <= L) {
#pragma unroll
                for (int k = 0; k < K; ++k) gptr(arr)[(size_t)k * ld + i] = v[hp][k];
            }
        }
        __syncthreads();
    }
    if (nruns > 1) {
        if (!band_barrier(gb, false)) return false;            // (only the run totals cross workgroups here: sc1 stores / loads)
        for (int r = (int)blockIdx.x < G ? (int)blockIdx.x : nruns; r < nruns; r += G) {
            if (r == 0) continue;
            if (tid < K) {
                double c = 0.0;
                for (int rr = 0; rr < r; ++rr) c += ld_shared(&gscan[(size_t)tid * nruns + rr]);
                carry[tid] = c;
            }
            __syncthreads();
            const int base = 1 + 1024 * r;
#pragma unroll
            for (int hp = 0; hp < NP; ++hp) {
                const int off = hp * kPT + tid, i = base + off;
                if (off < 1024 && i <= L) {
#pragma unroll
                    for (int k = 0; k < K; ++k) gptr(arr)[(size_t)k * ld + i] += carry[k];
                }
            }
            __syncthreads();
        }
    }
    return band_barrier(gb, true);
}
__device__ __forceinline__ bool band_scan(double* arr, int K, int L, int ld, double* lds, double* gscan, GridBar& gb, int Gc)
{
    bool alive = true;
    while (K >= 9 && alive) { alive = band_scan_k<9>(arr, L, ld, lds, gscan, gb, Gc); arr += 9 * (size_t)ld; K -= 9; }
    while (K >= 3 && alive) { alive = band_scan_k<3>(arr, L, ld, lds, gscan, gb, Gc); arr += 3 * (size_t)ld; K -= 3; }
    while (K >= 1 && alive) { alive = band_scan_k<1>(arr, L, ld, lds, gscan, gb, Gc); arr += (size_t)ld; K -= 1; }
    return alive;
}

// ---- blocked Cholesky on the banded layout (cluster_persist.hpp::chol_tile / pchol_factor, rows by BandRows) ----------
template <class AfterLoads>
__device__ __forceinline__ void bchol_tile(double* A, double* Lf, const BandLayout& B, const BandRows& TR, int k0, int nbk, bool has,
                                           int bx, int by, const double* DT, double* panel, int skip_until, unsigned long long* prof,
                                           AfterLoads after_loads)
{
    const unsigned long long ts0 = prof_now();
    const int t = threadIdx.x & 255, wv = t >> 6, lane = t & 63;
    const int k1 = k0 + nbk;
    const int i0v = bx * 64, j0v = by * 64;
    double (*Ai)[64 + 1] = reinterpret_cast<double (*)[64 + 1]>(panel);
    double (*Aj)[64 + 1] = reinterpret_cast<double (*)[64 + 1]>(panel + kLdsPanel);
    constexpr int LPR = 2, NS = kCB / LPR;
    const int pr = t >> 1, q = t & 1, lrow = pr & 63;
    const bool first = pr < 64;
    const int pv = (first ? i0v : j0v) + lrow;
    const bool pvalid = has && (first ? pv < TR.R : pv < TR.R - 1);     // the right-hand side (last row) only ever is a tile ROW
    const int prow = TR.row(pvalid ? pv : 0);
    double x[NS];
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) {
        const int c = s2 * LPR + q;
        const bool ok = c < nbk && pvalid && B.in(prow, k0 + c);
        x[s2] = ld_shared(&A[ok ? B.at32(prow, k0 + c) : 0u]);
    }
    // the tile's old values do not depend on the panel solve: requested with the panel rows (one trip to memory for
    // both; they come from memory -- sc1 -- and a trip costs 2 - 4 us here, more than the solve and the product together)
    const TileOwn own{wv, lane};
    double old[16];
    unsigned inmask = 0;
    const bool upd = has && j0v < TR.R - 1;
    auto elem = [&](int e, unsigned& adr) -> bool {           // element e of this thread: inside the profile, and where
        const int iv = i0v + own.i_of(e), jv = j0v + own.j_of(e);
        const bool inr = upd && jv < TR.R - 1 && iv < TR.R && iv >= jv;
        const int i = TR.row(inr ? iv : 0), j = TR.row(inr ? jv : 0);
        bool in = inr && B.in(i, j);
        // (the next diagonal block belongs to workgroup 0, which reads its old values while this tile runs)
        if (i < skip_until) in = false;                       // (skip_until: one past the next diagonal block, 0 = none)
        adr = in ? B.at32(i, j) : 0u;
        return in;
    };
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        unsigned adr;
        const bool in = elem(e, adr);
        inmask |= (in ? 1u : 0u) << e;
        old[e] = ld_shared(&A[adr]);
    }
    after_loads();
    prof_add1(prof, kProfTileBlock, ts0);
    unsigned long long tq = prof_now();
    if (has) {
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
            const int c = s2 * LPR + q;
            x[s2] = (c < nbk && pvalid && B.in(prow, k0 + c)) ? x[s2] : 0.0;
        }
        asm volatile("" :: "v"(x[0]), "v"(x[NS - 1]));
        prof_add1(prof, kProfTileSelect, tq); tq = prof_now();
        trsm32_lanes<LPR>(x, DT, q);
        prof_add1(prof, kProfTileTrsm, tq); tq = prof_now();
        double (*P)[64 + 1] = first ? Ai : Aj;
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) P[s2 * LPR + q][lrow] = (s2 * LPR + q) < nbk ? x[s2] : 0.0;
        if (first && by == 0 && pvalid) {
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) {
                const int c = s2 * LPR + q;
                if (c < nbk && B.in(prow, k0 + c)) st_shared(&Lf[B.at32(prow, k0 + c)], x[s2]);
            }
        }
    }
    __syncthreads();
    prof_add1(prof, kProfTileStore, tq);
    prof_add1(prof, kProfHelpSolve, ts0);
    const unsigned long long tu0 = prof_now();
    if (upd) {
        double acc[16];
        tile_product(Ai, Aj, own, acc);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            unsigned adr;
            elem(e, adr);
            if (inmask & (1u << e)) st_shared(&A[adr], old[e] - acc[e]);
        }
    }
    __syncthreads();
    prof_add1(prof, kProfHelpUpdate, tu0);
}

// Factor the banded system A (B.n unknowns, right-hand side = last dense row) into Lf / dinv; all G workgroups call it
// together.  Workgroup 0 runs one block column ahead (next diagonal block), the others apply the tiles; G == 1: all of
// it on the one workgroup.  Returns 0 or 1 + the first block column with a non-positive pivot.
// Columns c_begin .. c_end-1 only (c_end < B.n: a PARTIAL factorisation -- the Schur complement of the eliminated columns
// stays in A, in place; c_begin > 0: continues one).  g / gb: this workgroup's index in, and the barrier of, the TEAM of
// workgroups that work on this system (the split factorisation runs two teams on two systems side by side).
__device__ __noinline__ int bband_factor(double* A, double* Lf, double* dinv, const BandLayout B, GridBar& gb, int g, double* lds, bool& alive,
                                         int c_begin, int c_end)
{
    const int G = gb.G, tid = threadIdx.x, sg = tid >> 8;
    double* DT = lds + kLdsD;
    double (*Lrow)[64 + 1] = reinterpret_cast<double (*)[64 + 1]>(lds + kLdsR);
    double (*Dn)[kCB + 1] = reinterpret_cast<double (*)[kCB + 1]>(lds + kLdsR + kLdsPanel);
    double* Dninv = lds + kLdsR + kLdsPanel + kCB * (kCB + 1);
    int info = 0;
    auto dn_to_dt = [&](int nbb) {
        for (int idx = tid; idx < kCB * kCB; idx += kPT) {
            const int c = idx >> 5, r = idx & 31;
            DT[idx] = (r < nbb && c < nbb) ? (r > c ? Dn[r][c] : (r == c ? Dninv[c] : 0.0)) : (r == c ? 1.0 : 0.0);
        }
    };
    auto publish = [&](int kb0, int nbb) {
        for (int idx = tid; idx < kCB * kCB; idx += kPT) {
            const int r = idx % kCB, c = idx / kCB;
            if (r < nbb && c < nbb && r >= c) st_shared(&Lf[B.at32(kb0 + r, kb0 + c)], Dn[r][c]);
        }
        if (tid < nbb) st_shared(&dinv[kb0 + tid], Dninv[tid]);
    };
    if (g == 0) {
        const int nb0 = min(kCB, c_end - c_begin);
        for (int idx = tid; idx < kCB * kCB; idx += kPT) {
            const int r = idx % kCB, c = idx / kCB;
            Dn[r][c] = (r < nb0 && c < nb0 && r >= c) ? ld_shared(&A[B.at32(c_begin + r, c_begin + c)]) : (r == c ? 1.0 : 0.0);
        }
        __syncthreads();
        bool ok = true;
        if (tid < 64) ok = potrf32_wave(Dn, Dninv);
        if (tid == 0) lds[kLdsMisc] = ok ? 0.0 : 1.0;
        __syncthreads();
        if (lds[kLdsMisc] != 0.0) info = c_begin + 1;
        publish(c_begin, nb0);
        dn_to_dt(nb0);
    }
    alive = band_barrier(gb, false);
    for (int k0 = c_begin; k0 < c_end && alive; k0 += kCB) {
        const unsigned long long tw0 = prof_now();
        const int nbk = min(kCB, c_end - k0), k1 = k0 + nbk;
        const int nb2 = min(kCB, c_end - k1);                 // the next diagonal block (0: none -- the last column block of this call)
        const BandRows TR(B, k1);
        constexpr int kDtPass = kCB * kCB / kPT;
        double dtv[kDtPass];
        if (g > 0) {
#pragma unroll
            for (int qd = 0; qd < kDtPass; ++qd) {
                const int idx = tid + qd * kPT, c = idx >> 5, r = idx & 31;
                const bool in = r < nbk && c < nbk && r >= c;
                dtv[qd] = ld_shared(in ? (r == c ? &dinv[k0 + c] : &Lf[B.at32(k0 + r, k0 + c)]) : &dinv[k0]);
            }
        }
        auto write_dt = [&]() {
            if (g > 0) {
#pragma unroll
                for (int qd = 0; qd < kDtPass; ++qd) {
                    const int idx = tid + qd * kPT, c = idx >> 5, r = idx & 31;
                    const bool in = r < nbk && c < nbk;
                    DT[idx] = in ? (r >= c ? dtv[qd] : 0.0) : (r == c ? 1.0 : 0.0);
                }
                __syncthreads();
            }
        };
        const bool tiles_here = G == 1 || g > 0;
        if (tiles_here) {
            const int nti = (TR.R + 63) / 64, ntj = max((TR.R - 1 + 63) / 64, 1);
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
                    bchol_tile(A, Lf, B, TR, k0, nbk, has, by + rem, by, DT, lds + kLdsR + sg * 2 * kLdsPanel,
                               (G > 1 && has && by == 0 && rem == 0) ? k1 + nb2 : 0, gb.prof, write_dt);
                else
                    bchol_tile(A, Lf, B, TR, k0, nbk, has, by + rem, by, DT, lds + kLdsR + sg * 2 * kLdsPanel,
                               (G > 1 && has && by == 0 && rem == 0) ? k1 + nb2 : 0, gb.prof, [] {});
            }
            if (total == 0) write_dt();
        }
        if (g == 0 && nb2 > 0) {
            if (G == 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                for (int idx = tid; idx < kCB * kCB; idx += kPT) {
                    const int r = idx % kCB, c = idx / kCB;
                    Dn[r][c] = (r < nb2 && c < nb2 && r >= c) ? ld_shared(&A[B.at32(k1 + r, k1 + c)]) : (r == c ? 1.0 : 0.0);
                }
            } else {
                // rows k1 .. k1+nb2 of block column k0 against the block, then the update of the next diagonal block with
                // them: the operations bchol_tile applies, in its order (rows k1 .. k1+31 are inside the profile of every
                // column of the block: W >= 64)
                unsigned long long tl0 = prof_now();
                constexpr int kTri = kCB * (kCB + 1) / 2, kTriPass = (kTri + kPT - 1) / kPT;
                double oldv[kTriPass];
                int er[kTriPass], es[kTriPass];
#pragma unroll
                for (int q = 0; q < kTriPass; ++q) {
                    const int idx = tid + q * kPT;
                    int r = (int)((sqrtf(8.0f * idx + 1.0f) - 1.0f) * 0.5f);
                    if (r * (r + 1) / 2 > idx) --r;
                    if ((r + 1) * (r + 2) / 2 <= idx) ++r;
                    er[q] = r; es[q] = idx - r * (r + 1) / 2;
                    const bool in = idx < kTri && r < nb2;
                    oldv[q] = ld_shared(&A[in ? B.at32(k1 + r, k1 + es[q]) : 0u]);
                }
                if (tid < 128) {
                    constexpr int LPR = 4, NS = kCB / LPR;
                    const int lrow = tid >> 2, q = tid & 3, prow = k1 + lrow;
                    const bool pvalid = lrow < nb2;
                    double x[NS];
#pragma unroll
                    for (int s2 = 0; s2 < NS; ++s2) {
                        const int c = s2 * LPR + q;
                        x[s2] = ld_shared(&A[(c < nbk && pvalid) ? B.at32(prow, k0 + c) : 0u]);
                    }
#pragma unroll
                    for (int s2 = 0; s2 < NS; ++s2) x[s2] = ((s2 * LPR + q) < nbk && pvalid) ? x[s2] : 0.0;
                    prof_add(gb.prof, kProfLookLoad, tl0); tl0 = prof_now();
                    trsm32_lanes<LPR>(x, DT, q);
#pragma unroll
                    for (int s2 = 0; s2 < NS; ++s2) Lrow[s2 * LPR + q][lrow] = (s2 * LPR + q) < nbk ? x[s2] : 0.0;
                }
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
            publish(k1, nb2);
            dn_to_dt(nb2);
            prof_add(gb.prof, kProfLookPub, tp0);
        }
        prof_add(gb.prof, kProfFactorWork, tw0);
        const unsigned long long tb0 = prof_now();
        alive = band_barrier(gb, false);
        prof_add(gb.prof, kProfFactorWait, tb0);
        prof_add1(gb.prof, kProfHelpWait, tb0);
        if (gb.prof && threadIdx.x == 0 && blockIdx.x == 0) gb.prof[kProfSteps] += 1;
    }
    return info;
}

// L^T x = y (y = the last dense row of the factor), workgroup 0 only: cluster_persist.hpp::pchol_backsolve on the banded
// layout.  Per block column (from the last): the dots with the already solved unknowns run over the rows of the column's
// profile (BandRows without the right-hand side), their factor entries requested one block column ahead.
// Where unknown i of a system sits in the solution vector of the whole problem (the order of the cluster's loops):
//   mode 0  the system IS the problem;
//   mode 1  first system of a split factorisation: its band unknowns are the problem's first ones, its dense unknowns (the
//           wide loops) sit behind ALL band unknowns of the problem;
//   mode 2  second system: the loops behind the split point in REVERSE order (unknown i = component i % d of loop
//           nlb - 1 - i / d), then the same dense unknowns.
struct BandXMap {
    int mode, nb, d, nlb;
    __device__ __forceinline__ int operator()(int i) const
    {
        if (mode == 0) return i;
        if (i >= nb) return d * nlb + (i - nb);
        return mode == 1 ? i : d * (nlb - 1 - i / d) + i % d;
    }
};

// Two block columns' worth of factor entries are kept in flight (two register buffers, used alternately): a step's
// arithmetic is ~1 us, a trip to memory 3 - 4, and one buffer ahead left every step waiting for its operands.
struct BandBsBuf {
    static constexpr int NW = kPT / 64, CPW = (kCB + NW - 1) / NW, MAXM = 8, DPT = (kCB * kCB + kPT - 1) / kPT;
    double pre[CPW][MAXM], dpre[DPT], ypre;
};
// L^T x = y for the columns c_begin .. c_end-1 of a system (y = the last dense row of the factor), one workgroup:
// cluster_persist.hpp::pchol_backsolve on the banded layout.  The unknowns behind c_end must be solved already (they are
// read from the solution vector through xm).  Per block column (from the last): the dots with the solved unknowns run over
// the rows of the column's profile (BandRows without the right-hand side), their factor entries requested two block columns
// ahead.
__device__ __noinline__ void bband_backsolve(const double* Lf, const BandLayout B, double* x, const BandXMap xm, double* lds, const double* zero,
                                             int c_begin, int c_end)
{
    constexpr int NW = BandBsBuf::NW, CPW = BandBsBuf::CPW, MAXM = BandBsBuf::MAXM, DPT = BandBsBuf::DPT;
    double (*D)[kCB + 1] = reinterpret_cast<double (*)[kCB + 1]>(lds + kLdsD);
    double* t = lds + kLdsDinv;
    double* xs = lds + kLdsR;
    const int n = B.n;
    const bool x_in_lds = n <= kPSG * 2 * kLdsPanel;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l = lane & 31;
    const int nblk = (c_end - c_begin + kCB - 1) / kCB;
    if (nblk <= 0) return;
    if (x_in_lds && c_end < n) {                              // the unknowns an earlier call solved
        for (int i = c_end + tid; i < n; i += kPT) xs[i] = ld_shared(&x[xm(i)]);
        __syncthreads();
    }
    // entry (r, j) sits at j * (ldb - 1) + r for a band row, at j * ldb + (W - nb) + r for a dense row; entries outside
    // the profile are read from a word that holds 0.0, so the dots below need no mask
    auto prefetch = [&](int kb, BandBsBuf& Q) {
        const int k0 = c_begin + kb * kCB, nbk = min(kCB, c_end - k0), k1 = k0 + nbk;
        const BandRows TR(B, k1);
        const int Rm = TR.R - 1;
        unsigned jb[CPW], jd[CPW];
        int jc[CPW];
#pragma unroll
        for (int q = 0; q < CPW; ++q) {
            jc[q] = k0 + min(wave + q * NW, nbk - 1);             // (columns beyond a short block: a valid one, result unused)
            jb[q] = (unsigned)jc[q] * (unsigned)(B.ldb - 1);
            jd[q] = (unsigned)jc[q] * (unsigned)B.ldb + (unsigned)(B.W - B.nb);
        }
#pragma unroll
        for (int m = 0; m < MAXM; ++m) {
            if (64 * m < Rm) {                                // (wave-uniform)
                const int v = lane + 64 * m;
                const int r = TR.row(v < Rm ? v : 0);
                const bool dense = r >= B.nb;
#pragma unroll
                for (int q = 0; q < CPW; ++q) {
                    const bool ok = v < Rm && (dense || r - jc[q] < B.W);
                    Q.pre[q][m] = ld_shared(ok ? &Lf[(dense ? jd[q] : jb[q]) + (unsigned)r] : zero);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < DPT; ++q) {
            const int idx = tid + q * kPT, r = idx % kCB, c = idx / kCB;
            Q.dpre[q] = ld_shared(&Lf[(idx < kCB * kCB && r < nbk && c < nbk && r >= c) ? B.at32(k0 + r, k0 + c) : 0u]);
        }
        Q.ypre = ld_shared(&Lf[l < nbk ? B.at32(n, k0 + l) : 0u]);
    };
    // one block column: the dots with the solved unknowns, then (nxt >= 0) the request for block column nxt into the
    // buffer this step has just emptied, then the triangle
    auto step = [&](int kb, BandBsBuf& Q, int nxt) {
        const int k0 = c_begin + kb * kCB, nbk = min(kCB, c_end - k0), k1 = k0 + nbk;
        const BandRows TR(B, k1);
        const int Rm = TR.R - 1;
        double xr[MAXM];
#pragma unroll
        for (int m = 0; m < MAXM; ++m) {
            xr[m] = 0.0;
            if (64 * m < Rm) {                                // (wave-uniform)
                const int v = lane + 64 * m;
                const int rc = v < Rm ? TR.row(v) : -1;
                const double xv = x_in_lds ? xs[rc >= 0 ? rc : 0] : ld_shared(&x[xm(rc >= 0 ? rc : 0)]);     // (beyond LDS: sc1 both ways, no reliance on this CU's L1)
                xr[m] = rc >= 0 ? xv : 0.0;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < CPW; ++q) {
            const int c = wave + q * NW;
            if (c < nbk) {                                    // (wave-uniform)
                double acc = 0.0;
#pragma unroll
                for (int m = 0; m < MAXM; ++m)
                    if (64 * m < Rm) acc += Q.pre[q][m] * xr[m];
                for (int v = lane + 64 * MAXM; v < Rm; v += 64) {     // (profiles beyond 512 rows)
                    const int r = TR.row(v);
                    if (B.in(r, k0 + c)) acc += ld_shared(&Lf[B.at32(r, k0 + c)]) * (x_in_lds ? xs[r] : ld_shared(&x[xm(r)]));
                }
                acc = wave_sum(acc);
                t[c] = acc;
            }
        }
#pragma unroll
        for (int q = 0; q < DPT; ++q) {
            const int idx = tid + q * kPT, r = idx % kCB, c = idx / kCB;
            if (idx < kCB * kCB) D[r][c] = (r < nbk && c < nbk && r >= c) ? Q.dpre[q] : (r == c ? 1.0 : 0.0);
        }
        const double ycur = Q.ypre;
        if (nxt >= 0) prefetch(nxt, Q);
        __syncthreads();
        if (wave == 0) {
            double v = l < nbk ? ycur - t[l] : 0.0;
            const double dinv = 1.0 / D[l][l];
            double col[kCB];
#pragma unroll
            for (int r = 0; r < kCB; ++r) col[r] = D[r][l];
#pragma unroll
            for (int r = kCB - 1; r >= 0; --r) {
                const double xq = read_lane(v, r) * read_lane(dinv, r);
                v = l == r ? xq : (l < r ? fma(-col[r], xq, v) : v);
            }
            if (lane < nbk) {
                st_shared(&x[xm(k0 + lane)], v);
                if (x_in_lds) xs[k0 + lane] = v;
            }
        }
        if (!x_in_lds) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (x is read back from memory by the other waves when it is not in LDS)
        __syncthreads();
    };
    BandBsBuf Qa, Qb;
    prefetch(nblk - 1, Qa);
    if (nblk >= 2) prefetch(nblk - 2, Qb);
    for (int kb = nblk - 1; kb >= 0; kb -= 2) {
        step(kb, Qa, kb - 2);
        if (kb >= 1) step(kb - 1, Qb, kb - 3);
    }
}

// The banded factorisation + back substitution alone (ipc_debug_band_solve: against a host Cholesky on random systems)
__global__ __launch_bounds__(kPT, 1) void bband_test_kernel(BandArgs Q, double* x, PersistCtl* ctl, int* info)
{
    extern __shared__ double lds[];
    GridBar gb{&ctl->bar, 0u, (int)gridDim.x, &ctl->error, nullptr};
    bool alive = true;
    const int r = bband_factor(Q.A, Q.Lf, Q.dinv, Q.B, gb, (int)blockIdx.x, lds, alive, 0, Q.B.n);
    if (blockIdx.x == 0) {
        bband_backsolve(Q.Lf, Q.B, x, BandXMap{0, 0, 1, 0}, lds, Q.zero, 0, Q.B.n);
        if (threadIdx.x == 0) *info = alive ? r : -1;
    }
}

// ---- the kernel ---------------------------------------------------------------------------------------------------
// Band assembly of pose type T: block (l1, l2) of the capacitance system into the banded layout (unknown index of loop l,
// component r: kD * l + r -- the loops are already in band order)
template <class T>
__device__ __forceinline__ void band_assemble_block(const typename T::Dev& Dv, const BandArgs& Q, int l1, int l2, int r)
{
    constexpr int d = T::kD;
    double* A = Q.A;
    const BandLayout B = Q.B;
    // PLAIN stores, published by the fenced barrier behind the assembly: 8-byte write-through (sc1) stores are one fabric
    // write each, and at 36 per block pair the assembly was bound by them (266 us per iteration at 11 000 pairs).
    // (a diagonal block comes as a full d x d block; the banded layout holds the lower triangle only -- an entry above
    // the diagonal would land in the previous column's dense rows)
    if (Q.split_s < 0) {
        T::assemble_row(Dv, l1, l2, r, [&](int row, int col, double v) { if (row >= col) gptr(A)[B.at32(row, col)] = v; },
                        [&](int col, double v) { gptr(A)[B.at32(B.n, col)] = v; });
        return;
    }
    // split: unknown indices arrive in the order of the loops (band loops by first vertex, then the wide ones); system 1
    // keeps that order for its loops, system 2 holds its band loops reversed
    double* A2 = Q.A2;
    const BandLayout B2 = Q.B2;
    const int nlb = Q.nlb, NB = d * nlb, nb1 = B.nb, ms = d * Q.split_s;       // ms: first unknown of M
    auto to2 = [&](int i) { return i >= NB ? B2.nb + (i - NB) : d * (nlb - 1 - i / d) + i % d; };
    auto to1 = [&](int i) { return i >= NB ? nb1 + (i - NB) : i; };
    T::assemble_row(Dv, l1, l2, r,
        [&](int row, int col, double v) {
            if (row < col) return;
            const bool col1 = col < nb1 || col >= NB;                         // the column's loop lives in system 1 (T, M or wide)
            const bool row1 = row < nb1 || row >= NB;
            if (col1 && row1) { gptr(A)[B.at32(to1(row), to1(col))] = v; return; }
            if (col < ms) return;                                             // (a loop behind M against one in front of it: no overlap, never stored)
            const int a = to2(row), b = to2(col);                            // system 2: reversed, so the later loop has the smaller index
            gptr(A2)[B2.at32(max(a, b), min(a, b))] = v;
        },
        [&](int col, double v) {
            if (col < nb1 || col >= NB) gptr(A)[B.at32(B.n, to1(col))] = v;
            else gptr(A2)[B2.at32(B2.n, to2(col))] = v;
        });
}

template <class T>
__global__ __launch_bounds__(kPT, 1) void cluster_band_kernel(typename T::Dev D0, typename T::Dev D1, PersistArgs P, BandArgs Q)
{
    using Dev = typename T::Dev;
    constexpr size_t kView1 = (sizeof(Dev) + alignof(Dev) - 1) / alignof(Dev) * alignof(Dev);
    const __attribute__((address_space(4))) char* kargs = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    int vsel = 0;
    auto view = [&](int sel) {
        Dev d;
        __builtin_memcpy(&d, kargs + (__builtin_amdgcn_readfirstlane(sel) ? kView1 : 0), sizeof(Dev));
        return d;
    };
    (void)D0; (void)D1;
    extern __shared__ double lds[];
    const int tid = threadIdx.x, g = blockIdx.x, G = gridDim.x;
    const int Gc = G;                                         // workgroups that share the chain phases (all of them)
    GridBar gb{&P.ctl->bar, 0u, G, &P.ctl->error, P.prof};
    const int L = D0.L, nl = D0.nl, ld = D0.ld;
    constexpr int d = T::kD;
    const BandLayout B = Q.B;
    const int n = B.n;
    bool alive = true;
    const unsigned long long tk0 = prof_now();
    const int nidx = L + nl + 1, nblk = (nidx + 255) / 256;
    int n_commit = 0;
    unsigned red_parity = 0, team_target = 0;                 // (team_target: where this workgroup's team barrier stands, split factorisation)

    if (P.prof && tid == 0) atomicMax(&P.ctl->last_start, (tk0 << 8) | (unsigned long long)(g & 255));
    { const Dev Dv = view(0); band_for(L + 1, Gc, [&](int i) { T::load_initial(Dv, P.src, P.src_ld, i); }); }
    alive = band_barrier(gb, true);
    if (P.prof && tid == 0 && g == 0) {
        const unsigned long long v = __hip_atomic_load(&P.ctl->last_start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long skew = (v >> 8) > tk0 ? (v >> 8) - tk0 : 0ull;
        P.prof[kProfStartSkew] += skew;
        if (skew > 100000ull) P.prof[kProfStartSkewXcd01] += skew;      // (launches whose last workgroup came more than 1 ms late)
        P.prof[kProfStartLaunches] += 1;
        if (skew > 100000ull) P.prof[kProfHandoff] += 100;                // (their number, x 1 us)
    }
    prof_add(P.prof, kProfRest, tk0);                         // ("rest": the start -- initial poses and the first barrier, i.e. until the last workgroup of the launch has a CU)

    auto evaluate = [&](bool trial) {
        double tot[1];
        const Dev Dv = view(vsel);
        alive = band_reduce<1>(nblk, lds, Q.gpart, red_parity, gb, Gc, tot, [&](int i, double (&v)[1]) { T::eval(Dv, trial, i, v); }) && alive;
        return tot[0];
    };
    auto assemble = [&](const Dev& Dv) {
        // band blocks (l1, l1 - o), o = 0 .. bwb; the wide loops' rows against every loop; the entries of the profile that
        // no block covers (a column's last kD - 1 - c band rows, and the padding up to W) are zero and stay zero
        const int nlb = Q.nlb, bw1 = Q.bwb + 1;
        const long nband = (long)nlb * bw1, nwide = (long)(nl - nlb) * nl;
        const int Ga = G;
        constexpr int AR = T::kAsmRows;                       // units of work per block pair (SE3: one per block row)
        for (long u = (long)g * kPT + tid; u < (nband + nwide) * AR; u += (long)Ga * kPT) {
            const long q = u / AR;
            const int r = (int)(u - q * AR);
            int l1, l2;
            if (q < nband) { l1 = (int)(q / bw1); l2 = l1 - (int)(q - (long)l1 * bw1); }
            else { const long w = q - nband; l1 = nlb + (int)(w / nl); l2 = (int)(w - (long)(l1 - nlb) * nl); }
            if (l2 >= 0 && l2 <= l1) band_assemble_block<T>(Dv, Q, l1, l2, r);
        }
        const int covered = d * bw1;
        for (int j = g * kPT + tid; j < B.nb; j += Ga * kPT) {
            const int c = j % d;
            for (int o = covered - c; o < B.W; ++o) gptr(Q.A)[(size_t)j * B.ldb + o] = 0.0;
        }
        if (Q.split_s >= 0) {
            // system 2: the same for the columns of its own loops; the columns of M (reversed) and of the wide loops hold
            // nothing of the system itself -- they collect the Schur complement of the elimination and start from zero
            const BandLayout B2 = Q.B2;
            const int own = B2.nb - B2.W;
            for (int j = g * kPT + tid; j < own; j += Ga * kPT) {
                const int c = j % d;
                for (int o = covered - c; o < B2.W; ++o) gptr(Q.A2)[(size_t)j * B2.ldb + o] = 0.0;
            }
            const long zn = (long)(B2.n - own) * B2.ldb;
            double* z0 = Q.A2 + (size_t)own * B2.ldb;
            for (long q = (long)g * kPT + tid; q < zn; q += (long)Ga * kPT) gptr(z0)[q] = 0.0;
        }
    };
    auto linearize = [&](double& bb, double& bHb, double& hh, double& bh) {
        unsigned long long t0 = prof_now();
        { const Dev Dv = view(vsel); band_for(nidx, Gc, [&](int i) { T::force(Dv, i); }); }
        alive = band_barrier(gb, true) && alive;
        { const Dev Dv = view(vsel); double tot[1]; alive = band_reduce<1>(nblk, lds, Q.gpart, red_parity, gb, Gc, tot, [&](int i, double (&v)[1]) { T::b(Dv, i, v); }) && alive; bb = tot[0]; }
        { const Dev Dv = view(vsel); double tot[1]; alive = band_reduce<1>(nblk, lds, Q.gpart, red_parity, gb, Gc, tot, [&](int i, double (&v)[1]) { T::bHb_psi(Dv, i, v); }) && alive; bHb = tot[0]; }
        { const Dev Dv = view(vsel); alive = band_scan(Dv.ps, T::kNPS, L, ld, lds, Q.gscan, gb, Gc) && alive; }
        prof_add(P.prof, kProfPre, t0); t0 = prof_now();
        { const Dev Dv = view(vsel); assemble(Dv); }
        alive = band_barrier(gb, true) && alive;             // (release / acquire: the system was written with plain stores)
        prof_add(P.prof, kProfAssemble, t0); t0 = prof_now();
        int info = 0;
        if (Q.split_s < 0) {
            if (alive) info = bband_factor(Q.A, Q.Lf, Q.dinv, B, gb, g, lds, alive, 0, n);
            prof_add(P.prof, kProfFactor, t0); t0 = prof_now();
            if (g == 0 && alive) {
                const Dev Dv = view(vsel);
                bband_backsolve(Q.Lf, B, Dv.rhs, BandXMap{0, 0, 1, 0}, lds, Q.zero, 0, n);
            }
        } else {
            // Split factorisation: two teams eliminate the loops in front of / behind M at the same time (system 2 holds its
            // loops in reverse order, so both run the same top-down band Cholesky), system 2's Schur complement on M and the
            // wide loops is added to system 1's, team 0 finishes system 1; the back substitution runs M + wide first, then
            // both halves side by side.  Half the dependent block columns of the single chain.
            const BandLayout B2 = Q.B2;
            const int Gt = G / 2, team = g / Gt, gt = g - team * Gt;
            const int ms = d * Q.split_s, own2 = B2.nb - B2.W;
            GridBar tb{&Q.team_bar[team], team_target, Gt, gb.error, team == 0 ? P.prof : nullptr};
            int inf = 0;
            if (alive) inf = team == 0 ? bband_factor(Q.A, Q.Lf, Q.dinv, B, tb, gt, lds, alive, 0, ms)
                                       : bband_factor(Q.A2, Q.Lf2, Q.dinv2, B2, tb, gt, lds, alive, 0, own2);
            if (team == 1 && gt == 0 && tid == 0) __hip_atomic_store(&P.ctl->cmd, inf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            alive = band_barrier(gb, false) && alive;
            {
                const int nc = B2.n - own2, nr = nc + 1;                  // columns of M (reversed) + wide; rows: the same + the right-hand side
                auto to1 = [&](int i2) {                                 // an unknown of system 2 behind its own loops, in system 1
                    if (i2 == B2.n) return B.n;
                    if (i2 >= B2.nb) return B.nb + (i2 - B2.nb);
                    return d * (Q.nlb - 1 - i2 / d) + i2 % d;
                };
                for (int q = g * kPT + tid; q < nc * nr; q += G * kPT) {
                    const int cb = q / nr, ri = q - cb * nr;
                    if (ri < cb) continue;
                    const int i2 = own2 + ri, j2 = own2 + cb;
                    const double u = ld_shared(&Q.A2[B2.at32(i2, j2)]);
                    const int a = to1(i2), b = to1(j2);
                    double* dst = &Q.A[B.at32(max(a, b), min(a, b))];
                    st_shared(dst, ld_shared(dst) + u);
                }
            }
            alive = band_barrier(gb, false) && alive;
            int inf3 = 0;
            if (team == 0 && alive) inf3 = bband_factor(Q.A, Q.Lf, Q.dinv, B, tb, gt, lds, alive, ms, n);
            team_target = tb.target;
            if (g == 0) {
                const int inf2 = __hip_atomic_load(&P.ctl->cmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                info = inf ? inf : (inf2 ? inf2 : inf3);
            }
            prof_add(P.prof, kProfFactor, t0); t0 = prof_now();
            const BandXMap xm1{1, B.nb, d, Q.nlb}, xm2{2, B2.nb, d, Q.nlb};
            if (g == 0 && alive) {
                const Dev Dv = view(vsel);
                bband_backsolve(Q.Lf, B, Dv.rhs, xm1, lds, Q.zero, ms, n);
            }
            alive = band_barrier(gb, false) && alive;
            if (g == 0 && alive) {
                const Dev Dv = view(vsel);
                bband_backsolve(Q.Lf, B, Dv.rhs, xm1, lds, Q.zero, 0, ms);
            } else if (g == Gt && alive) {
                const Dev Dv = view(vsel);
                bband_backsolve(Q.Lf2, B2, Dv.rhs, xm2, lds, Q.zero, 0, own2);
            }
        }
        if (g == 0 && tid == 0) __hip_atomic_store(&P.ctl->le_sel, info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (the solver's word, for everyone)
        alive = band_barrier(gb, true) && alive;
        info = __hip_atomic_load(&P.ctl->le_sel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        prof_add(P.prof, kProfBacksolve, t0); t0 = prof_now();
        { const Dev Dv = view(vsel); band_for(nl, Gc, [&](int l) { T::nu(Dv, l); }); }
        alive = band_barrier(gb, true) && alive;
        { const Dev Dv = view(vsel); band_for(L + 2, Gc, [&](int j) { T::events(Dv, j); }); }
        alive = band_barrier(gb, true) && alive;
        { const Dev Dv = view(vsel); alive = band_scan(Dv.nd, T::kNND, L, ld, lds, Q.gscan, gb, Gc) && alive; }
        { const Dev Dv = view(vsel); band_for(nidx, Gc, [&](int i) { T::rho(Dv, i); }); }
        alive = band_barrier(gb, true) && alive;
        { const Dev Dv = view(vsel); alive = band_scan(Dv.sc, T::kSC1, L, ld, lds, Q.gscan, gb, Gc) && alive; }
        { const Dev Dv = view(vsel); band_for(nidx, Gc, [&](int i) { T::term(Dv, i); }); }
        alive = band_barrier(gb, true) && alive;
        { const Dev Dv = view(vsel); alive = band_scan(Dv.sc + (size_t)T::kSC1 * ld, T::kSC2, L, ld, lds, Q.gscan, gb, Gc) && alive; }
        { const Dev Dv = view(vsel); double tot[2]; alive = band_reduce<2>(nblk, lds, Q.gpart, red_parity, gb, Gc, tot, [&](int i, double (&v)[2]) { T::h(Dv, i, v); }) && alive; hh = tot[0]; bh = tot[1]; }
        prof_add(P.prof, kProfPost, t0);
        if (P.prof && tid == 0 && g == 0) P.prof[kProfIterations] += 1;
        return info;
    };

    // g2o OptimizationAlgorithmDogleg::solve / SparseOptimizer::optimize as cluster_persist_kernel runs it; every
    // workgroup takes the same branches (its scalars are the same bits everywhere)
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
        // the host's abort word, read by workgroup 0 and published in front of this iteration's first barrier (every
        // workgroup must take the same way out)
        if (P.abort_word) {
            if (g == 0 && tid == 0) {
                const int a = __hip_atomic_load(P.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= P.launch_id ? it + 1 : 0;
                __hip_atomic_store(Q.abort_seen, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            alive = band_barrier(gb, false) && alive;
            if (__hip_atomic_load(Q.abort_seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == it + 1) { aborted = true; break; }
        }
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
                { const Dev Dv = view(vsel); alive = band_reduce<2>(nblk, lds, Q.gpart, red_parity, gb, Gc, tot, [&](int i, double (&v)[2]) { T::blend(Dv, alpha, i, v); }) && alive; }
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
            { const Dev Dv = view(vsel); alive = band_reduce<1>(nblk, lds, Q.gpart, red_parity, gb, Gc, changed, [&](int i, double (&v)[1]) { T::update(Dv, pcoef, qcoef, i, v); }) && alive; }
            const bool anyChanged = changed[0] != 0.0;
            const double newChi = evaluate(true);
            if (!alive) break;
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
        if (!alive) break;
        lastGN = goodStep && numTries == 1 && hgnNorm < deltaAtEntry;
        o.iterations = it + 1;
        o.tries += numTries;
        if (numTries == maxTrials || !goodStep) { o.flags |= 1; break; }
    }
    }
    // per-edge chi2 of the committed state, their maximum on workgroup 0
    { const Dev Dv = view(vsel); band_for(nidx, Gc, [&](int i) { T::chi_edges(Dv, i); }); }
    if (alive) alive = band_barrier(gb, true);
    if (g != 0) return;
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
            double mm = 0.0, f = 0.0;
            for (int w = 0; w < kPT / 64; ++w) { mm = fmax(mm, red[w]); f += red[16 + w]; }
            o.max_chi2 = (f != 0.0 && !(mm > 0.0)) ? __builtin_nan("") : mm;
            o.chi2_total = currentChi;
            o.x_sel = n_commit & 1;
            o.error = !alive ? 1 : (aborted ? 2 : 0);
            o.device_ticks = (int)(prof_now() - tk0);
            *P.out = o;
            prof_add(P.prof, kProfTotal, tk0);
        }
    }
}

// ---- host: the band structure of a cluster ---------------------------------------------------------------------------
// a[l], b[l]: first / last vertex of loop l (any common origin).  order: the loops by first vertex with the wide ones
// (the k_wide longest spans) moved to the end; bwb: the largest number of band loops behind a band loop that overlap it.
struct BandPlan {
    bool use = false;
    int nlb = 0, bwb = 0;
    std::vector<int> order;
};
inline int band_halfwidth(const std::vector<int>& a_sorted, const std::vector<int>& b_sorted)
{
    int bw = 0;
    const int n = (int)a_sorted.size();
    for (int p = 0; p < n; ++p) {
        // loops behind p that start in front of p's last vertex (strictly: a shared end vertex is no overlap,
        // reference src/consensus.cpp:157-159 and gk_assemble_core's bq > a)
        const int last = (int)(std::lower_bound(a_sorted.begin() + p + 1, a_sorted.end(), b_sorted[p]) - a_sorted.begin()) - 1;
        bw = std::max(bw, last - p);
    }
    return bw;
}
inline BandPlan band_plan(int d, const std::vector<int>& a, const std::vector<int>& b, int min_n)
{
    BandPlan P;
    const int nl = (int)a.size(), n = d * nl;
    const bool force = min_n == 0;                             // (tests: every cluster of two or more loops through the band kernel)
    if (nl < 2 || n < min_n) return P;
    auto before = [&](int x, int y) { return a[x] != a[y] ? a[x] < a[y] : (b[x] != b[y] ? b[x] < b[y] : x < y); };
    std::vector<int> idx(nl);
    for (int l = 0; l < nl; ++l) idx[l] = l;
    // (cluster_of hands large clusters over sorted by first vertex, the candidate last: one insertion instead of a sort)
    bool presorted = true;
    for (int l = 0; l + 2 < nl && presorted; ++l) presorted = !before(l + 1, l);
    if (presorted) {
        const auto it = std::upper_bound(idx.begin(), idx.end() - 1, nl - 1, before);
        std::rotate(it, idx.end() - 1, idx.end());
    } else {
        std::sort(idx.begin(), idx.end(), before);
    }
    std::vector<int> as(nl), bs(nl), reach(nl);
    for (int p = 0; p < nl; ++p) { as[p] = a[idx[p]]; bs[p] = b[idx[p]]; }
    // loops behind p that start in front of p's last vertex (strictly: a shared end vertex is no overlap, reference
    // src/consensus.cpp:157-159 and gk_assemble_core's bq > a)
    for (int p = 0; p < nl; ++p)
        reach[p] = (int)(std::lower_bound(as.begin() + p + 1, as.end(), bs[p]) - as.begin()) - 1 - p;
    // the wide loops: removing the k loops of largest reach leaves a band of at most the (k+1)-th largest reach
    const int kmax = std::min(nl - 1, 96);
    std::vector<int> top(nl);
    for (int p = 0; p < nl; ++p) top[p] = p;
    std::partial_sort(top.begin(), top.begin() + kmax + 1, top.end(), [&](int x, int y) { return reach[x] != reach[y] ? reach[x] > reach[y] : x < y; });
    const int minblocks = (64 + d - 1) / d;                    // W >= 64
    int best_k = 0;
    double best = 0.0;
    for (int k = 0; k <= kmax; ++k) {
        const int bwb = std::max(reach[top[k]], minblocks - 1);
        const double nb = (double)d * (nl - k), W = (double)d * (bwb + 1), m = (double)d * k + 1;
        const double c = nb * (W + m) * (W + m) + m * m * m / 3.0;
        if (k == 0 || c < 0.98 * best) { best = c; best_k = k; }
    }
    std::vector<char> wide(nl, 0);
    for (int q = 0; q < best_k; ++q) wide[top[q]] = 1;
    // the exact half-bandwidth of what is left
    std::vector<int> ra, rb;
    ra.reserve(nl); rb.reserve(nl);
    for (int p = 0; p < nl; ++p) if (!wide[p]) { ra.push_back(as[p]); rb.push_back(bs[p]); }
    const int bwb = std::max(band_halfwidth(ra, rb), minblocks - 1);
    const double W = (double)d * (bwb + 1), m = (double)d * best_k + 1;
    if (!force && (W + m) > 0.6 * n) return P;                 // (banded only where it is clearly narrower than the dense system)
    P.use = true;
    P.bwb = bwb;
    P.nlb = nl - best_k;
    P.order.reserve(nl);
    for (int p = 0; p < nl; ++p) if (!wide[p]) P.order.push_back(idx[p]);
    for (int p = 0; p < nl; ++p) if (wide[p]) P.order.push_back(idx[p]);
    return P;
}

}  // namespace ipc
