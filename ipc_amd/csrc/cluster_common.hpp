// Shared pieces of the cluster solvers (cluster_se2.hpp, cluster_se3.hpp): grid-kernel reduction
// helpers, the one-workgroup prefix sum, the loop tables and the host-side dog-leg control flow
// (g2o OptimizationAlgorithmDogleg::solve with the analytic gains and exact shortcuts of the cell
// kernels).  The pose-type specific work hides behind an `Ops` object.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "block_prims.hpp"
#include "dense_chol.hpp"

namespace ipc {

#define IPC_CL_CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

constexpr int kGB = 256;                     // threads per block of the grid kernels

// Loads / stores of data that ANOTHER workgroup writes / reads inside one launch (the persistent cluster
// kernel, cluster_persist.hpp): agent-scope relaxed atomics = global_load / global_store ... sc1, which bypass the
// CU's L1 and write through the XCD's L2 (per-XCD L2s are not coherent with each other), so that no cache
// write-back / invalidate is needed around the grid barriers.  In the one-kernel-per-phase path they cost nothing.
// A pointer known to point into global memory.  The cluster solvers' arrays hang off a struct; where that struct is read
// through a pointer (the persistent kernel switches between two views of it) the compiler no longer knows the address
// space of its members and emits flat loads, which count against both vmcnt and lgkmcnt and so can only be waited for
// all at once.  Indexed through this cast they are global loads / stores again.
template <class T>
__device__ __forceinline__ __attribute__((address_space(1))) T* gptr(T* p) { return (__attribute__((address_space(1))) T*)p; }

__device__ __forceinline__ double ld_shared(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_shared(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Block sum of K per-thread values (blocks of kGB = 256 threads): a DPP scan inside each wave, then the four wave
// totals in wave order.  The persistent kernel (cluster_persist.hpp::lead_reduce) adds in exactly this order.
template <int K>
__device__ __forceinline__ void gk_block_reduce_store(double (&v)[K], double* partial_row)
{
    __shared__ double shm[K][kGB / 64];
    const int t = threadIdx.x;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const double ws = wave_sum(v[k]);
        if ((t & 63) == 0) shm[k][t >> 6] = ws;
    }
    __syncthreads();
    if (t == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) partial_row[k] = ((shm[k][0] + shm[k][1]) + shm[k][2]) + shm[k][3];
    }
}

// sum of the per-block partials in block order (deterministic) -> scal[off + k]
__global__ void gk_sum(const double* partial, int nblocks, int K, double* scal, int off)
{
    const int k = threadIdx.x;
    if (k >= K) return;
    double acc = 0.0;
    for (int bq = 0; bq < nblocks; ++bq) acc += partial[bq * 4 + k];
    scal[off + k] = acc;
}

// in-place inclusive prefix sums over indices 1..L of K arrays (row length ld), one workgroup per array
__global__ __launch_bounds__(1024) void gk_scan(double* arr, int K, int L, int ld)
{
    __shared__ double wsum[16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int k = blockIdx.x; k < K; k += gridDim.x) {
        double* a = arr + (size_t)k * ld;
        double carry = 0.0;
        for (int base = 1; base <= L; base += 1024) {
            const int i = base + tid;
            double v = i <= L ? a[i] : 0.0;
            v = wave_inclusive_scan(v);
            if (lane == 63) wsum[wave] = v;
            __syncthreads();
            double off = carry;
            for (int w = 0; w < wave; ++w) off += wsum[w];
            double tot = carry;
            for (int w = 0; w < 16; ++w) tot += wsum[w];
            if (i <= L) a[i] = v + off;
            carry = tot;
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct ClusterOut {
    double max_chi2 = 0.0, chi2_total = 0.0, chi2_initial = 0.0;
    int iterations = 0, tries = 0, flags = 0, evals = 0;    // flags: 1 terminated, 2 solve failed
};


// Loop tables of a cluster in one int array (host layout == device layout):
//   lfrom[nl] lto[nl] lcand[nl]  local end points / candidate record index
//   adj_ptr[L+2] adj_item[2 nl]  per pose p: items l*2 + role (0 = from, 1 = to)
//   ev_ptr[L+3]  ev_item[2 nl]   per index j: items l*2 + kind (0 = range start, 1 = one past its end)
struct LoopTables {
    std::vector<int> host;
    int L = 0, nl = 0;
    size_t size() const { return host.size(); }
    static size_t capacity(int L, int nl) { return 7 * (size_t)nl + 2 * ((size_t)L + 2) + 8; }
    void build(int lo, int hi, const std::vector<int>& members, const int* from, const int* to)
    {
        L = hi - lo; nl = (int)members.size();
        host.assign(capacity(L, nl), 0);
        int* lf = host.data();
        int* lt = lf + nl;
        int* lc = lt + nl;
        int* adj_ptr = lc + nl;
        int* adj_item = adj_ptr + (L + 2);
        int* ev_ptr = adj_item + 2 * nl;
        int* ev_item = ev_ptr + (L + 3);
        for (int l = 0; l < nl; ++l) {
            lf[l] = from[members[l]] - lo; lt[l] = to[members[l]] - lo; lc[l] = members[l];
            ++adj_ptr[lf[l] + 1]; ++adj_ptr[lt[l] + 1];
            const int a = std::min(lf[l], lt[l]), b = std::max(lf[l], lt[l]);
            ++ev_ptr[a + 1 + 1]; ++ev_ptr[b + 1 + 1];
        }
        for (int j = 0; j <= L; ++j) adj_ptr[j + 1] += adj_ptr[j];
        for (int j = 0; j <= L + 1; ++j) ev_ptr[j + 1] += ev_ptr[j];
        std::vector<int> ca(adj_ptr, adj_ptr + L + 1), ce(ev_ptr, ev_ptr + L + 2);
        for (int l = 0; l < nl; ++l) {
            adj_item[ca[lf[l]]++] = 2 * l;
            adj_item[ca[lt[l]]++] = 2 * l + 1;
            const int a = std::min(lf[l], lt[l]), b = std::max(lf[l], lt[l]);
            ev_item[ce[a + 1]++] = 2 * l;
            ev_item[ce[b + 1]++] = 2 * l + 1;
        }
    }
    // device pointers into a copy of `host` at d
    const int* lfrom(const int* d) const { return d; }
    const int* lto(const int* d) const { return d + nl; }
    const int* lcand(const int* d) const { return d + 2 * nl; }
    const int* adj_ptr(const int* d) const { return d + 3 * nl; }
    const int* adj_item(const int* d) const { return adj_ptr(d) + (L + 2); }
    const int* ev_ptr(const int* d) const { return adj_item(d) + 2 * nl; }
    const int* ev_item(const int* d) const { return ev_ptr(d) + (L + 3); }
};

// ---- banded storage of an SPD system with a dense border (cluster_band.hpp has the factorisation) -----------------
struct BandLayout {
    int nb;          // band unknowns: columns 0 .. nb-1
    int m;           // dense rows: the wide loops' unknowns (m - 1) and the right-hand side (last)
    int W;           // stored band rows per column (row i of column j at offset i - j < W), >= 64
    int ldb;         // column stride = W + m
    int n;           // unknowns = nb + m - 1; rows 0 .. n
    __host__ __device__ __forceinline__ size_t at(int i, int j) const { return (size_t)j * ldb + (i < nb ? i - j : W + (i - nb)); }
    // is entry (i, j), i >= j, inside the stored profile?
    __host__ __device__ __forceinline__ bool in(int i, int j) const { return i >= nb || i - j < W; }
    // the same address in 32 bits (a system holds < 2^31 doubles): one v_mad_u32 instead of a 64-bit multiply
    __device__ __forceinline__ unsigned at32(int i, int j) const { return (unsigned)j * (unsigned)ldb + (unsigned)(i < nb ? i - j : W + (i - nb)); }
    __host__ __device__ size_t doubles() const { return (size_t)(n > 0 ? n : 1) * ldb + 64; }
};

struct PersistCtl;
// factorisation + back substitution of one banded + bordered system on `workgroups` workgroups (cluster_band.hpp,
// bband_test_kernel); *info: 0, or 1 + the first column of the block column whose pivot was not positive.  A / Lf:
// B.doubles() each, dinv / x: B.n + 64 doubles, zero: a device word that holds 0.0
inline hipError_t band_system_solve(const BandLayout& B, double* A, double* Lf, double* dinv, double* x, PersistCtl* ctl, int* info,
                                    const double* zero, int workgroups, hipStream_t st);

// ---- where the literal normal equations of the Levenberg retry are stored ------------------------------------------
// g2o's own H (block tridiagonal from the chain + one off-diagonal block per loop with two free ends) is what carries the
// damping (cluster_dogleg below).  Dense (rounds 3 - 5): (n + 1) x n column major, right-hand side in row n -- d L <= 24 000
// unknowns.  Banded (round 6): the poses in chain order are a band of half-width (longest ordinary loop span + 1) blocks;
// the later end of every loop that spans more goes to the dense border.  n (W + m) doubles instead of n^2: C5's clusters
// (34 000 poses = 204 000 unknowns, spans <= 200) are 2 GB instead of 333 GB, and C4's (15 000 unknowns) factor in
// n W^2 = 1.4e9 operations instead of n^3 / 3 = 1.1e12.
struct DenseStore {
    double* A; int nn, d;                                   // nn unknowns
    __device__ __forceinline__ int unk(int p) const { return d * (p - 1); }
    __device__ __forceinline__ int n() const { return nn; }
    __device__ __forceinline__ double* lower(int i, int j) const { return &A[(size_t)j * (nn + 1) + i]; }     // i >= j; i == n: rhs
};
struct BandStore {
    double* A; BandLayout B; const int* ublk; int d;        // ublk[p]: block of pose p in the band order (border poses last)
    __device__ __forceinline__ int unk(int p) const { return d * ublk[p]; }
    __device__ __forceinline__ int n() const { return B.n; }
    __device__ __forceinline__ double* lower(int i, int j) const { return &A[B.at(i, j)]; }
};
// x in the store's unknown order -> [d (p - 1) + k], the order gk*_h_from_dense reads
__global__ void gk_unpermute_blocks(const double* x, const int* ublk, int d, int L, double* out)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < 1 || p > L) return;
    for (int k = 0; k < d; ++k) out[(size_t)d * (p - 1) + k] = x[(size_t)d * ublk[p] + k];
}

// The band structure of the pose-space system: lf / lt = local end poses of the loops (0 = the gauge, fixed: no block).
// S = the largest span that stays in the band; loops that span more send their later end to the border.  Chosen to make
// L (S + 1 + border poses)^2 small; use == false: not worth it (the dense store is as good or the border would be huge).
struct PoseBandPlan {
    bool use = false;
    int S = 0, nbp = 0, nborder = 0;
    std::vector<int> ublk;                                  // [L + 1]
};
inline PoseBandPlan pose_band_plan(int d, int L, int nl, const int* lf, const int* lt, int max_border = 64)
{
    PoseBandPlan P;
    std::vector<int> span;
    span.reserve(nl);
    for (int l = 0; l < nl; ++l)
        if (lf[l] >= 1 && lt[l] >= 1 && lf[l] != lt[l]) span.push_back(std::abs(lt[l] - lf[l]));
    std::vector<int> s(span);
    std::sort(s.begin(), s.end());
    // candidates: keep every loop (S = largest span), or cut behind the k widest
    double best = -1.0;
    int bestS = 1;
    const int ns = (int)s.size();
    for (int k = 0; k <= std::min(ns, max_border); ++k) {
        const int S = std::max(1, k < ns ? s[ns - 1 - k] : 1);
        const double w = (double)(S + 1) + k;                // (at most k border poses)
        const double cost = (double)L * w * w;
        if (best < 0 || cost < best) { best = cost; bestS = S; }
    }
    P.S = bestS;
    std::vector<char> border(L + 1, 0);
    for (int l = 0; l < nl; ++l)
        if (lf[l] >= 1 && lt[l] >= 1 && std::abs(lt[l] - lf[l]) > P.S) border[std::max(lf[l], lt[l])] = 1;
    P.ublk.assign(L + 1, 0);
    int nb = 0;
    for (int p = 1; p <= L; ++p) if (!border[p]) P.ublk[p] = nb++;
    P.nbp = nb;
    for (int p = 1; p <= L; ++p) if (border[p]) P.ublk[p] = nb++;
    P.nborder = L - P.nbp;
    const double wd = (double)d * (P.S + 1) + (double)d * P.nborder;
    P.use = P.nbp >= 1 && wd * 3.0 <= (double)d * L;         // (a third of the dense width at most: otherwise the dense store)
    return P;
}

struct LiteralBand;                                         // cluster_literal_band.hpp
inline void literal_band_free(LiteralBand* p);
template <class Solver>
hipError_t literal_band_damped(Solver& S, double lambda, bool& used, double* x_out, int* d_info);

// Ops:  evaluate_committed(chi) | linearize(bb, bHb, hh, bh, info) | blend(alpha, c, bma)
//       | trial(p, q, newChi, anyChanged) | commit() | max_edge_chi2(mx)
//       | damped_solve(lambda, ok, hh, bh, bHh, hHh)   (H + lambda I) h = b on the literal normal equations
//
// Levenberg retry of the linear solve (g2o OptimizationAlgorithmDogleg::solve, as the tests' CPU restatement follows it): the Gauss-Newton step is first asked of the plain system; once a factorisation has met a
// non-positive pivot (`wasPD` false, sticky for the rest of the optimisation) every solve adds currentLambda to the
// diagonal of H -- times 10 per failure up to 1e3, then Fail; divided by 5 (not below 1e-12) per success.  The
// capacitance formulation of the cluster solvers cannot carry lambda (H + lambda I is not chain-structured in the
// u = J_c h variables), so damped solves go to Ops::damped_solve: the dense normal equations of g2o's own H.
template <class Ops>
hipError_t cluster_dogleg(Ops& ops, int iterations, ClusterOut& out, double term_eps = 0.0, int n_edges = 1, bool allow_damping = true)
{
    out = ClusterOut{};
    double currentChi;
    IPC_CL_CHK(ops.evaluate_committed(currentChi));
    out.chi2_initial = currentChi;
    double delta = 1e4, currentLambda = 1e-7;
    const double lambdaFactor = 10.0, minLambda = 1e-12, maxLambda = 1e3;
    const int maxTrials = 100;
    bool lastGN = false, wasPD = true;
    for (int it = 0; it < iterations; ++it) {
        double bb, bHb, hh, bh;
        int info = 0;
        ops.want_plain_solve(wasPD);                  // (once a factorisation has failed every step is a damped one: the capacitance system is not even set up)
        IPC_CL_CHK(ops.linearize(bb, bHb, hh, bh, info));
        double hHh = bh, bHh = bb;                    // H h_gn = b (plain solve)
        {
            bool solverOk = wasPD && info == 0, failed = false, first = true;
            while (!solverOk) {
                if (!first || wasPD) {                // a solve has just failed: g2o's bookkeeping
                    wasPD = false;
                    currentLambda *= lambdaFactor;
                    if (currentLambda > maxLambda) { currentLambda = maxLambda; failed = true; break; }
                }
                first = false;
                if (!allow_damping) { failed = true; break; }
                bool ok = false;
                IPC_CL_CHK(ops.damped_solve(currentLambda, ok, hh, bh, bHh, hHh));
                if (ok) { solverOk = true; currentLambda = std::max(currentLambda / (0.5 * lambdaFactor), minLambda); }
            }
            if (failed) { out.flags |= 2; out.iterations = it + 1; break; }
        }
        const double alpha = bb / bHb, hsdNorm = std::sqrt(alpha * alpha * bb), hgnNorm = std::sqrt(hh);
        if (wasPD && lastGN && hgnNorm < delta && std::fabs(bh) * n_edges < term_eps * currentChi) {   // converged (Se2View::term_eps)
            out.iterations = it + 1; out.tries += maxTrials; out.flags |= 1;
            break;
        }
        const double deltaAtEntry = delta;
        bool goodStep = false;
        int numTries = 0;
        do {
            ++numTries;
            int stepType;
            double beta = 0.0, sdScale = 0.0;
            if (hgnNorm < delta) stepType = 0;
            else if (hsdNorm > delta) { stepType = 1; sdScale = delta / hsdNorm; }
            else {
                stepType = 2;
                double c, bma;
                IPC_CL_CHK(ops.blend(alpha, c, bma));
                const double hsdSq = alpha * alpha * bb;
                if (c <= 0.) beta = (-c + std::sqrt(c * c + bma * (delta * delta - hsdSq))) / bma;
                else beta = (delta * delta - hsdSq) / (c + std::sqrt(c * c + bma * (delta * delta - hsdSq)));
            }
            double pcoef, qcoef, hdlNorm;
            if (stepType == 0) { pcoef = 0.0; qcoef = 1.0; hdlNorm = hgnNorm; }
            else if (stepType == 1) { pcoef = sdScale * alpha; qcoef = 0.0; hdlNorm = delta; }
            else { pcoef = alpha - beta * alpha; qcoef = beta; hdlNorm = delta; }
            const double hdlHhdl = pcoef * pcoef * bHb + 2 * pcoef * qcoef * bHh + qcoef * qcoef * hHh;
            const double bhdl = pcoef * bb + qcoef * bh;
            double linearGain = -1 * hdlHhdl + 2 * bhdl;
            double newChi;
            bool anyChanged;
            IPC_CL_CHK(ops.trial(pcoef, qcoef, newChi, anyChanged));
            ++out.evals;
            const double nonLinearGain = currentChi - newChi;
            if (std::fabs(linearGain) < 1e-12) linearGain = 1e-12;
            const double rho = nonLinearGain / linearGain;
            if (rho > 0) {
                goodStep = true;
                currentChi = newChi;
                ops.commit();
            }
            if (rho > 0.75) delta = std::max(delta, 3 * hdlNorm);
            else if (rho < 0.25) delta *= 0.5;
            if (!goodStep) {
                if (rho != rho) {
                    numTries = maxTrials;       // NaN gain ratio: g2o leaves delta alone, so every retry is this same trial
                } else if (stepType == 0) {
                    while (numTries < maxTrials && hgnNorm < delta) { ++numTries; delta *= 0.5; }
                } else if (stepType == 1 && !anyChanged) {
                    numTries = maxTrials;
                }
            }
        } while (!goodStep && numTries < maxTrials);
        lastGN = goodStep && numTries == 1 && hgnNorm < deltaAtEntry;
        out.iterations = it + 1;
        out.tries += numTries;
        if (numTries == maxTrials || !goodStep) { out.flags |= 1; break; }
    }
    IPC_CL_CHK(ops.max_edge_chi2(out.max_chi2));
    out.chi2_total = currentChi;
    if (!wasPD) out.flags |= 4;                       // the linear solve needed Levenberg damping
    return hipSuccess;
}

}  // namespace ipc
