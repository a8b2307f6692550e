// libipc_amd.so -- C ABI (include/ipc_amd.h) over the gfx950 kernels.
//
// Device-side pipeline of one ipc_solve_rows():
//   k_plan_count / k_plan_fill   enumerate the cells of this rank's rows, bin them by chain
//                                length L into (threads, poses-per-thread) kernel variants
//   se2_cells_kernel<T,M,NL>     one workgroup per cell (se2_cell.hpp)            <- hot kernel
//   k_scatter_bits               cell results -> bit rows of the shard
// then ipc_assemble_matrix (k_assemble) and ipc_set_max (k_set_max).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdlib>
#include <utility>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <map>
#include <mutex>
#include <string>
#include <chrono>
#include <thread>
#include <vector>

#include "../../include/ipc_amd.h"
#include "cell_kernels.hpp"
#include "cluster_se2.hpp"
#include "cluster_se3.hpp"
#include "cluster_persist.hpp"

using namespace ipc;

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(IPC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                               \
    } while (0)

extern "C" const char* ipc_last_error(void) { return g_err.c_str(); }

// ------------------------------------------------------------------------------------------
// kernel variants: (threads per cell, poses per thread); capacity = T*M poses
// ------------------------------------------------------------------------------------------
constexpr int kMaxBins = 32;
struct BinCaps { int n; int cap[kMaxBins]; };

__device__ __forceinline__ int bin_of(const BinCaps& bc, int L)
{
    int b = 0;
    while (b < bc.n && L > bc.cap[b]) ++b;
    return b;                                       // == bc.n => too long
}

// ------------------------------------------------------------------------------------------
// one-off preparation kernels
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void inv_sym3(const double* a, double* o)
{
    // a = (00 01 02 11 12 22)
    const double c00 = a[3] * a[5] - a[4] * a[4];
    const double c01 = a[2] * a[4] - a[1] * a[5];
    const double c02 = a[1] * a[4] - a[2] * a[3];
    const double det = a[0] * c00 + a[1] * c01 + a[2] * c02;
    const double id = 1.0 / det;
    o[0] = c00 * id; o[1] = c01 * id; o[2] = c02 * id;
    o[3] = (a[0] * a[5] - a[2] * a[2]) * id;
    o[4] = (a[1] * a[2] - a[0] * a[4]) * id;
    o[5] = (a[0] * a[3] - a[1] * a[1]) * id;
}

// raw file record -> field-major SE2 record k (robustifyVoters: info *= scale)
__device__ __forceinline__ void se2_prep_record(const double* m, const double* inf, double scale, double unscale, double* rec,
                                                int stride, int k)
{
    const double tx = m[0], ty = m[1], th = m[2];
    double s, c;
    sincos(th, &s, &c);
    rec[(size_t)F_TZX * stride + k] = tx;
    rec[(size_t)F_TZY * stride + k] = ty;
    rec[(size_t)F_CZ * stride + k] = c;
    rec[(size_t)F_SZ * stride + k] = s;
    rec[(size_t)F_THZ * stride + k] = th;
    double om[6], sg[6];
    // unscale != 1: the harness's final map divides the scaled information by s again
    // (reference src/simulation.cpp:55-56), i.e. (info * s) / s, not the file value
    for (int q = 0; q < 6; ++q) om[q] = unscale == 1.0 ? inf[q] * scale : (inf[q] * scale) / unscale;
    inv_sym3(om, sg);
    for (int q = 0; q < 6; ++q) {
        rec[(size_t)(F_OM + q) * stride + k] = om[q];
        rec[(size_t)(F_SG + q) * stride + k] = sg[q];
    }
}
__global__ void k_se2_prep(int n, const double* meas, const double* info, double scale, double* rec, int stride,
                           double unscale = 1.0)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    se2_prep_record(meas + 3 * (size_t)k, info + 6 * (size_t)k, scale, unscale, rec, stride, k);
}

// field-major records -> record-major copy (rows beyond n stay zero)
__global__ void k_records(int n, int nf, const double* rec, int stride, double* out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    for (int f = 0; f < nf; ++f) out[(size_t)k * nf + f] = rec[(size_t)f * stride + k];
}

// propagateGuess (reference src/consensus_utils.cpp:99-116): v0 at origin, v[i] = v[i-1] * z[i-1]
// (SE2::operator*: t += R t2, theta = normalize(theta + theta2)).  Sequential by nature and run
// once per engine, so a single lane walks the chain.
__global__ void k_se2_propagate(int V, const double* rec, int stride, double* pose0)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double x = 0, y = 0, th = 0;
    pose0[0] = x; pose0[V] = y; pose0[2 * (size_t)V] = th;
    for (int i = 1; i < V; ++i) {
        double s, c;
        sincos(th, &s, &c);
        const double tx = rec[(size_t)F_TZX * stride + i - 1], ty = rec[(size_t)F_TZY * stride + i - 1];
        x += c * tx - s * ty;
        y += s * tx + c * ty;
        th = normalize_theta(th + rec[(size_t)F_THZ * stride + i - 1]);
        pose0[i] = x; pose0[(size_t)V + i] = y; pose0[2 * (size_t)V + i] = th;
    }
}

// ---- SE3: EdgeSE3::read semantics (quaternion re-normalised), 6x6 information and its inverse ----
__device__ void inv_sym6(const double* up, double* out)      // 21 upper -> 21 upper of the inverse
{
    double A[6][6], Li[6][6];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) { A[i][j] = up[sym6_idx(i, j)]; Li[i][j] = 0.0; }
    // Cholesky A = L L^T (in the lower triangle of A)
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j <= i; ++j) {
            double sum = A[i][j];
            for (int k = 0; k < j; ++k) sum -= A[i][k] * A[j][k];
            A[i][j] = (j < i) ? sum / A[j][j] : sqrt(sum);
        }
    // Li = L^-1
    for (int c = 0; c < 6; ++c)
        for (int i = c; i < 6; ++i) {
            double sum = (i == c) ? 1.0 : 0.0;
            for (int k = c; k < i; ++k) sum -= A[i][k] * Li[k][c];
            Li[i][c] = sum / A[i][i];
        }
    // inverse = Li^T Li
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j) {
            double sum = 0.0;
            for (int k = j; k < 6; ++k) sum += Li[k][i] * Li[k][j];
            out[sym6_idx(i, j)] = sum;
        }
}

__device__ __forceinline__ void se3_prep_record(const double* m, const double* inf, double scale, double unscale, double* rec,
                                                int stride, int k)
{
    const double qn = sqrt(m[3] * m[3] + m[4] * m[4] + m[5] * m[5] + m[6] * m[6]);
    double R[9];
    R_from_quat(m[6] / qn, m[3] / qn, m[4] / qn, m[5] / qn, R);
    for (int q = 0; q < 9; ++q) rec[(size_t)(G_RZ + q) * stride + k] = R[q];
    for (int q = 0; q < 3; ++q) rec[(size_t)(G_TZ + q) * stride + k] = m[q];
    double om[21], sg[21];
    for (int q = 0; q < 21; ++q) om[q] = unscale == 1.0 ? inf[q] * scale : (inf[q] * scale) / unscale;
    inv_sym6(om, sg);
    for (int q = 0; q < 21; ++q) {
        rec[(size_t)(G_OM + q) * stride + k] = om[q];
        rec[(size_t)(G_SG + q) * stride + k] = sg[q];
    }
}
__global__ void k_se3_prep(int n, const double* meas, const double* info, double scale, double* rec, int stride,
                           double unscale = 1.0)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    se3_prep_record(meas + 7 * (size_t)k, info + 21 * (size_t)k, scale, unscale, rec, stride, k);
}

// ipc_append_candidate: ONE candidate record, handed over in the kernel arguments (no staging buffer, nothing to free, no
// host synchronisation), written to slot k of the candidate arrays by the same per-record code as the batch kernels
struct RawCandidate { double meas[7]; double info[21]; int from, to; };
template <int DIM>
__global__ void k_prep_one(RawCandidate r, int k, double* rec, int stride, int* from, int* to, int* lo, int* hi)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double m[7], inf[21];
    for (int q = 0; q < 7; ++q) m[q] = r.meas[q];
    for (int q = 0; q < 21; ++q) inf[q] = r.info[q];
    if (DIM == 2) se2_prep_record(m, inf, 1.0, 1.0, rec, stride, k);
    else se3_prep_record(m, inf, 1.0, 1.0, rec, stride, k);
    from[k] = r.from; to[k] = r.to;
    lo[k] = r.from < r.to ? r.from : r.to;
    hi[k] = r.from < r.to ? r.to : r.from;
}

// field-major SE3 records -> blocks of 64 edges x kSe3BlkPairs double2 (se3_lds_cell.hpp reads them with
// one coalesced 16-byte load per lane): pairs 0..5 = Rz, tz; 6..16 = information (+ pad); 17..27 = covariance (+ pad)
__global__ void k_se3_blocks(int n, const double* rec, int stride, double2* out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    double2* o = out + (size_t)(k >> 6) * (kSe3BlkPairs * 64) + (k & 63);
    auto F = [&](int f) { return rec[(size_t)f * stride + k]; };
    for (int p = 0; p < 6; ++p) o[p * 64] = make_double2(F(2 * p), F(2 * p + 1));
    for (int p = 0; p < 11; ++p) {
        o[(6 + p) * 64] = make_double2(F(G_OM + 2 * p), 2 * p + 1 < 21 ? F(G_OM + 2 * p + 1) : 0.0);
        o[(17 + p) * 64] = make_double2(F(G_SG + 2 * p), 2 * p + 1 < 21 ? F(G_SG + 2 * p + 1) : 0.0);
    }
}

// propagateGuess for SE3: v0 = identity, v[i] = v[i-1] * z[i-1] (Isometry3 product)
__global__ void k_se3_propagate(int V, const double* rec, int stride, double* pose0)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
    for (int q = 0; q < 9; ++q) pose0[(size_t)q * V] = R[q];
    for (int q = 0; q < 3; ++q) pose0[(size_t)(9 + q) * V] = t[q];
    for (int i = 1; i < V; ++i) {
        double Rz[9], tz[3], Rn[9], d[3];
        for (int q = 0; q < 9; ++q) Rz[q] = rec[(size_t)(G_RZ + q) * stride + i - 1];
        for (int q = 0; q < 3; ++q) tz[q] = rec[(size_t)(G_TZ + q) * stride + i - 1];
        m3_mul(R, Rz, Rn);
        m3_vec(R, tz, d);
        for (int q = 0; q < 3; ++q) t[q] += d[q];
        for (int q = 0; q < 9; ++q) R[q] = Rn[q];
        for (int q = 0; q < 9; ++q) pose0[(size_t)q * V + i] = R[q];
        for (int q = 0; q < 3; ++q) pose0[(size_t)(9 + q) * V + i] = t[q];
    }
}

// ------------------------------------------------------------------------------------------
// planning: which cells does this rank solve, and with which kernel variant
// ------------------------------------------------------------------------------------------
// counts layout: [2][kMaxBins+1][kPlanSub]  (0: diagonal cells, 1: pair cells; last slot = too long)
constexpr int kPlanSub = 32;
// phase 0: every cell of this rank's rows; 1: the diagonal cells only; 2: the pair cells of candidates whose own cell
// passed (diagonal bit set in `diag`, the shard of a one-rank run) -- the set-only mode of ipc_run_set_only
// Rows are visited in the order of `rowperm` (rows sorted by the first vertex of their interval) and a bin's cell list is
// the concatenation of kPlanSub sub-lists, sub-list s holding the rows of the s-th chunk of that order: the list ends up
// sorted by chain position.  The cell kernels draw cells from the head of the list, so the cells in flight at any moment
// are neighbours along the trajectory and their chain records are shared through L2 (C5, 21.6 MB of records against
// 4 MB of L2 per XCD: round 3 fetched every record from beyond L2 in every pass, 35.8x the algorithmic bytes).  Blocks
// that run side by side (consecutive u) land in different chunks, i.e. on different counters.  `rowperm` lists the rows of
// the calling rank only (ensure_row_map: rank-major groups, each sorted by first vertex).
__global__ void k_plan(int N, const int* lo, const int* hi, const int* rowperm, int nrows, BinCaps bc,
                       unsigned* counters, const unsigned* offsets, int2* cells, int fill, int phase = 0,
                       const unsigned long long* diag = nullptr, int words = 0)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;          // this thread's candidate, its interval read once
    const int loj = j < N ? lo[j] : 0, hij = j < N ? hi[j] : 0;
    const bool alivej = phase != 2 || (j < N && ((diag[(size_t)j * words + (j >> 6)] >> (j & 63)) & 1ull));
    const int chunk = (nrows + kPlanSub - 1) / kPlanSub;
    for (int u = blockIdx.y; u < chunk * kPlanSub; u += gridDim.y) {
    const int sub = u & (kPlanSub - 1);
    const int q = sub * chunk + (u / kPlanSub);                   // position in the row order
    if (q >= nrows || (u / kPlanSub) >= chunk) continue;
    const int i = rowperm[q];                                     // (the rows of THIS rank only: the pass costs 1 / world of the matrix's rows)
    if (phase == 2 && !((diag[(size_t)i * words + (i >> 6)] >> (i & 63)) & 1ull)) continue;
    if ((int)((blockIdx.x + 1) * blockDim.x) <= i) continue;       // this block's candidates all precede row i (j < i)
    int slot = -1;
    if (j < N && j >= i) {
        const int loi = lo[i], hii = hi[i];
        if (j == i) { if (phase != 2) slot = bin_of(bc, hii - loi); }
        else if (phase != 1 && alivej) {
            if (min(hii, hij) - max(loi, loj) > 0)            // reference src/consensus.cpp:157-159
                slot = (kMaxBins + 1) + bin_of(bc, max(hii, hij) - min(loi, loj));
        }
    }
    // wave-aggregated append: one atomic per (wave, slot present)
    unsigned long long todo = __ballot(slot >= 0);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int sl = __shfl(slot, leader, 64);
        const unsigned long long same = __ballot(slot == sl);
        const int nsame = __popcll(same);
        const int lane = threadIdx.x & 63;
        unsigned base = 0;
        // kPlanSub sub-counters per slot (chosen by block): 377 000 appends on ~20 addresses were 3.4 ms of atomics
        const int sc = sl * kPlanSub + sub;
        if (lane == leader) base = atomicAdd(&counters[sc], (unsigned)nsame);
        base = __shfl(base, leader, 64);
        if (slot == sl && fill) {
            const int rnk = __popcll(same & ((1ull << lane) - 1ull));
            cells[offsets[sc] + base + rnk] = make_int2(i, j);
        }
        todo &= ~same;
    }
    }
}

// ------------------------------------------------------------------------------------------
// results -> bits
// ------------------------------------------------------------------------------------------
__global__ void k_scatter_bits(int ncells, const int2* cells, const double* chi, double fast_th,
                               double slow_th, int rpr, const int* slot, int words, unsigned long long* upper)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncells) return;
    const int2 cc = cells[c];
    const double th = cc.x == cc.y ? fast_th : slow_th;
    const bool ok = !(chi[c] > th);                 // consensus_utils.cpp:18 (NaN agrees, as there)
    if (ok) atomicOr(&upper[(size_t)(slot[cc.x] % rpr) * words + (cc.y >> 6)], 1ull << (cc.y & 63));
}

// Symmetric N x N bit matrix from the gathered upper-triangle rows (row a at gathered row slot[a] = owner * rpr + its
// index among the owner's rows, ipc_row_assignment).  One wave per tile of 64 rows x one 64-bit word: lane l owns candidate j = 64 w + l
// (its interval and its diagonal bit), a row's overlap test is ONE compare per lane and a ballot, and the common
// case -- no candidate of the word overlaps row i (reference src/consensus.cpp:157-159: then C[i][j] = C[i][i] & C[j][j])
// -- is a single select of the word of diagonal bits; only overlapping pairs read their solved bit.
__global__ __launch_bounds__(256) void k_assemble(int N, int words, const int* slot, const int* lo, const int* hi,
                                                  const unsigned long long* gathered, unsigned long long* bits)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int w = blockIdx.x;
    auto U = [&](int a, int c) -> unsigned {        // a <= c
        const unsigned long long word = gathered[(size_t)slot[a] * words + (c >> 6)];
        return (unsigned)((word >> (c & 63)) & 1ull);
    };
    const int j = w * 64 + lane;
    const bool vj = j < N;
    const int loj = vj ? lo[j] : 0, hij = vj ? hi[j] : 0;
    const unsigned long long Dword = __ballot(vj && U(j, j));
    for (int rb = blockIdx.y * 4 + wave; rb * 64 < N; rb += gridDim.y * 4) {
        const int ir = rb * 64 + lane;                 // this lane's row of the tile
        const bool vr = ir < N;
        const int lor = vr ? lo[ir] : 0, hir = vr ? hi[ir] : 0;
        const unsigned long long dmask = __ballot(vr && U(ir, ir));
        const int nrows = min(64, N - rb * 64);
        unsigned long long mine = 0ull;
        for (int r = 0; r < nrows; ++r) {
            const int i = rb * 64 + r;
            const int loi = __builtin_amdgcn_readlane(lor, r), hii = __builtin_amdgcn_readlane(hir, r);
            unsigned long long word = ((dmask >> r) & 1ull) ? Dword : 0ull;
            const bool ov = vj && j != i && (min(hii, hij) - max(loi, loj) > 0);
            const unsigned long long ovmask = __ballot(ov);
            if (ovmask) {
                const bool bit = ov && U(min(i, j), max(i, j));
                word = (word & ~ovmask) | __ballot(bit);
            }
            if (lane == r) mine = word;
        }
        if (vr) bits[(size_t)ir * words + w] = mine;
    }
}

// Greedy set-max: candidates in processing order, 64 per round (four per wave -- 32 / two until round 5: a round is two
// trips to memory whatever it holds, and at N = 25 000 the 160 rounds were 1.3 ms that no rank of an 8-GPU run can shed;
// the rows of a wave's candidates are requested together,
// a round is a memory round trip).  Candidates whose own cell failed can never join, so the order list is first
// compacted to those with a set diagonal bit (order preserved).  Within a round the candidates are taken in order: one
// joins if its row covers the set as it stood before the round AND every member of the round taken before it.
__global__ __launch_bounds__(1024) void k_set_max(int N, int words, const int* order,
                                                  const unsigned long long* bits, unsigned char* accepted, int* live)
{
    constexpr int CPW = 4, RC = 16 * CPW;           // candidates per wave / per round
    extern __shared__ unsigned long long acc[];     // [words] accepted mask
    __shared__ int okflag[RC];
    __shared__ unsigned long long conf[RC];
    __shared__ int kk[RC];
    __shared__ int wcount[16];
    __shared__ int nlive_s;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int w = tid; w < words; w += blockDim.x) acc[w] = 0ull;
    for (int k = tid; k < N; k += blockDim.x) accepted[k] = 0;
    if (tid == 0) nlive_s = 0;
    __syncthreads();
    for (int base = 0; base < N; base += 1024) {     // ordered compaction, 1024 positions per pass
        const int pos = base + tid;
        const int k = pos < N ? order[pos] : -1;
        const bool keep = k >= 0 && ((bits[(size_t)k * words + (k >> 6)] >> (k & 63)) & 1ull);
        const unsigned long long m = __ballot(keep);
        if (lane == 0) wcount[wave] = __popcll(m);
        __syncthreads();
        int off = nlive_s;
        for (int v = 0; v < wave; ++v) off += wcount[v];
        if (keep) live[off + __popcll(m & ((1ull << lane) - 1ull))] = k;
        __syncthreads();
        if (tid == 0) { int t = 0; for (int v = 0; v < 16; ++v) t += wcount[v]; nlive_s += t; }
        __syncthreads();
    }
    __threadfence_block();
    const int nlive = nlive_s;
    for (int base = 0; base < nlive; base += RC) {
        int k[CPW];
        bool bad[CPW];
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const int pos = base + c * 16 + wave;    // (slot c * 16 + wave of the round: slots are in processing order)
            k[c] = pos < nlive ? live[pos] : -1;
            bad[c] = false;
        }
        for (int w = lane; w < words; w += 64) {
            const unsigned long long a = acc[w];
#pragma unroll
            for (int c = 0; c < CPW; ++c) {
                const unsigned long long r = k[c] >= 0 ? bits[(size_t)k[c] * words + w] : ~0ull;
                bad[c] = bad[c] || (r & a) != a;
            }
        }
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const bool ok = k[c] >= 0 && __ballot(bad[c]) == 0ull;
            if (lane == 0) { okflag[c * 16 + wave] = ok ? 1 : 0; kk[c * 16 + wave] = k[c]; }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            if (k[c] < 0) continue;
            // bits of this candidate against the other members of the round
            unsigned m = 0;
            if (lane < RC && kk[lane] >= 0) {
                const int o = kk[lane];
                m = (unsigned)((bits[(size_t)k[c] * words + (o >> 6)] >> (o & 63)) & 1ull);
            }
            const unsigned long long bal = __ballot(m != 0);
            if (lane == 0) conf[c * 16 + wave] = bal;
        }
        __syncthreads();
        if (tid == 0) {
            unsigned long long taken = 0;
            for (int w = 0; w < RC; ++w) {
                if (kk[w] < 0 || !okflag[w]) continue;
                if ((conf[w] & taken) != taken) continue;
                taken |= 1ull << w;
                acc[kk[w] >> 6] |= 1ull << (kk[w] & 63);
                accepted[kk[w]] = 1;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// engine
// ------------------------------------------------------------------------------------------
// Chain-length bins: each bin is served by one (W, M) kernel variant.  The policy string
// (env IPC_SE2_POLICY, default below) lists the variants to use as "WxM" tokens (block kernel, W
// waves per cell, M poses per lane), "wM" tokens (SE2 wave kernel, one wave per cell, M poses per
// lane), "pM" tokens (SE2 pair kernel, two waves per cell) or "qM" tokens (SE2 quad kernel, four
// waves per cell); a cell goes to the listed variant of
// smallest capacity 64*W*M that holds it.
struct BinPlan {
    BinCaps caps;
    int variant[kMaxBins];
    // SE3: variant for a bin with too few cells to fill the GPU (more waves per cell, so the few
    // cells finish sooner), and the cell count below which it is used
    int latency_variant[kMaxBins];
    int latency_below[kMaxBins];
};
static const char* kDefaultPolicy = "w1,w3,w5,w7,w9,w11,w13,p7,p9,p11,q7,q9,q11,q13,16x4,16x8,16x16";
// engine streams + the caller's stream = the runtime's four hardware queues for SE2; the long SE3
// launches gain a little from a fourth engine stream
static const int kDefaultSideStreams2 = 7, kDefaultSideStreams3 = 7;      // (round 5: 2 / 3 until the side streams came from the process pool -- at 8 ranks a shard's bins hold fewer cells than the GPU has CUs and only concurrent launches fill it: C2 emulated 6.53x -> 7.07x, one rank unchanged)
static const char* kDefaultPolicy3 = "w1,w2,w3,w4,w6,w8,g3,g4,g5,g6,g7,g8,g9,g10,16x4";
static const char* kBlockPolicy3 = "1x1,1x2,1x3,2x2,2x3,4x2,4x4,8x4,8x5,16x4";   // round-1 block kernels only (IPC_SE3_POLICY=block)
// SE3 variants for thin bins, by capacity (IPC_SE3_LATENCY_POLICY; "none" disables the switch)
static const char* kDefaultLatencyPolicy3 = "1x1,2x1,4x1,4x2,4x4,8x4,8x5,16x4";
static bool make_plan(BinPlan& bp, int dim, std::string& err)
{
    const char* env = getenv(dim == 2 ? "IPC_SE2_POLICY" : "IPC_SE3_POLICY");
    std::string pol = env && *env ? env : (dim == 2 ? kDefaultPolicy : kDefaultPolicy3);
    if (dim == 3 && pol == "block") pol = kBlockPolicy3;
    const Variant* table = dim == 2 ? kVariants : kVariants3;
    const int ntable = dim == 2 ? kNumVariants : kNumVariants3;
    std::vector<std::pair<int, int>> items;          // (cap, variant)
    size_t pos = 0;
    while (pos < pol.size()) {
        size_t e = pol.find(',', pos);
        if (e == std::string::npos) e = pol.size();
        int w = 0, m = 0;
        const std::string tok = pol.substr(pos, e - pos);
        int v = -1;
        if (dim == 2 && sscanf(tok.c_str(), "w%d", &m) == 1) {          // wave kernel, M poses per lane
            w = 1;
            for (int k = 0; k < kNumWaveM; ++k) if (kWaveM[k] == m) v = kWaveVariantBase + m;
        } else if (dim == 2 && sscanf(tok.c_str(), "p%d", &m) == 1) {   // pair kernel, two waves per cell
            w = 2;
            for (int k = 0; k < kNumPairM; ++k) if (kPairM[k] == m) v = kPairVariantBase + m;
        } else if (dim == 2 && sscanf(tok.c_str(), "q%d", &m) == 1) {   // quad kernel, four waves per cell
            w = 4;
            for (int k = 0; k < kNumQuadM; ++k) if (kQuadM[k] == m) v = kQuadVariantBase + m;
        } else if (dim == 3 && (sscanf(tok.c_str(), "w%d", &m) == 1 || sscanf(tok.c_str(), "g%d", &m) == 1)) {
            w = tok[0] == 'w' ? 1 : 4;                                     // LDS-pose kernel, teams of 1 / 4 waves
            for (int k = 0; k < kNumLdsVariants3; ++k)
                if (kLdsVariants3[k].W == w && kLdsVariants3[k].M == m) v = kLdsVariantBase3 + k;
        } else {
            if (sscanf(tok.c_str(), "%dx%d", &w, &m) != 2) { err = "bad IPC_SE2_POLICY token"; return false; }
            for (int k = 0; k < ntable; ++k) if (table[k].W == w && table[k].M == m) v = k;
        }
        if (v < 0) { err = "IPC_SE*_POLICY names a variant that is not compiled: " + tok; return false; }
        items.push_back({64 * w * m, v});
        pos = e + 1;
    }
    std::sort(items.begin(), items.end());
    if (items.empty() || (int)items.size() > kMaxBins) { err = "IPC_SE2_POLICY: 1..32 variants"; return false; }
    bp.caps.n = (int)items.size();
    for (int b = 0; b < kMaxBins; ++b) {
        bp.caps.cap[b] = b < bp.caps.n ? items[b].first : 0;
        bp.variant[b] = b < bp.caps.n ? items[b].second : -1;
        bp.latency_variant[b] = -1;
        bp.latency_below[b] = 0;
    }
    if (dim == 3) {
        const char* lenv = getenv("IPC_SE3_LATENCY_POLICY");
        const std::string lpol = lenv && *lenv ? lenv : kDefaultLatencyPolicy3;
        if (lpol != "none") {
            size_t lp = 0;
            std::vector<std::pair<int, int>> lat;       // (cap, variant)
            while (lp < lpol.size()) {
                size_t e = lpol.find(',', lp);
                if (e == std::string::npos) e = lpol.size();
                int w = 0, m = 0, v = -1;
                if (sscanf(lpol.substr(lp, e - lp).c_str(), "%dx%d", &w, &m) != 2) { err = "bad IPC_SE3_LATENCY_POLICY token"; return false; }
                for (int k = 0; k < ntable; ++k) if (table[k].W == w && table[k].M == m) v = k;
                if (v < 0) { err = "IPC_SE3_LATENCY_POLICY names a variant that is not compiled"; return false; }
                lat.push_back({64 * w * m, v});
                lp = e + 1;
            }
            std::sort(lat.begin(), lat.end());
            for (int b = 0; b < bp.caps.n; ++b) {
                if (bp.variant[b] >= kLdsVariantBase3) continue;             // block variants only
                const int wt = table[bp.variant[b]].W;
                for (const auto& lv : lat) {
                    if (lv.first < bp.caps.cap[b]) continue;
                    // only a variant with more waves per cell is a latency variant
                    if (table[lv.second].W > wt) {
                        bp.latency_variant[b] = lv.second;
                        bp.latency_below[b] = wt <= 4 ? 4 / wt : 1;      // x CUs at launch time
                    }
                    break;
                }
            }
        }
    }
    return true;
}

static hipError_t launch_se3_lds(int nl, int idx, int n, hipStream_t st, const Se3View& P, const int2* cells,
                                 SolveParams prm, CellOut out, unsigned* counter, int n_cu)
{
#define IPC_LCASE(i, WW, MM) case i: return launch_se3_lds_##WW##_##MM(nl, n, st, P, cells, prm, out, counter, n_cu);
    switch (idx) {
        IPC_LCASE(0, 1, 1) IPC_LCASE(1, 1, 2) IPC_LCASE(2, 1, 3) IPC_LCASE(3, 1, 4) IPC_LCASE(4, 1, 6) IPC_LCASE(5, 1, 8)
        IPC_LCASE(6, 4, 2) IPC_LCASE(7, 4, 3) IPC_LCASE(8, 4, 4) IPC_LCASE(9, 4, 5) IPC_LCASE(10, 4, 6) IPC_LCASE(11, 4, 7)
        IPC_LCASE(12, 4, 8) IPC_LCASE(13, 4, 9) IPC_LCASE(14, 4, 10)
        default: return hipErrorInvalidValue;
    }
#undef IPC_LCASE
}

struct ipc_engine {
    int dim = 2, V = 0, N = 0, device = 0;
    ipc_params_t prm{};
    double term_eps = 1e-13;                           // IPC_TERMINATE_EPS (0: g2o's literal trial loop), Se2View::term_eps
    BinPlan plan{};
    hipStream_t own_stream = nullptr;
    // chain
    double* d_chain = nullptr; int estride = 0;
    double* d_chain_rec = nullptr;                     // record-major copy [E + 64][F_NFIELDS | G_NFIELDS]
    double2* d_chain_blk = nullptr;                    // SE3: blocked copy [E / 64 + 2][kSe3BlkPairs][64]
    double* d_pose0 = nullptr;
    // candidates
    double* d_cand = nullptr; int cstride = 0;        // cstride = capacity in records (>= N): the list grows in place
    int *d_from = nullptr, *d_to = nullptr, *d_lo = nullptr, *d_hi = nullptr, *d_order = nullptr;
    int* d_live = nullptr;                             // set-max: candidates with a set diagonal bit, in processing order
    int* d_rowperm = nullptr;                          // rows grouped by owning rank (of slot_world), each group in the order the planning pass visits them (by first vertex, then index)
    std::vector<int> row_group_off;                    // [world + 1] offsets of the groups
    bool order_stale = false;                          // d_order is behind `order` (appends): re-sent by the next matrix-mode call
    std::vector<void*> retired;                        // candidate arrays a growth replaced while solves in flight may still read them
    hipEvent_t ev_cand = nullptr; bool cand_event = false;   // behind the last record written by ipc_append_candidate (own_stream)
    std::vector<int> order, h_lo, h_hi;
    std::vector<int> h_cand_ids; std::vector<double> h_cand_meas, h_cand_info;   // raw candidate records in file order
    // plan / results of the last solve
    unsigned* d_counters = nullptr;   // [2*(kMaxBins+1)]
    unsigned* d_offsets = nullptr;
    unsigned* d_wave_ctr = nullptr;   // work-queue heads of the wave-kernel launches, one per (nl, bin)
    int n_cu = 256;
    int2* d_cells = nullptr; size_t cells_cap = 0;
    double *d_chi = nullptr, *d_chitot = nullptr; int4* d_meta = nullptr;
    // The cell lists of (rank, world) change only with the candidates: the two planning passes and their read-back (the
    // first of ipc_solve_rows' two host waits) are paid once per candidate list, not once per step (round 5).
    bool plan_cached = false; int plan_rank = -1, plan_world = 0; size_t plan_total = 0;
    int slow_first_iterations = 96;                    // IPC_SLOW_FIRST (0: off): cells that took this many iterations go to the head of their slot
    std::vector<unsigned> plan_counts, plan_offsets;
    // Borderline cells (borderline_band) are solved again with g2o's literal trial loop BY THE CELL KERNELS (term_eps 0)
    // over compact per-slot lists built on the device -- no per-cell copies, no host-driven solves (round 5).
    int2* d_lit_cells = nullptr; int* d_lit_idx = nullptr; double *d_lit_chi = nullptr, *d_lit_chitot = nullptr; int4* d_lit_meta = nullptr;
    unsigned* d_slot_off = nullptr;                    // [slots + 1] first cell of each (loop count, bin) slot
    int* d_recount = nullptr; int* h_recount = nullptr;   // [slots] borderline cells per slot, then the failed-cell count; pinned copy
    int last_cells = 0, last_long_cells = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr; bool ev_valid = false; int last_launches = 0;
    // side streams: the bin launches of one solve are spread over them so that the tail of one
    // launch (a few cells that run to the iteration cap) overlaps the next bins
    static constexpr int kMaxSide = 7;
    int n_side = 0;
    bool side_forced = false;                          // IPC_SIDE_STREAMS given: used as is, whatever the size of the step
    hipStream_t side[kMaxSide] = {};
    hipEvent_t ev_fork = nullptr, ev_join[kMaxSide] = {}, ev_join_own = nullptr;
    // scratch for ipc_run
    unsigned long long *d_upper = nullptr, *d_bits = nullptr; unsigned char* d_acc = nullptr; size_t run_cap = 0;
    // incremental mode / final map (SE2)
    std::vector<double> h_odom_meas, h_odom_info;      // file values, for the un-scaled chain
    std::vector<int> h_from, h_to, cns;
    double* d_chain1 = nullptr;                        // chain records with (info * s) / s
    double* d_open = nullptr;                          // [5][V] open-loop x y th cos sin
    double* d_cur = nullptr;                           // [5][V] current estimates
    ClusterSolver2* cluster = nullptr;
    ClusterSolver3* cluster3 = nullptr;                // SE3: d_open is d_pose0 itself, d_cur is [12][V]
    // device-resident dog-leg (cluster_persist.hpp): the default; IPC_CLUSTER_MODE=host keeps the host-driven kernels
    bool persist = true;
    bool last_persist = false;                         // which solver holds the poses of the last cluster solve
    int* d_slot = nullptr; int slot_world = 0; int row_policy = 1;    // row -> shard slot of the last world size (IPC_ROW_BALANCE=cyclic|cost)
    int* d_failed = nullptr; int failed_cap = 0; int last_lm_cells = 0;    // cells of the last solve redone with Levenberg damping
    bool lm_retry = true;                              // IPC_LM_RETRY=0: a failed linear solve ends the optimisation (flags & 2), no damping
    // Cells whose max chi2 ends within this relative distance of their threshold are solved again with g2o's literal
    // trial loop (term_eps 0): the convergence test can move an edge's chi2 by up to 2 sqrt(term_eps) relative,
    // so outside the band the decision of the literal loop is the one already taken.  IPC_BORDERLINE_BAND (0: off).
    double borderline_band = -1.0;                     // < 0: 4 sqrt(term_eps)
    int last_literal_cells = 0;
    long lm_fallbacks = 0;
    long persist_timeouts = 0;                         // persistent launches whose grid barrier gave up (lost launches)
    long persist_relaunches = 0;                       // ... that were launched again (cluster_solve: twice before the host-driven solver)
    bool cns_dups = false;                             // some edge sits in the consensus set more than once (an accepted re-check, src/consensus.cpp:70)
    PersistSolver<PersistSe2>* persist2 = nullptr;
    PersistSolver<PersistSe3>* persist3 = nullptr;
    unsigned long long* d_prof = nullptr;              // IPC_PERSIST_PROF=1: phase clocks of the persistent kernel's leader, printed by ipc_destroy
    int max_helpers = -1;                              // IPC_PERSIST_HELPERS
    PersistKnobs knobs;                                            // IPC_BAND_* / IPC_PERSIST_FAULT_EVERY as they stood at ipc_create: every solver instance of the engine gets them
    int literal_band_min_n = 3072;                     // IPC_LITERAL_BAND_MIN_N, likewise
    // Speculative candidate pipeline of the faithful mode (IPC_SPEC_WINDOW solves in flight, default 10; 1 = off).
    // A rejected candidate leaves the state untouched (reference src/consensus.cpp:63-67), so the checks of the candidates
    // that FOLLOW in the processing order can start from the same state before its verdict is known; and the state an
    // accept leaves behind is known as soon as ITS solve ends, so the candidates after it start from that state while
    // slower, earlier solves are still running (assumed to reject -- 87 % do on C2).  Solves run in persistent launches
    // on their own streams; finished results are parked on the host until their candidate is asked for.  A result is
    // used only if the state it started from is, at that moment, the committed one: every decision is the one the
    // one-at-a-time loop takes.  An accept nobody assumed tells the later solves to stop (host-mapped word) and they are
    // redone from the new state.
    struct SpecSlot {
        PersistSolver<PersistSe2>* s2 = nullptr;
        PersistSolver<PersistSe3>* s3 = nullptr;
        hipStream_t st = nullptr;
        hipEvent_t done = nullptr;                     // behind the result copy of the solve in flight
        int cand = -1, pos = -1;                       // candidate / processing position of the solve in flight, -1: idle
        int state = -1;                                // index of the pose state it started from (spec_states)
        int launch_id = 0;
        int busy_wgs = 0;                              // workgroups of the last launch until its `done` completes (also after an abort)
        int expect = -1;                               // the verdict the scheduler expected when it launched the solve (1 accept, 0 reject, -1 none): IPC_SPEC_STATS
        std::chrono::steady_clock::time_point t_launch;   // (IPC_SPEC_STATS: launch-to-collect time per dog-leg iteration)
        int lo = 0, hi = 0, nclu = 0;
        double th = 0.0;
        double pred_ratio = -1.0;                      // the candidate's own chi2 at the state it starts from / the slow threshold (IPC_SPEC_LOG)
    };
    struct SpecState {                                 // a pose state solves start from
        double* d_poses = nullptr;                     // d_cur (not owned) or a buffer of its own, [5 | 12][V]
        bool owned = false;
        hipEvent_t ready = nullptr;                    // its poses are complete (recorded on the stream that wrote them)
        bool has_ready = false;
        std::vector<int> cns;                          // the consensus set it stands for
        int pos = -1;                                  // the accept at this processing position on top of its parent
        int users = 0;                                 // solves in flight that read it
        bool live = false;                             // committed, or in the tentative chain
        // predictions: every candidate's own chi2 at these poses (k_cand_own_chi2), copied to pinned host memory
        double* d_pred = nullptr; double* h_pred = nullptr; int pred_cap = 0, pred_n = 0;
        hipEvent_t pred_ev = nullptr; bool has_pred = false, pred_ready = false;
    };
    struct SpecResult {                                // a finished solve waiting for its candidate's turn
        bool valid = false, agree = false, retry_host = false;
        int state = -1, child = -1;                    // started from / the tentative state its accept made
        int lo = 0, hi = 0, nclu = 0;
        ClusterOut o;
    };
    std::vector<SpecSlot> slots;
    std::vector<SpecState> spec_states;
    std::vector<int> tent;                             // tentative states, by position
    std::vector<SpecResult> spec_res;                  // by processing position
    int committed_state = -1;
    // The pipeline's PREDICTION of the caller's order: positions -> candidates.  Starts as `order` (cmpTime, ties by index);
    // a caller that asks for another candidate than the one at the head has that candidate moved to the head (the solves
    // behind it assumed rejects in front of them, which is still what they assume) -- the prediction costs time when it
    // is wrong, never a decision.  Positions < spec_head are the candidates already handed out.
    std::vector<int> porder, ppos;
    std::vector<char> handed;                          // by candidate: its verdict has been handed to the caller since the last reset
    int spec_head = -1;                                // next position to hand out; -1: pipeline empty
    int spec_window = 4, spec_ahead = 256;             // solves in flight / positions ahead of the head (IPC_SPEC_AHEAD)
    bool window_forced = false;                        // IPC_SPEC_WINDOW given: taken as is, no probe
    int stream_concurrency = 0;                        // streams of the window measured to run side by side
    int spec_active = 0;                               // slots the pipeline uses (fewer than the window when the streams share hardware queues)
    double accept_rate = 0.5;                          // running mean over the recent verdicts: how far ahead it pays to assume "reject"
    // Predicted verdicts (scheduling only, never a decision): a solve behind a candidate that is expected to be accepted
    // and whose verdict is still out will most likely be thrown away, so at most spec_behind of them are started; in
    // front of it everything in flight is useful and the window may be as wide as the CUs allow.
    std::vector<char> pred_accept;                     // by candidate: 1 = expected to be accepted (IPC_SPEC_PREDICT_FILE: a recorded run, experiments)
    double pred_k = 30.0;                              // predicted accept: own chi2 at the state it starts from <= pred_k x the slow threshold (IPC_SPEC_PREDICT; 0: off; 10 until round 5: on C5 a third of the mispredicted accepts -- each drops the ~15 solves in flight behind it -- lie between 10 and 30, C5 prefix 23.0 -> 21.5 s, C2 and C4 unchanged)
    std::vector<double> pred_last;                     // the newest predictions that have arrived (stand-in while a state's own are on their way)
    int spec_behind = 1;                               // IPC_SPEC_BEHIND (C1: 0.56 / 0.59 / 0.59 / 0.63 s with 0 / 1 / 2 / 4 -- every launch is host time on the accept chain; C2, C4m: within noise)
    // IPC_SPEC_HEDGE: while an expected accept is out, the NEXT expected accepts in line (this many) are started as well, on the state
    // in front of it -- the chain's continuation if it rejects after all (C4: one expected accept in ten does, and takes ten times as
    // long as an accept to say so; the chain stood still for the 2.5 accept times of the gate: the tail of the run's accept-to-
    // next-solve gaps, median 0.3 ms, 90th percentile 18 ms).  -1: on when an accepted solve takes >= 3 ms (C1 / C2: a launch is
    // host time on the chain and the solves are 2 ms).  MEASURED: no gain -- C4 prefix 26.1 / 26.6 / 27.6 s and C5 13.5 / 13.7 / 13.9 s with 0 / 1 / 2,
    // one barrier time-out with 3: what the hedges win on the runs of rejecting expected accepts the crowding takes from every
    // other solve (an accept solve takes 10 ms on the device with ten solves in flight, 15 ms with sixteen) -- default 0
    double uncertain_from = 0.0;                       // IPC_SPEC_UNCERTAIN: an expected accept whose own chi2 is above this x the slow threshold is a coin toss (C4: 107 accepted,
                                                       // 107 rejected between 3 and 30) -- the next expected accept is started beside it instead of behind it.  0: off
    int xcd_cus = 0;                                   // IPC_SPEC_XCD_CUS: workgroups of solves per XCD (0: its CUs less one)
    int spec_hedge = 0;
    int helper_limit_reject = 8;                       // helper workgroups of a solve that is expected to reject (IPC_PERSIST_HELPERS_REJECT):
                                                       // the rejects are the bulk of the work and independent of each other -- many of them
                                                       // side by side; the expected accepts are the serial chain -- each as fast as it can be
    double gate_release_ms = 0.0;                      // a predicted accept still running after this long counts as a reject (0: 2.5 x the mean accepted solve)
    int helper_limit = 39;
    double st_acc_s = 0, st_rej_s = 0; long st_acc_it = 0, st_rej_it = 0, st_acc_n = 0, st_rej_n = 0;   // IPC_SPEC_STATS
    double st_acc_dev_s = 0, st_rej_dev_s = 0;         // the same solves by the leader's own clock
    unsigned long long commit_count = 0;
    hipEvent_t ev_commit = nullptr;
    int* h_abort = nullptr;                            // host-mapped: one word per slot, the launch id to give up
    int* d_abort = nullptr;
    int next_launch_id = 1;
    long spec_hits = 0, spec_launches = 0, spec_wasted = 0, spec_tentative = 0, spec_promoted = 0;
    double spec_t_launch = 0, spec_t_tent = 0, spec_t_total = 0;   // host seconds (IPC_SPEC_STATS)
    // IPC_SPEC_STATS: why slots stood empty, in slot x pump calls (the caller's thread spins on the pump, so this is time):
    // [0] behind an expected accept, [1] look-ahead used up / no candidate left, [2] CU budget, [3] cluster beyond the persistent solver, [4] total slot-pumps
    unsigned long long idle_why[5] = {0, 0, 0, 0, 0};
    hipStream_t pred_stream = nullptr;                 // the predictions of new tentative states (process pool, not owned)
    FILE* spec_log = nullptr;                          // IPC_SPEC_LOG=<file>: one line per finished solve / tentative state / verdict handed out (tools/spec_chain.py)
    std::chrono::steady_clock::time_point spec_log_t0;
    long pred_conf[3][2] = {{0, 0}, {0, 0}, {0, 0}};   // [expectation + 1][verdict] over the solves whose result was kept (IPC_SPEC_STATS)
};

static int spec_quiesce(ipc_engine* h, bool state_changes);
static int spec_reset(ipc_engine* h);
// The pipelines of one DEVICE share its CUs (round 6; until round 5 one pipeline per PROCESS: a check on engine B reset the
// look-ahead of engine A).  Every workgroup of every persistent solve on a GPU must be resident, so the workgroup budget of
// spec_pump counts the solves of EVERY engine on the device (foreign_busy below), each engine still runs its own window on
// the device's one stream pool, and nobody resets anybody: two replicas on one GPU, asked alternately, each keep their
// look-ahead.  Engines on DIFFERENT devices have nothing to share at all -- the faithful mode is "replicas only" across
// GPUs: one process with an engine per device (the C++ testers, IPC_AMD_DEVICES).
struct DevicePipeline {
    ipc_engine* active = nullptr;                      // the engine whose check ran last (matrix mode gives every pipeline of the device up)
    std::vector<ipc_engine*> engines;                  // engines with a pipeline on this device (spec_ensure ... ipc_destroy), under run_mu
    // Held for the WHOLE of every call that reads or edits the pipeline state of an engine on this device (slots, parked
    // results, tentative states, head): a check on engine A stops the pipeline of engine B ON THE SAME DEVICE (spec_reset of
    // a FOREIGN engine), which B's owner thread may be pumping at that moment -- round 4 only guarded the pointer.  Recursive:
    // a check that falls back to the host-driven solver quiesces its own pipeline from inside.  Two threads that each run a
    // faithful loop on ONE device therefore take turns check by check (they would undo each other's look-ahead anyway).
    std::recursive_mutex run_mu;
};
static std::mutex g_pipeline_mu;                       // guards the maps below (engines of different host threads)
static std::map<int, DevicePipeline*> g_device_pipeline;      // never freed: a mutex other threads may hold must not die
static DevicePipeline& device_pipeline(int device)
{
    std::lock_guard<std::mutex> lk(g_pipeline_mu);
    DevicePipeline*& p = g_device_pipeline[device];
    if (!p) p = new DevicePipeline();
    return *p;
}
// The pipeline's streams belong to the PROCESS, per device, not to an engine: only one pipeline runs at a time, and beyond
// about two dozen streams the runtime runs them one after the other -- a second live engine with sixteen streams of its own
// was enough to take C1 from 0.56 to 1.2 s (round 4: the full test suite in one process).  Created on demand, never destroyed.
static std::map<int, std::vector<hipStream_t>> g_slot_streams;
static hipError_t pipeline_stream(int device, int q, hipStream_t* out)
{
    std::lock_guard<std::mutex> lk(g_pipeline_mu);
    std::vector<hipStream_t>& pool = g_slot_streams[device];
    while ((int)pool.size() <= q) {
        hipStream_t st = nullptr;
        const hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (e != hipSuccess) return e;
        pool.push_back(st);
    }
    *out = pool[q];
    return hipSuccess;
}

// Device-to-device copy the host can rely on when it returns.  (hipMemcpy of this kind goes to the NULL stream and is not
// complete -- not even submitted, as it turned out -- when the call returns; the engine's streams are non-blocking, so a
// solve enqueued right behind it could read the buffer before the copy: round 4, a faithful run that started from the
// previous run's poses once in a dozen repetitions.)
static hipError_t copy_d2d_now(ipc_engine* h, void* dst, const void* src, size_t bytes)
{
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, h->own_stream);
    return e != hipSuccess ? e : hipStreamSynchronize(h->own_stream);
}
extern "C" int ipc_rows_per_rank(int n, int world) { return world > 0 ? (n + world - 1) / world : 0; }

// Which rank solves which row, and where the row sits in that rank's shard: slot[i] = owner * rpr + index, rpr =
// ipc_rows_per_rank(n, world).  policy 0: row-cyclic (i % world).  policy 1 (default): balanced by cost -- a row's
// cost is the number of poses its cells sweep, sum over the overlapping pairs (i, j > i) of the union chain length
// plus its own chain ("cost proportional to sum L, not to the row count"); rows go, costliest first, to
// the least loaded rank that still has a free slot (ties: lower rank, then lower index).  Pure host code, no GPU: the
// engine and every rank of a distributed run compute the same map from the same candidate list.
extern "C" int ipc_row_assignment(int n, const int* ids, int world, int policy, int* slot_out)
{
    if (n < 0 || world < 1 || (n > 0 && (!ids || !slot_out))) return fail(IPC_ERR_ARG, "ipc_row_assignment: bad argument");
    const int rpr = ipc_rows_per_rank(n, world);
    if (policy == 0 || world == 1) {
        for (int i = 0; i < n; ++i) slot_out[i] = (i % world) * rpr + i / world;
        return IPC_OK;
    }
    std::vector<int> lo(n), hi(n);
    for (int i = 0; i < n; ++i) { lo[i] = std::min(ids[2 * i], ids[2 * i + 1]); hi[i] = std::max(ids[2 * i], ids[2 * i + 1]); }
    // cost[i] = own chain + sum over the overlapping pairs (i, j > i) of the union chain length.  By a sweep over the
    // intervals sorted by first vertex: O(N log N + overlapping pairs) instead of N^2 / 2 compares (round 6: at N = 25 000
    // -- C5 -- the all-pairs loop was ~170 ms of host time in front of the FIRST step of every (rank, world), eight times
    // the 23 ms a rank of an 8-GPU run then solves for; the sums are the same integers, so is the assignment)
    std::vector<long long> cost(n);
    for (int i = 0; i < n; ++i) cost[i] = hi[i] - lo[i];
    {
        std::vector<int> ord(n);
        std::iota(ord.begin(), ord.end(), 0);
        std::sort(ord.begin(), ord.end(), [&](int a, int b) { return lo[a] != lo[b] ? lo[a] < lo[b] : a < b; });
        for (int p = 0; p < n; ++p) {
            const int a = ord[p], loa = lo[a], hia = hi[a];
            for (int q = p + 1; q < n && lo[ord[q]] < hia; ++q) {      // (lo_b >= lo_a: the overlap is min(hi) - lo_b)
                const int b = ord[q];
                if (std::min(hia, hi[b]) - lo[b] > 0) cost[std::min(a, b)] += std::max(hia, hi[b]) - loa;
            }
        }
    }
    std::vector<int> rows(n);
    std::iota(rows.begin(), rows.end(), 0);
    std::stable_sort(rows.begin(), rows.end(), [&](int a, int b) { return cost[a] > cost[b]; });
    std::vector<long long> load(world, 0);
    std::vector<int> used(world, 0);
    for (int i : rows) {
        int best = -1;
        for (int r = 0; r < world; ++r)
            if (used[r] < rpr && (best < 0 || load[r] < load[best])) best = r;
        slot_out[i] = best * rpr + used[best];
        ++used[best];
        load[best] += cost[i];
    }
    return IPC_OK;
}

extern "C" int ipc_create(int dim, int n_vertices, const double* odom_meas, const double* odom_info,
                          const ipc_params_t* params, int device, ipc_engine_t** out)
{
    if (!out) return fail(IPC_ERR_ARG, "ipc_create: out is NULL");
    *out = nullptr;
    if (dim != 2 && dim != 3) return fail(IPC_ERR_ARG, "ipc_create: dim must be 2 (SE2) or 3 (SE3), got %d", dim);
    if (n_vertices < 2 || !odom_meas || !odom_info || !params)
        return fail(IPC_ERR_ARG, "ipc_create: need >= 2 vertices and non-NULL arrays");
    if (!(params->s_factor > 0)) return fail(IPC_ERR_ARG, "ipc_create: s_factor must be > 0");
    // (The speculative window of the faithful mode runs one persistent launch per stream; streams that share a hardware
    // queue run their kernels one after the other, and the runtime's default is 4 queues.  The HOST PROGRAM exports
    // GPU_MAX_HW_QUEUES=24 before its first HIP call -- the testers' main() and bench.py do; a library does not edit its
    // process's environment (round 5).  Without it the window below is 4 and the stream probe of spec_ensure applies.)
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(IPC_ERR_ARG, "ipc_create: device %d of %d", device, ndev);
    HIPCHK(hipSetDevice(device));
    ipc_engine* h = new ipc_engine();
    h->dim = dim; h->V = n_vertices; h->prm = *params; h->device = device;
    {
        std::string perr;
        if (!make_plan(h->plan, dim, perr)) { delete h; return fail(IPC_ERR_ARG, "%s", perr.c_str()); }
    }
    if (const char* te = getenv("IPC_TERMINATE_EPS")) {
        if (*te) {
            char* end = nullptr;
            h->term_eps = strtod(te, &end);                          // "1e-13x", "off": refused, not read as 0
            while (end && (*end == ' ' || *end == '\t')) ++end;
            if (end == te || (end && *end)) { delete h; return fail(IPC_ERR_ARG, "IPC_TERMINATE_EPS: '%s' is not a number", te); }
        }
        if (!(h->term_eps >= 0) || h->term_eps > 1e-6) { delete h; return fail(IPC_ERR_ARG, "IPC_TERMINATE_EPS must be in [0, 1e-6]"); }
    }
    if (const char* pp = getenv("IPC_PERSIST_PROF")) {
        if (*pp && strcmp(pp, "0")) {
            if (hipMalloc(&h->d_prof, sizeof(unsigned long long) * kProfN) != hipSuccess) h->d_prof = nullptr;
            else hipMemset(h->d_prof, 0, sizeof(unsigned long long) * kProfN);
            hipStreamSynchronize(nullptr);                             // (NULL-stream copies are not ordered against the engine's non-blocking streams: copy_d2d_now)
        }
    }
    if (const char* mh = getenv("IPC_PERSIST_HELPERS")) { if (*mh) h->max_helpers = atoi(mh); }
    if (const char* lm = getenv("IPC_LM_RETRY")) { if (*lm) h->lm_retry = atoi(lm) != 0; }
    h->knobs = PersistKnobs::from_env();
    if (const char* e = getenv("IPC_LITERAL_BAND_MIN_N")) { if (*e) h->literal_band_min_n = atoi(e); }
    if (const char* bb = getenv("IPC_BORDERLINE_BAND")) {
        if (*bb) {
            char* end = nullptr;
            h->borderline_band = strtod(bb, &end);
            if (end == bb || *end || !(h->borderline_band >= 0) || h->borderline_band > 0.5) {
                delete h;
                return fail(IPC_ERR_ARG, "IPC_BORDERLINE_BAND must be a number in [0, 0.5]");
            }
        }
    }
    if (const char* sf = getenv("IPC_SLOW_FIRST")) { if (*sf) h->slow_first_iterations = std::max(0, atoi(sf)); }
    if (const char* rb = getenv("IPC_ROW_BALANCE")) {
        if (!strcmp(rb, "cyclic")) h->row_policy = 0;
        else if (*rb && strcmp(rb, "cost")) { delete h; return fail(IPC_ERR_ARG, "IPC_ROW_BALANCE must be 'cost' or 'cyclic'"); }
    }
    {   // window: as many solves in flight as there are hardware queues to run them side by side (IPC_SPEC_WINDOW overrides)
        const char* q = getenv("GPU_MAX_HW_QUEUES");
        const int nq = q ? atoi(q) : 4;
        h->spec_window = nq >= 24 ? 16 : (nq >= 13 ? 10 : (nq >= 9 ? 8 : 4));          // (+ the engine's own stream and its two side streams)
    }
    if (const char* sw = getenv("IPC_SPEC_WINDOW")) { if (*sw) { h->spec_window = std::max(1, std::min(32, atoi(sw))); h->window_forced = true; } }
    if (const char* sa = getenv("IPC_SPEC_AHEAD")) { if (*sa) h->spec_ahead = std::max(1, std::min(1024, atoi(sa))); }
    if (const char* pk = getenv("IPC_SPEC_PREDICT")) { if (*pk) h->pred_k = std::max(0.0, atof(pk)); }
    if (const char* hr = getenv("IPC_PERSIST_HELPERS_REJECT")) { if (*hr) h->helper_limit_reject = std::max(0, atoi(hr)); }
    if (const char* sb = getenv("IPC_SPEC_BEHIND")) { if (*sb) h->spec_behind = std::max(0, std::min(64, atoi(sb))); }
    if (const char* lg = getenv("IPC_SPEC_LOG")) { if (*lg) { h->spec_log = fopen(lg, "w"); h->spec_log_t0 = std::chrono::steady_clock::now(); } }
    if (const char* un = getenv("IPC_SPEC_UNCERTAIN")) { if (*un) h->uncertain_from = std::max(0.0, atof(un)); }
    if (const char* xc = getenv("IPC_SPEC_XCD_CUS")) { if (*xc) h->xcd_cus = std::max(0, atoi(xc)); }
    if (const char* hg = getenv("IPC_SPEC_HEDGE")) { if (*hg) h->spec_hedge = std::max(-1, std::min(8, atoi(hg))); }
    if (const char* gr = getenv("IPC_SPEC_GATE_MS")) { if (*gr) h->gate_release_ms = std::max(0.0, atof(gr)); }
    if (const char* cm = getenv("IPC_CLUSTER_MODE")) {
        if (!strcmp(cm, "host")) h->persist = false;
        else if (*cm && strcmp(cm, "persist")) { delete h; return fail(IPC_ERR_ARG, "IPC_CLUSTER_MODE must be 'persist' or 'host'"); }
    }
    const int E = n_vertices - 1;
    const int ms = dim == 2 ? 3 : 7, is = dim == 2 ? 6 : 21, nf = dim == 2 ? (int)F_NFIELDS : (int)G_NFIELDS;
    const int ps = dim == 2 ? 3 : 12;
    h->estride = (E + 63) & ~63;
    HIPCHK(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreate(&h->ev0));
    HIPCHK(hipEventCreate(&h->ev1));
    {
        const char* env = getenv("IPC_SIDE_STREAMS");
        h->n_side = env && *env ? atoi(env) : (h->dim == 2 ? kDefaultSideStreams2 : kDefaultSideStreams3);
        h->side_forced = env && *env;
        if (h->n_side < 0) h->n_side = 0;
        if (h->n_side > ipc_engine::kMaxSide) h->n_side = ipc_engine::kMaxSide;
        HIPCHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&h->ev_join_own, hipEventDisableTiming));
        // the side streams come from the per-device pool of the PROCESS, the one the pipeline's slots use (the pipeline is
        // quiesced before any matrix-mode launch, matrix_mode_enter): an engine owns one stream, so how many engines are
        // alive no longer decides whether the runtime's couple of dozen usable streams are exceeded (round 5)
        for (int k = 0; k < h->n_side; ++k) {
            HIPCHK(pipeline_stream(device, k, &h->side[k]));
            HIPCHK(hipEventCreateWithFlags(&h->ev_join[k], hipEventDisableTiming));
        }
    }
    HIPCHK(hipMalloc(&h->d_chain, sizeof(double) * (nf * (size_t)h->estride + 64)));   // + read-ahead padding
    HIPCHK(hipMalloc(&h->d_pose0, sizeof(double) * ps * (size_t)n_vertices));
    HIPCHK(hipMalloc(&h->d_counters, sizeof(unsigned) * 2 * (kMaxBins + 1) * kPlanSub));
    HIPCHK(hipMalloc(&h->d_offsets, sizeof(unsigned) * 2 * (kMaxBins + 1) * kPlanSub));
    HIPCHK(hipMalloc(&h->d_wave_ctr, sizeof(unsigned) * 2 * (kMaxBins + 1)));
    {
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, device));
        h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    double *d_m = nullptr, *d_i = nullptr;
    HIPCHK(hipMalloc(&d_m, sizeof(double) * ms * E));
    HIPCHK(hipMalloc(&d_i, sizeof(double) * is * E));
    HIPCHK(hipMemcpy(d_m, odom_meas, sizeof(double) * ms * E, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_i, odom_info, sizeof(double) * is * E, hipMemcpyHostToDevice));
    HIPCHK(hipStreamSynchronize(nullptr));                     // (NULL-stream copies are not ordered against the engine's non-blocking streams: copy_d2d_now)
    HIPCHK(hipMemsetAsync(h->d_chain, 0, sizeof(double) * (nf * (size_t)h->estride + 64), h->own_stream));
    if (dim == 2) {
        hipLaunchKernelGGL(k_se2_prep, dim3((E + 255) / 256), dim3(256), 0, h->own_stream, E, d_m, d_i,
                           params->s_factor, h->d_chain, h->estride);
        hipLaunchKernelGGL(k_se2_propagate, dim3(1), dim3(64), 0, h->own_stream, n_vertices, h->d_chain, h->estride,
                           h->d_pose0);
        HIPCHK(hipMalloc(&h->d_chain_rec, sizeof(double) * (size_t)F_NFIELDS * (E + 64)));
        HIPCHK(hipMemsetAsync(h->d_chain_rec, 0, sizeof(double) * (size_t)F_NFIELDS * (E + 64), h->own_stream));
        hipLaunchKernelGGL(k_records, dim3((E + 255) / 256), dim3(256), 0, h->own_stream, E, (int)F_NFIELDS, h->d_chain,
                           h->estride, h->d_chain_rec);
    } else {
        hipLaunchKernelGGL(k_se3_prep, dim3((E + 63) / 64), dim3(64), 0, h->own_stream, E, d_m, d_i,
                           params->s_factor, h->d_chain, h->estride);
        hipLaunchKernelGGL(k_se3_propagate, dim3(1), dim3(64), 0, h->own_stream, n_vertices, h->d_chain, h->estride,
                           h->d_pose0);
        HIPCHK(hipMalloc(&h->d_chain_rec, sizeof(double) * (size_t)G_NFIELDS * (E + 64)));
        HIPCHK(hipMemsetAsync(h->d_chain_rec, 0, sizeof(double) * (size_t)G_NFIELDS * (E + 64), h->own_stream));
        hipLaunchKernelGGL(k_records, dim3((E + 255) / 256), dim3(256), 0, h->own_stream, E, (int)G_NFIELDS, h->d_chain,
                           h->estride, h->d_chain_rec);
        const size_t nblk = (size_t)(E + 63) / 64 + 2;
        HIPCHK(hipMalloc(&h->d_chain_blk, sizeof(double2) * nblk * kSe3BlkPairs * 64));
        HIPCHK(hipMemsetAsync(h->d_chain_blk, 0, sizeof(double2) * nblk * kSe3BlkPairs * 64, h->own_stream));
        hipLaunchKernelGGL(k_se3_blocks, dim3((E + 255) / 256), dim3(256), 0, h->own_stream, E, h->d_chain, h->estride,
                           h->d_chain_blk);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->own_stream));
    HIPCHK(hipFree(d_m));
    HIPCHK(hipFree(d_i));
    h->h_odom_meas.assign(odom_meas, odom_meas + (size_t)ms * E);
    h->h_odom_info.assign(odom_info, odom_info + (size_t)is * E);
    *out = h;
    return IPC_OK;
}

static void free_candidates(ipc_engine* h)
{
    hipFree(h->d_cand); hipFree(h->d_from); hipFree(h->d_to); hipFree(h->d_lo); hipFree(h->d_hi); hipFree(h->d_order); hipFree(h->d_live);
    hipFree(h->d_rowperm); h->d_rowperm = nullptr;
    hipFree(h->d_slot); h->d_slot = nullptr; h->slot_world = 0;
    for (void* q : h->retired) hipFree(q);
    h->retired.clear();
    h->d_cand = nullptr; h->d_from = h->d_to = h->d_lo = h->d_hi = h->d_order = h->d_live = nullptr;
    h->N = 0; h->cstride = 0; h->order_stale = false; h->cand_event = false;
}

// candidate arrays for `cap` records (cap a multiple of 64); the record array zeroed on own_stream
static int alloc_candidates(ipc_engine* h, int cap)
{
    const int nf = h->dim == 2 ? (int)F_NFIELDS : (int)G_NFIELDS;
    HIPCHK(hipMalloc(&h->d_cand, sizeof(double) * nf * (size_t)cap));
    HIPCHK(hipMalloc(&h->d_from, sizeof(int) * cap));
    HIPCHK(hipMalloc(&h->d_to, sizeof(int) * cap));
    HIPCHK(hipMalloc(&h->d_lo, sizeof(int) * cap));
    HIPCHK(hipMalloc(&h->d_hi, sizeof(int) * cap));
    HIPCHK(hipMalloc(&h->d_order, sizeof(int) * cap));
    HIPCHK(hipMalloc(&h->d_live, sizeof(int) * cap));
    HIPCHK(hipMalloc(&h->d_rowperm, sizeof(int) * cap));
    HIPCHK(hipMemsetAsync(h->d_cand, 0, sizeof(double) * nf * (size_t)cap, h->own_stream));
    h->cstride = cap;
    return IPC_OK;
}

// The list outgrew its arrays: twice the capacity, the records copied over on own_stream.  The old arrays are only
// RETIRED (freed with the next full upload or the engine): solves in flight still read them, and hipFree would wait for
// every one of them.  log2(N) growths over a run, 2x the final size held at most.
static int grow_candidates(ipc_engine* h, int need)
{
    const int nf = h->dim == 2 ? (int)F_NFIELDS : (int)G_NFIELDS;
    int cap = std::max(64, h->cstride);
    while (cap < need) cap *= 2;
    double* o_cand = h->d_cand; const int o_stride = h->cstride;
    int *o_from = h->d_from, *o_to = h->d_to, *o_lo = h->d_lo, *o_hi = h->d_hi, *o_order = h->d_order, *o_live = h->d_live, *o_perm = h->d_rowperm;
    if (int rc = alloc_candidates(h, cap)) return rc;
    if (h->N > 0) {
        HIPCHK(hipMemcpy2DAsync(h->d_cand, sizeof(double) * cap, o_cand, sizeof(double) * o_stride, sizeof(double) * h->N, nf,
                                hipMemcpyDeviceToDevice, h->own_stream));
        HIPCHK(hipMemcpyAsync(h->d_from, o_from, sizeof(int) * h->N, hipMemcpyDeviceToDevice, h->own_stream));
        HIPCHK(hipMemcpyAsync(h->d_to, o_to, sizeof(int) * h->N, hipMemcpyDeviceToDevice, h->own_stream));
        HIPCHK(hipMemcpyAsync(h->d_lo, o_lo, sizeof(int) * h->N, hipMemcpyDeviceToDevice, h->own_stream));
        HIPCHK(hipMemcpyAsync(h->d_hi, o_hi, sizeof(int) * h->N, hipMemcpyDeviceToDevice, h->own_stream));
    }
    for (void* q : {(void*)o_cand, (void*)o_from, (void*)o_to, (void*)o_lo, (void*)o_hi, (void*)o_order, (void*)o_live, (void*)o_perm})
        if (q) h->retired.push_back(q);
    h->order_stale = true;
    return IPC_OK;
}

extern "C" int ipc_destroy(ipc_engine_t* h)
{
    if (!h) return IPC_OK;
    hipSetDevice(h->device);
    DevicePipeline& dp = device_pipeline(h->device);
    std::lock_guard<std::recursive_mutex> run_lk(dp.run_mu);               // (no other thread's check may be resetting this engine's pipeline)
    spec_quiesce(h, true);
    {
        std::lock_guard<std::mutex> lk(g_pipeline_mu);
        if (dp.active == h) dp.active = nullptr;
        dp.engines.erase(std::remove(dp.engines.begin(), dp.engines.end(), h), dp.engines.end());
    }
    free_candidates(h);
    hipFree(h->d_chain); hipFree(h->d_chain_rec); hipFree(h->d_chain_blk); hipFree(h->d_pose0); hipFree(h->d_counters); hipFree(h->d_offsets); hipFree(h->d_wave_ctr);
    hipFree(h->d_cells); hipFree(h->d_chi); hipFree(h->d_chitot); hipFree(h->d_meta);
    hipFree(h->d_upper); hipFree(h->d_bits); hipFree(h->d_acc); hipFree(h->d_failed);
    hipFree(h->d_lit_cells); hipFree(h->d_lit_idx); hipFree(h->d_lit_chi); hipFree(h->d_lit_chitot); hipFree(h->d_lit_meta);
    hipFree(h->d_slot_off); hipFree(h->d_recount); if (h->h_recount) hipHostFree(h->h_recount);
    hipFree(h->d_chain1); if (h->d_open != h->d_pose0) hipFree(h->d_open); hipFree(h->d_cur);
    delete h->cluster;
    delete h->cluster3;
    if (h->d_prof) {
        unsigned long long p[kProfN] = {};
        hipDeviceSynchronize();
        hipMemcpy(p, h->d_prof, sizeof(p), hipMemcpyDeviceToHost);
        static const char* names[kProfN] = {"total", "pre", "handoff", "assemble", "factor", "factor_work", "factor_wait", "backsolve",
                                            "post", "trial", "rest", "iterations", "steps", "help_dt", "help_solve", "help_update", "help_wait",
                                            "look_load", "look_solve", "look_fill", "look_potrf", "look_publish", "bs_dots", "bs_prefetch", "bs_sync", "bs_triangle",
                                            "start_skew", "start_skew_of_launches_over_1_ms", "band_launches", "tile_block", "tile_select", "tile_trsm", "tile_store"};
        fprintf(stderr, "{\"persist_profile_us\": {");
        for (int k = 0; k < kProfN; ++k)
            fprintf(stderr, "%s\"%s\": %.1f", k ? ", " : "", names[k], (k == kProfIterations || k == kProfSteps || k == kProfStartLaunches) ? (double)p[k] : p[k] * 0.01);
        fprintf(stderr, "}}\n");
        hipFree(h->d_prof);
    }
    for (auto& sl : h->slots) {
        if (sl.st) hipStreamSynchronize(sl.st);
        delete sl.s2; delete sl.s3;
        if (sl.done) hipEventDestroy(sl.done);
        // (sl.st belongs to the process: pipeline_stream)
    }
    for (auto& stt : h->spec_states) {
        if (stt.owned) hipFree(stt.d_poses);
        if (stt.ready) hipEventDestroy(stt.ready);
        if (stt.h_pred) hipHostFree(stt.h_pred);
        if (stt.pred_ev) hipEventDestroy(stt.pred_ev);
    }
    if (h->d_prof || getenv("IPC_SPEC_STATS"))
        fprintf(stderr, "{\"speculation\": {\"window\": %d, \"slots_in_use\": %d, \"streams_abreast\": %d, \"persist_timeouts\": %ld, \"ahead\": %d, \"launches\": %ld, \"results_used\": %ld, \"discarded\": %ld, "
                        "\"tentative_states\": %ld, \"promoted\": %ld, \"host_s_in_checks\": %.3f, \"host_s_launching\": %.3f, "
                        "\"host_s_tentative\": %.3f, "
                        "\"accept_solves\": %ld, \"accept_us_per_iteration\": %.1f, \"accept_ms_per_solve\": %.2f, "
                        "\"reject_solves\": %ld, \"reject_us_per_iteration\": %.1f, \"reject_ms_per_solve\": %.2f, "
                        "\"accept_ms_per_solve_on_the_device\": %.2f, \"reject_ms_per_solve_on_the_device\": %.2f, "
                        "\"empty_slot_share\": {\"behind_an_expected_accept\": %.3f, \"no_candidate_within_the_look_ahead\": %.3f, \"cu_budget\": %.3f, \"cluster_too_large\": %.3f}, "
                        "\"kept_results_by_expectation\": {\"expected_accept\": {\"accepted\": %ld, \"rejected\": %ld}, \"expected_reject\": {\"accepted\": %ld, \"rejected\": %ld}, \"none\": {\"accepted\": %ld, \"rejected\": %ld}}}}\n",
                h->spec_window, h->spec_active, h->stream_concurrency, h->persist_timeouts, h->spec_ahead, h->spec_launches, h->spec_hits, h->spec_wasted, h->spec_tentative, h->spec_promoted,
                h->spec_t_total, h->spec_t_launch, h->spec_t_tent,
                h->st_acc_n, 1e6 * h->st_acc_s / std::max(1L, h->st_acc_it), 1e3 * h->st_acc_s / std::max(1L, h->st_acc_n),
                h->st_rej_n, 1e6 * h->st_rej_s / std::max(1L, h->st_rej_it), 1e3 * h->st_rej_s / std::max(1L, h->st_rej_n),
                1e3 * h->st_acc_dev_s / std::max(1L, h->st_acc_n), 1e3 * h->st_rej_dev_s / std::max(1L, h->st_rej_n),
                (double)h->idle_why[0] / std::max(1ull, h->idle_why[4]), (double)h->idle_why[1] / std::max(1ull, h->idle_why[4]),
                (double)h->idle_why[2] / std::max(1ull, h->idle_why[4]), (double)h->idle_why[3] / std::max(1ull, h->idle_why[4]),
                h->pred_conf[2][1], h->pred_conf[2][0], h->pred_conf[1][1], h->pred_conf[1][0], h->pred_conf[0][1], h->pred_conf[0][0]);
    if (h->spec_log) fclose(h->spec_log);
    if (h->ev_commit) hipEventDestroy(h->ev_commit);
    if (h->h_abort) hipHostFree(h->h_abort);
    delete h->persist2;
    delete h->persist3;
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->ev_join_own) hipEventDestroy(h->ev_join_own);
    for (int k = 0; k < h->n_side; ++k) {
        if (h->ev_join[k]) hipEventDestroy(h->ev_join[k]);
        // (side[k] belongs to the process's stream pool)
    }
    if (h->own_stream) hipStreamDestroy(h->own_stream);
    delete h;
    return IPC_OK;
}


// Uploads the candidate list (file order); the consensus set and the current poses go back to the open-loop state.
static int upload_candidates(ipc_engine* h, int n, const int* ids, const double* meas, const double* info)
{
    for (int k = 0; k < n; ++k) {
        const int f = ids[2 * k], t = ids[2 * k + 1];
        if (f < 0 || t < 0 || f >= h->V || t >= h->V)
            return fail(IPC_ERR_ARG, "candidate %d joins vertex %d-%d outside 0..%d", k, f, t, h->V - 1);
        if (std::abs(f - t) < 2)
            return fail(IPC_ERR_ARG, "candidate %d joins adjacent vertices %d-%d (an odometry edge, "
                        "reference src/utils.cpp:184)", k, f, t);
    }
    HIPCHK(hipSetDevice(h->device));
    if (int rc = spec_quiesce(h, true)) return rc;
    free_candidates(h);
    h->last_cells = 0;
    h->plan_cached = false;
    h->ev_valid = false;
    h->cns.clear(); h->cns_dups = false;
    if (h->d_cur && h->d_open)
        HIPCHK(copy_d2d_now(h, h->d_cur, h->d_open, sizeof(double) * (h->dim == 2 ? 5 : 12) * (size_t)h->V));
    const int ms = h->dim == 2 ? 3 : 7, is = h->dim == 2 ? 6 : 21;
    h->h_cand_ids.assign(ids, ids + 2 * (size_t)n);                       // raw records (row assignment, appends)
    h->h_cand_meas.assign(meas, meas + (size_t)ms * n);
    h->h_cand_info.assign(info, info + (size_t)is * n);
    h->h_from.resize(n); h->h_to.resize(n); h->h_lo.resize(n); h->h_hi.resize(n);
    for (int k = 0; k < n; ++k) {
        h->h_from[k] = ids[2 * k]; h->h_to[k] = ids[2 * k + 1];
        h->h_lo[k] = std::min(h->h_from[k], h->h_to[k]); h->h_hi[k] = std::max(h->h_from[k], h->h_to[k]);
    }
    // cmpTime order (src/utils.cpp:379-390) with the (max id, index) tie-break
    h->order.resize(n);
    std::iota(h->order.begin(), h->order.end(), 0);
    std::stable_sort(h->order.begin(), h->order.end(), [&](int a, int b) { return h->h_hi[a] < h->h_hi[b]; });
    h->porder = h->order;
    h->handed.assign(n, 0);
    h->ppos.assign(n, 0);
    for (int q = 0; q < n; ++q) h->ppos[h->porder[q]] = q;
    if (n == 0) return IPC_OK;
    if (int rc = alloc_candidates(h, (n + 63) & ~63)) return rc;
    h->N = n;
    HIPCHK(hipMemcpy(h->d_from, h->h_from.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_to, h->h_to.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_lo, h->h_lo.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_hi, h->h_hi.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_order, h->order.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    double *d_m = nullptr, *d_i = nullptr;
    HIPCHK(hipMalloc(&d_m, sizeof(double) * ms * n));
    HIPCHK(hipMalloc(&d_i, sizeof(double) * is * n));
    HIPCHK(hipMemcpy(d_m, meas, sizeof(double) * ms * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_i, info, sizeof(double) * is * n, hipMemcpyHostToDevice));
    HIPCHK(hipStreamSynchronize(nullptr));                     // (NULL-stream copies are not ordered against the engine's non-blocking streams: copy_d2d_now)
    if (h->dim == 2)
        hipLaunchKernelGGL(k_se2_prep, dim3((n + 255) / 256), dim3(256), 0, h->own_stream, n, d_m, d_i, 1.0,
                           h->d_cand, h->cstride);
    else
        hipLaunchKernelGGL(k_se3_prep, dim3((n + 63) / 64), dim3(64), 0, h->own_stream, n, d_m, d_i, 1.0,
                           h->d_cand, h->cstride);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->own_stream));
    HIPCHK(hipFree(d_m));
    HIPCHK(hipFree(d_i));
    return IPC_OK;
}

extern "C" int ipc_set_candidates(ipc_engine_t* h, int n, const int* ids, const double* meas, const double* info)
{
    if (!h) return fail(IPC_ERR_ARG, "ipc_set_candidates: NULL handle");
    if (n < 0 || (n > 0 && (!ids || !meas || !info))) return fail(IPC_ERR_ARG, "ipc_set_candidates: bad arrays");
    return upload_candidates(h, n, ids, meas, info);
}

static void spec_insert_position(ipc_engine* h, int k);
static void spec_move_to_head(ipc_engine* h, int k);


// One more candidate at the end of the list (the harness hands IPC::agreementCheck an edge nobody announced,
// reference src/simulation.cpp:34-47).  The consensus set, the poses and whatever the pipeline has in flight stay as they
// are: one record is written in place by a one-thread kernel that carries it in its arguments -- no re-upload, no
// allocation (the arrays grow geometrically), no hipFree, no host synchronisation.
extern "C" int ipc_append_candidate(ipc_engine_t* h, const int* ids, const double* meas, const double* info, int* index_out)
{
    if (!h || !ids || !meas || !info) return fail(IPC_ERR_ARG, "ipc_append_candidate: NULL argument");
    const int ms = h->dim == 2 ? 3 : 7, is = h->dim == 2 ? 6 : 21;
    const int f = ids[0], t = ids[1], k = h->N;
    if (f < 0 || t < 0 || f >= h->V || t >= h->V)
        return fail(IPC_ERR_ARG, "candidate %d joins vertex %d-%d outside 0..%d", k, f, t, h->V - 1);
    if (std::abs(f - t) < 2)
        return fail(IPC_ERR_ARG, "candidate %d joins adjacent vertices %d-%d (an odometry edge, reference src/utils.cpp:184)", k, f, t);
    HIPCHK(hipSetDevice(h->device));
    if (k + 1 > h->cstride) { if (int rc = grow_candidates(h, k + 1)) return rc; }
    if (!h->ev_cand) HIPCHK(hipEventCreateWithFlags(&h->ev_cand, hipEventDisableTiming));
    RawCandidate r{};
    for (int q = 0; q < ms; ++q) r.meas[q] = meas[q];
    for (int q = 0; q < is; ++q) r.info[q] = info[q];
    r.from = f; r.to = t;
    if (h->dim == 2)
        hipLaunchKernelGGL(k_prep_one<2>, dim3(1), dim3(64), 0, h->own_stream, r, k, h->d_cand, h->cstride, h->d_from, h->d_to, h->d_lo, h->d_hi);
    else
        hipLaunchKernelGGL(k_prep_one<3>, dim3(1), dim3(64), 0, h->own_stream, r, k, h->d_cand, h->cstride, h->d_from, h->d_to, h->d_lo, h->d_hi);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(h->ev_cand, h->own_stream));                   // (what the pipeline's streams wait for before a launch)
    h->cand_event = true;
    h->h_cand_ids.insert(h->h_cand_ids.end(), ids, ids + 2);
    h->h_cand_meas.insert(h->h_cand_meas.end(), meas, meas + ms);
    h->h_cand_info.insert(h->h_cand_info.end(), info, info + is);
    h->h_from.push_back(f); h->h_to.push_back(t);
    h->h_lo.push_back(std::min(f, t)); h->h_hi.push_back(std::max(f, t));
    // (max id, index): behind every candidate that ends at or before its later vertex
    const int hi = h->h_hi[k];
    const auto it = std::upper_bound(h->order.begin(), h->order.end(), hi, [&](int v, int c) { return v < h->h_hi[c]; });
    h->order.insert(it, k);
    h->N = k + 1;
    h->order_stale = true;
    h->last_cells = 0;                                   // the cells of the last matrix solve are those of the shorter list
    h->plan_cached = false;
    h->ev_valid = false;
    if (h->d_slot) { h->retired.push_back(h->d_slot); h->d_slot = nullptr; }
    h->slot_world = 0;
    {
        std::lock_guard<std::recursive_mutex> run_lk(device_pipeline(h->device).run_mu);
        spec_insert_position(h, k);
    }
    if (index_out) *index_out = k;
    return IPC_OK;
}

// Entry of every matrix-mode call that launches on `st`.  (i) The pipeline of the faithful mode must not be on the GPU
// beside the cell kernels: its persistent launches meet at grid barriers and need all their workgroups resident, which a
// full machine does not grant -- whatever it has in flight is given up (it restarts with the next
// ipc_agreement_check).  (ii) Records appended since the last call were written on own_stream: `st` waits for them, and
// the processing order on the device is brought up to date.
static int matrix_mode_enter(ipc_engine* h, hipStream_t st)
{
    if (!h->slots.empty() && h->spec_head >= 0) { if (int rc = spec_quiesce(h, true)) return rc; }
    {
        // ... and so must the pipelines of the OTHER engines of this device (round 6: they are no longer reset by each other's checks)
        DevicePipeline& dp = device_pipeline(h->device);
        std::lock_guard<std::recursive_mutex> run_lk(dp.run_mu);
        for (ipc_engine* o : dp.engines)
            if (o != h && !o->slots.empty() && o->spec_head >= 0) { if (int rc = spec_reset(o)) return rc; }
    }
    if (h->order_stale && h->N > 0) {
        HIPCHK(hipMemcpyAsync(h->d_order, h->order.data(), sizeof(int) * h->N, hipMemcpyHostToDevice, h->own_stream));
        HIPCHK(hipStreamSynchronize(h->own_stream));     // (pageable source: the copy has left the host vector when this returns)
        h->order_stale = false;
    }
    if (h->cand_event && st != h->own_stream) HIPCHK(hipStreamWaitEvent(st, h->ev_cand, 0));
    return IPC_OK;
}

extern "C" int ipc_candidate_order(ipc_engine_t* h, int* order_out)
{
    if (!h || !order_out) return fail(IPC_ERR_ARG, "ipc_candidate_order: NULL argument");
    std::copy(h->order.begin(), h->order.end(), order_out);
    return IPC_OK;
}

extern "C" int ipc_initial_poses(ipc_engine_t* h, double* poses_out)
{
    if (!h || !poses_out) return fail(IPC_ERR_ARG, "ipc_initial_poses: NULL argument");
    HIPCHK(hipSetDevice(h->device));
    const int ps = h->dim == 2 ? 3 : 12;
    std::vector<double> tmp(ps * (size_t)h->V);
    HIPCHK(hipMemcpy(tmp.data(), h->d_pose0, sizeof(double) * tmp.size(), hipMemcpyDeviceToHost));
    for (int i = 0; i < h->V; ++i)
        for (int f = 0; f < ps; ++f) poses_out[ps * (size_t)i + f] = tmp[(size_t)f * h->V + i];
    return IPC_OK;
}

// debug side channel of the cell kernels (phase timing builds only; NULL in production)
static double* dbg_buffer()
{
#if defined(IPC_PHASE_TIMING)
    static double* dbgbuf = nullptr;
    if (!dbgbuf) { hipMalloc(&dbgbuf, sizeof(double) * 8 * 4096); hipMemset(dbgbuf, 0, sizeof(double) * 8 * 4096); }
    return dbgbuf;
#else
    return nullptr;
#endif
}

static Se2View make_view(const ipc_engine* h)
{
    Se2View P;
    P.chain = h->d_chain; P.estride = h->estride; P.pose0 = h->d_pose0; P.V = h->V;
    P.chain_rec = h->d_chain_rec;
    P.cand = h->d_cand; P.cstride = h->cstride; P.cand_from = h->d_from; P.cand_to = h->d_to;
    P.dbg = dbg_buffer();
    P.term_eps = h->term_eps;
    return P;
}

static Se3View make_view3(const ipc_engine* h)
{
    Se3View P;
    P.chain = h->d_chain; P.estride = h->estride; P.pose0 = h->d_pose0; P.V = h->V;
    P.chain_rec = h->d_chain_rec;
    P.chain_blk = h->d_chain_blk;
    P.term_eps = h->term_eps;
    P.cand = h->d_cand; P.cstride = h->cstride; P.cand_from = h->d_from; P.cand_to = h->d_to;
    P.dbg = dbg_buffer();
    return P;
}

static int ensure_incremental(ipc_engine* h, const char* who);
static PoseArr pose_arr(double* base, int V);

// One cluster solve (chain lo..hi of the records `chain` + the loops `members`, from the poses `src`: SE2 [5][V],
// SE3 [12][V]) on the engine's stream, by the device-resident dog-leg or (IPC_CLUSTER_MODE=host) the host-driven one.
// Blocks until the result record is back; the optimised poses stay in the solver (cluster_result2 / cluster_result3).
static hipError_t cluster_solve(ipc_engine* h, const double* chain, double* src, int lo, int hi, const std::vector<int>& members,
                                int iters, ClusterOut& o, bool force_host = false)
{
    // (nothing of the candidate pipeline may share the GPU with this launch: its workgroups and the pipeline's together
    // could exceed the CUs, and workgroups that wait at grid barriers must all be resident)
    if (!h->slots.empty() && h->spec_head >= 0) { if (int rc = spec_quiesce(h, true)) return hipErrorUnknown; }
    h->last_persist = !force_host && h->persist && PersistSolver<PersistSe2>::fits(hi - lo, (int)members.size());
    if (h->last_persist) {
        // A LOST launch (a grid barrier timed out: some workgroup of the launch never became resident, e.g. beside a foreign
        // tenant's kernels) says nothing about the problem: the same launch is tried again -- the pipeline is quiesced here, so it
        // has the GPU to itself as far as this process goes -- twice, before the host-driven kernels (which need no
        // co-residency, one launch per phase) take over.  Round 5 went straight to them.
        bool lost = false;
        for (int attempt = 0; attempt < 3; ++attempt) {
            if (h->dim == 3) {
                IPC_CL_CHK(h->persist3->launch(h->own_stream, chain, h->estride, h->d_cand, h->cstride, src, h->V, lo, hi, members,
                                               h->h_from.data(), h->h_to.data(), iters));
                IPC_CL_CHK(h->persist3->wait(o));
                lost = h->persist3->timed_out();
            } else {
                IPC_CL_CHK(h->persist2->launch(h->own_stream, chain, h->estride, h->d_cand, h->cstride, src, h->V, lo, hi, members,
                                               h->h_from.data(), h->h_to.data(), iters));
                IPC_CL_CHK(h->persist2->wait(o));
                lost = h->persist2->timed_out();
            }
            if (!lost) break;
            ++h->persist_timeouts;
            if (attempt < 2) ++h->persist_relaunches;
        }
        if (lost) o.flags |= 2;
        // a non-positive pivot in the capacitance factorisation: g2o would retry with Levenberg damping -- the
        // host-driven solver does (literal normal equations, banded or dense), from the same start
        if (!lost && (!(o.flags & 2) || !h->lm_retry)) return hipSuccess;
        h->last_persist = false;
        ++h->lm_fallbacks;
    }
    if (h->dim == 3)
        return h->cluster3->solve(h->own_stream, chain, h->estride, h->d_cand, h->cstride, src, h->V, lo, hi, members,
                                  h->h_from.data(), h->h_to.data(), iters, o, nullptr);
    return h->cluster->solve(h->own_stream, chain, h->estride, h->d_cand, h->cstride, pose_arr(src, h->V), lo, hi, members,
                             h->h_from.data(), h->h_to.data(), iters, o, nullptr);
}
static PoseArr cluster_result2(const ipc_engine* h)
{
    if (!h->last_persist) return h->cluster->result();
    return h->persist2->result_in_second() ? h->persist2->dev().Xn : h->persist2->dev().X;
}
static const double* cluster_result3(const ipc_engine* h, int& ld)
{
    if (!h->last_persist) { ld = h->cluster3->ld(); return h->cluster3->result(); }
    ld = h->persist3->ld();
    return h->persist3->result_in_second() ? h->persist3->dev().Xn : h->persist3->dev().X;
}

// Cells beyond the capacity of every cell kernel (SE3 > 4096 poses, SE2 > 16384 with the default
// policies): the same check (reference src/consensus_utils.cpp:7-22 on chain lo..hi + the one or two
// loop edges, open-loop start, fast / slow iteration base) through the cluster solver, whose state
// lives in HBM.  Host driven and serial -- meant for the handful of full-span loops of a long
// trajectory (reference cfg/3D/GRID_params.yaml), not as a fast path.
static int solve_long_cells(ipc_engine* h, hipStream_t st, int nb, const unsigned* counts, const unsigned* offsets)
{
    if (int rc = ensure_incremental(h, "ipc_solve_rows")) return rc;
    HIPCHK(hipStreamSynchronize(st));
    for (int nl = 1; nl <= 2; ++nl) {
        const int s = (nl == 1 ? 0 : (kMaxBins + 1)) + nb;
        const unsigned n = counts[s];
        if (!n) continue;
        std::vector<int2> cells(n);
        HIPCHK(hipMemcpy(cells.data(), h->d_cells + offsets[s], sizeof(int2) * n, hipMemcpyDeviceToHost));
        std::vector<double> chi(n), tot(n);
        std::vector<int4> meta(n);
        for (unsigned c = 0; c < n; ++c) {
            const int i = cells[c].x, j = cells[c].y;
            const int lo = std::min(h->h_lo[i], h->h_lo[j]), hi = std::max(h->h_hi[i], h->h_hi[j]);
            std::vector<int> members{i};
            if (nl == 2) members.push_back(j);
            int iters = nl == 1 ? h->prm.fast_reject_iter_base : h->prm.slow_reject_iter_base;
            if ((hi - lo) + nl > 100) iters *= 5;                              // consensus_utils.cpp:12-13
            ClusterOut o;
            HIPCHK(cluster_solve(h, h->d_chain, h->d_open, lo, hi, members, iters, o));
            chi[c] = o.max_chi2; tot[c] = o.chi2_total;
            meta[c] = make_int4(o.iterations, o.tries, o.flags, o.evals);
        }
        HIPCHK(hipStreamSynchronize(h->own_stream));
        HIPCHK(hipMemcpy(h->d_chi + offsets[s], chi.data(), sizeof(double) * n, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_chitot + offsets[s], tot.data(), sizeof(double) * n, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_meta + offsets[s], meta.data(), sizeof(int4) * n, hipMemcpyHostToDevice));
    }
    HIPCHK(hipStreamSynchronize(nullptr));                     // (NULL-stream copies are not ordered against the engine's non-blocking streams: copy_d2d_now)
    return IPC_OK;
}

static int ensure_row_map(ipc_engine* h, int world)
{
    if (h->d_slot && h->slot_world == world) return IPC_OK;
    std::vector<int> slot(h->N);
    if (int rc = ipc_row_assignment(h->N, h->h_cand_ids.data(), world, h->row_policy, slot.data())) return rc;
    if (!h->d_slot) HIPCHK(hipMalloc(&h->d_slot, sizeof(int) * h->N));
    HIPCHK(hipMemcpy(h->d_slot, slot.data(), sizeof(int) * h->N, hipMemcpyHostToDevice));
    // the rows of each rank, in the order its planning pass visits them: by the first vertex of the candidate's interval,
    // then by index (cell lists sorted by chain position, see k_plan)
    const int rpr = ipc_rows_per_rank(h->N, world);
    std::vector<int> perm(h->N);
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) {
        const int ra = slot[a] / rpr, rb = slot[b] / rpr;
        return ra != rb ? ra < rb : h->h_lo[a] < h->h_lo[b];
    });
    h->row_group_off.assign(world + 1, 0);
    for (int i = 0; i < h->N; ++i) ++h->row_group_off[slot[i] / rpr + 1];
    for (int r = 0; r < world; ++r) h->row_group_off[r + 1] += h->row_group_off[r];
    HIPCHK(hipMemcpy(h->d_rowperm, perm.data(), sizeof(int) * h->N, hipMemcpyHostToDevice));
    HIPCHK(hipStreamSynchronize(nullptr));                     // (NULL-stream copies are not ordered against the engine's non-blocking streams: copy_d2d_now)
    h->slot_world = world;
    return IPC_OK;
}

// Cells whose capacitance factorisation met a non-positive pivot (flags & 2: degenerate information matrices, NaN
// poses): g2o retries such a solve with Levenberg damping.  The cell kernels cannot (see cluster_common.hpp), so the
// few cells concerned are solved again by the host-driven cluster solver, which can -- same check, open-loop start.
// Borderline cells (see ipc_engine::borderline_band) are collected by the same pass, listed as ~c.
__device__ __forceinline__ int slot_of_cell(const unsigned* slot_off, int nslots, int c)
{
    int lo = 0, hi = nslots;                                 // last slot whose first cell is <= c (empty slots share an offset: take the last)
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (slot_off[mid] <= (unsigned)c) lo = mid; else hi = mid; }
    return lo;
}
// Failed cells -> list (host-driven Levenberg retry, rare); borderline cells -> the compact list of their slot (lit_cells /
// lit_idx at the slot's own offset: a slot has room for all of its cells), counted per slot in recount[0 .. nslots).
__global__ void k_collect_failed(int ncells, const int4* meta, const double* chi, const int2* cells, double fast_th,
                                 double slow_th, double band, bool want_failed, int cap, int* list, int* recount,
                                 const unsigned* slot_off, int nslots, int2* lit_cells, int* lit_idx, int long_bin)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncells) return;
    const bool failed = want_failed && (meta[c].z & 2);
    bool border = false;
    const int2 cc = cells[c];
    if (!failed && band > 0.0) {
        const double th = cc.x == cc.y ? fast_th : slow_th;
        border = fabs(chi[c] - th) <= band * th;                     // (NaN: no)
    }
    if (failed) {
        const int q = atomicAdd(recount + nslots, 1);
        if (q < cap) list[q] = c;
    } else if (border) {
        const int sl = slot_of_cell(slot_off, nslots, c);
        if (sl == long_bin || sl == (nslots >> 1) + long_bin) {
            // a chain beyond every cell kernel (solve_long_cells): no kernel re-solves that slot -- onto the host list, where
            // the cluster solver runs it again with g2o's literal loop (resolve_failed_cells tells the two kinds apart by flags & 2)
            const int q = atomicAdd(recount + nslots, 1);
            if (q < cap) list[q] = c;
            return;
        }
        const int q = atomicAdd(recount + sl, 1);
        lit_cells[slot_off[sl] + q] = cc;
        lit_idx[slot_off[sl] + q] = c;
    }
}
// ---- slow cells first (round 5) ---------------------------------------------------------------------------------
// A bin's launch draws cells from the head of its list; a cell that runs to the iteration cap (500 iterations against a
// mean of ~30: ~10 ms on a 1000-pose chain) and is drawn late IS the tail of the launch -- nothing on one GPU, whose CUs
// always find other work, but 15 - 20 % of a step once the rows are spread over 8 ranks (54 ms of work per rank).  Which
// cells are slow is known exactly after one solve of the list: the step that plans a candidate list ends by moving the
// cells that took >= slow_first_iterations iterations to the head of their slot, everything else keeps its order (sorted by
// chain position: L2 locality).  Results move with their cells; every cell's arithmetic is unchanged.
__global__ void k_slow_flags(int ncells, const int4* meta, int threshold, int* normal)     // normal[c] = 1: stays behind the slow cells
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c <= ncells) normal[c] = (c < ncells && meta[c].x < threshold) ? 1 : 0;
}
// in-place exclusive prefix sum of n ints, one workgroup
__global__ __launch_bounds__(1024) void k_scan_int(int* a, int n)
{
    __shared__ int wsum[16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int carry = 0;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const int v0 = i < n ? a[i] : 0;
        int v = v0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o, 64); if (lane >= o) v += t; }
        if (lane == 63) wsum[wave] = v;
        __syncthreads();
        int off = carry, tot = carry;
        for (int w = 0; w < 16; ++w) { if (w < wave) off += wsum[w]; tot += wsum[w]; }
        if (i < n) a[i] = off + v - v0;
        carry = tot;
        __syncthreads();
    }
}
__global__ void k_slow_first_permute(int ncells, const unsigned* slot_off, int nslots, const int* excl_normal, const int2* cells,
                                     const double* chi, const double* chitot, const int4* meta, int2* cells2, double* chi2,
                                     double* chitot2, int4* meta2)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncells) return;
    const int sl = slot_of_cell(slot_off, nslots, c);
    const int s0 = (int)slot_off[sl], s1 = (int)slot_off[sl + 1];
    const int n0 = excl_normal[s0], nc = excl_normal[c], n1 = excl_normal[s1];
    const bool normal = excl_normal[c + 1] != nc;
    const int nslow = (s1 - s0) - (n1 - n0);
    const int dst = normal ? s0 + nslow + (nc - n0) : s0 + ((c - s0) - (nc - n0));
    cells2[dst] = cells[c]; chi2[dst] = chi[c]; chitot2[dst] = chitot[c]; meta2[dst] = meta[c];
}

// the literal re-solves back to where their cells sit
__global__ void k_scatter_literal(int ncells, const unsigned* slot_off, int nslots, const int* recount, const int* lit_idx,
                                  const double* lit_chi, const double* lit_chitot, const int4* lit_meta, double* chi, double* chitot, int4* meta)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ncells) return;
    const int sl = slot_of_cell(slot_off, nslots, t);
    if ((int)(t - slot_off[sl]) >= recount[sl]) return;
    const int c = lit_idx[t];
    chi[c] = lit_chi[t]; chitot[c] = lit_chitot[t]; meta[c] = lit_meta[t];
}
static int resolve_failed_cells(ipc_engine* h, int n)
{
    h->last_lm_cells = 0;
    if (n == 0) return IPC_OK;
    n = std::min(n, h->failed_cap);
    if (int rc = ensure_incremental(h, "ipc_solve_rows")) return rc;
    std::vector<int> idx(n);
    HIPCHK(hipMemcpy(idx.data(), h->d_failed, sizeof(int) * n, hipMemcpyDeviceToHost));
    std::sort(idx.begin(), idx.end());
    for (int q = 0; q < n; ++q) {
        int2 cell;
        HIPCHK(hipMemcpy(&cell, h->d_cells + idx[q], sizeof(int2), hipMemcpyDeviceToHost));
        const int i = cell.x, j = cell.y, nl = i == j ? 1 : 2;
        const int lo = std::min(h->h_lo[i], h->h_lo[j]), hi = std::max(h->h_hi[i], h->h_hi[j]);
        std::vector<int> members{i};
        if (nl == 2) members.push_back(j);
        int iters = nl == 1 ? h->prm.fast_reject_iter_base : h->prm.slow_reject_iter_base;
        if ((hi - lo) + nl > 100) iters *= 5;                                  // consensus_utils.cpp:12-13
        ClusterOut o;
        int4 was;
        HIPCHK(hipMemcpy(&was, h->d_meta + idx[q], sizeof(int4), hipMemcpyDeviceToHost));
        if (was.z & 2) {
            HIPCHK(cluster_solve(h, h->d_chain, h->d_open, lo, hi, members, iters, o, true));    // (damping: host-driven solver)
            ++h->last_lm_cells;
        } else {
            // a borderline cell of the long slots (k_collect_failed): g2o's literal trial loop, convergence test off
            const double te = h->term_eps;
            auto set_eps = [&](double v) {
                if (h->cluster) h->cluster->term_eps = v;
                if (h->cluster3) h->cluster3->term_eps = v;
                if (h->persist2) h->persist2->term_eps = v;
                if (h->persist3) h->persist3->term_eps = v;
            };
            set_eps(0.0);
            const hipError_t e = cluster_solve(h, h->d_chain, h->d_open, lo, hi, members, iters, o);
            set_eps(te);
            HIPCHK(e);
            ++h->last_literal_cells;
        }
        const int4 meta = make_int4(o.iterations, o.tries, o.flags, o.evals);
        HIPCHK(hipMemcpy(h->d_chi + idx[q], &o.max_chi2, sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_chitot + idx[q], &o.chi2_total, sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_meta + idx[q], &meta, sizeof(int4), hipMemcpyHostToDevice));
    }
    HIPCHK(hipStreamSynchronize(nullptr));                     // (NULL-stream copies are not ordered against the engine's non-blocking streams: copy_d2d_now)
    return IPC_OK;
}

static int solve_rows_impl(ipc_engine* h, int rank, int world, uint64_t* d_upper, void* stream, int phase);

extern "C" int ipc_solve_rows(ipc_engine_t* h, int rank, int world, uint64_t* d_upper, void* stream)
{
    if (!h) return fail(IPC_ERR_ARG, "ipc_solve_rows: NULL handle");
    if (h->N <= 0) return fail(IPC_ERR_STATE, "ipc_solve_rows: no candidates set");
    if (world < 1 || rank < 0 || rank >= world) return fail(IPC_ERR_ARG, "ipc_solve_rows: rank %d of %d", rank, world);
    if (!d_upper) return fail(IPC_ERR_ARG, "ipc_solve_rows: d_upper is NULL");
    return solve_rows_impl(h, rank, world, d_upper, stream, 0);
}

// phase 0: all cells of the rank's rows.  Set-only mode (one rank): phase 1 = the diagonal cells, phase 2 = the pair cells
// among the candidates whose diagonal bit is set in d_upper (left in place; the pair bits are OR-ed into it).
static int solve_rows_impl(ipc_engine* h, int rank, int world, uint64_t* d_upper, void* stream, int phase)
{
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->own_stream;
    const int N = h->N, words = (N + 63) / 64, rpr = ipc_rows_per_rank(N, world);
    if (int rc = matrix_mode_enter(h, st)) return rc;
    if (int rc = ensure_row_map(h, world)) return rc;
    const BinCaps bc = h->plan.caps;
    const int nb = bc.n;
    constexpr int NS = 2 * (kMaxBins + 1);
    if (phase != 2) HIPCHK(hipMemsetAsync(d_upper, 0, sizeof(uint64_t) * (size_t)rpr * words, st));
    unsigned counts[NS], offsets[NS];
    size_t total = 0;
    // The cell lists of this rank (d_cells, grouped by slot = (loop count, chain-length bin)) depend on the candidates,
    // the rank and the world only: a repeated step reuses them and skips the two planning passes with their read-back.
    // (The set-only phases plan from the diagonal bits of the step and are never cached.)
    const bool cached = phase == 0 && h->plan_cached && h->plan_rank == rank && h->plan_world == world;
    if (cached) {
        std::copy(h->plan_counts.begin(), h->plan_counts.end(), counts);
        std::copy(h->plan_offsets.begin(), h->plan_offsets.end(), offsets);
        total = h->plan_total;
    } else {
        h->plan_cached = false;
        // pass 1: count
        HIPCHK(hipMemsetAsync(h->d_counters, 0, sizeof(unsigned) * NS * kPlanSub, st));
        const int nrows = h->row_group_off[rank + 1] - h->row_group_off[rank];
        const int* rows = h->d_rowperm + h->row_group_off[rank];
        const dim3 pgrid((N + 255) / 256, std::max(1, std::min(nrows, 2048))), pblock(256);     // few fat blocks: dispatching one block per (row, 256 candidates) cost more than the compares
        hipLaunchKernelGGL(k_plan, pgrid, pblock, 0, st, N, h->d_lo, h->d_hi, rows, nrows, bc, h->d_counters,
                           h->d_offsets, (int2*)nullptr, 0, phase, (const unsigned long long*)d_upper, words);
        HIPCHK(hipGetLastError());
        static thread_local unsigned subcounts[NS * kPlanSub], suboffsets[NS * kPlanSub];
        HIPCHK(hipMemcpyAsync(subcounts, h->d_counters, sizeof subcounts, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        for (int s = 0; s < NS; ++s) {
            counts[s] = 0;
            for (int q = 0; q < kPlanSub; ++q) counts[s] += subcounts[s * kPlanSub + q];
        }
        for (int s = 0; s < NS; ++s) { offsets[s] = (unsigned)total; total += counts[s]; }
        for (int s = 0; s < NS; ++s) {                    // a slot's sub-lists are contiguous: one cell list per slot
            unsigned o = offsets[s];
            for (int q = 0; q < kPlanSub; ++q) { suboffsets[s * kPlanSub + q] = o; o += subcounts[s * kPlanSub + q]; }
        }
        if (total > h->cells_cap) {
            hipFree(h->d_cells); hipFree(h->d_chi); hipFree(h->d_chitot); hipFree(h->d_meta);
            hipFree(h->d_lit_cells); hipFree(h->d_lit_idx); hipFree(h->d_lit_chi); hipFree(h->d_lit_chitot); hipFree(h->d_lit_meta);
            h->d_cells = h->d_lit_cells = nullptr; h->d_chi = h->d_chitot = h->d_lit_chi = h->d_lit_chitot = nullptr;
            h->d_meta = h->d_lit_meta = nullptr; h->d_lit_idx = nullptr;
            h->cells_cap = total + total / 8 + 1024;
            HIPCHK(hipMalloc(&h->d_cells, sizeof(int2) * h->cells_cap));
            HIPCHK(hipMalloc(&h->d_chi, sizeof(double) * h->cells_cap));
            HIPCHK(hipMalloc(&h->d_chitot, sizeof(double) * h->cells_cap));
            HIPCHK(hipMalloc(&h->d_meta, sizeof(int4) * h->cells_cap));
            HIPCHK(hipMalloc(&h->d_lit_cells, sizeof(int2) * h->cells_cap));
            HIPCHK(hipMalloc(&h->d_lit_idx, sizeof(int) * (h->cells_cap + 1)));      // (+ 1: k_slow_flags / k_scan_int write and scan total + 1 entries)
            HIPCHK(hipMalloc(&h->d_lit_chi, sizeof(double) * h->cells_cap));
            HIPCHK(hipMalloc(&h->d_lit_chitot, sizeof(double) * h->cells_cap));
            HIPCHK(hipMalloc(&h->d_lit_meta, sizeof(int4) * h->cells_cap));
        }
        if (!h->d_slot_off) {
            HIPCHK(hipMalloc(&h->d_slot_off, sizeof(unsigned) * (NS + 1)));
            HIPCHK(hipMalloc(&h->d_recount, sizeof(int) * (NS + 1)));
            HIPCHK(hipHostMalloc(&h->h_recount, sizeof(int) * (NS + 1)));
        }
        // pass 2: fill
        static thread_local unsigned slot_off[NS + 1];
        for (int s = 0; s < NS; ++s) slot_off[s] = offsets[s];
        slot_off[NS] = (unsigned)total;
        HIPCHK(hipMemcpyAsync(h->d_slot_off, slot_off, sizeof slot_off, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(h->d_offsets, suboffsets, sizeof suboffsets, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemsetAsync(h->d_counters, 0, sizeof(unsigned) * NS * kPlanSub, st));
        hipLaunchKernelGGL(k_plan, pgrid, pblock, 0, st, N, h->d_lo, h->d_hi, rows, nrows, bc, h->d_counters,
                           h->d_offsets, h->d_cells, 1, phase, (const unsigned long long*)d_upper, words);
        HIPCHK(hipGetLastError());
        if (phase == 0) {
            h->plan_counts.assign(counts, counts + NS);
            h->plan_offsets.assign(offsets, offsets + NS);
            h->plan_total = total; h->plan_rank = rank; h->plan_world = world;
            h->plan_cached = true;
        }
    }
    // cells whose chain is longer than the largest kernel variant of the policy go through the cluster
    // solver below (one at a time, state in HBM: no length limit) instead of failing the matrix
    const unsigned n_long = counts[nb] + counts[(kMaxBins + 1) + nb];
    // solve: longest chains first
    Se2View P = make_view(h);
    Se3View P3 = make_view3(h);
    const SolveParams sp{h->prm.fast_reject_iter_base, h->prm.slow_reject_iter_base};
    // one slot's cells through the kernel variant of its bin
    auto launch_slot = [&](int b, int nl, int n, const int2* cells, const CellOut& out, hipStream_t ls, unsigned* ctr) -> hipError_t {
        int var = h->plan.variant[b];
        if (h->plan.latency_variant[b] >= 0 && n < h->plan.latency_below[b] * h->n_cu) var = h->plan.latency_variant[b];
        if (h->dim == 2)
            return var >= kQuadVariantBase   ? launch_se2_quad(nl, var - kQuadVariantBase, n, ls, P, cells, sp, out, ctr, h->n_cu)
                   : var >= kPairVariantBase ? launch_se2_pair(nl, var - kPairVariantBase, n, ls, P, cells, sp, out, ctr, h->n_cu)
                   : var >= kWaveVariantBase ? launch_se2_wave(nl, var - kWaveVariantBase, n, ls, P, cells, sp, out, ctr, h->n_cu)
                                             : launch_se2_block(nl, var, n, ls, P, cells, sp, out);
        if (var >= kLdsVariantBase3) return launch_se3_lds(nl, var - kLdsVariantBase3, n, ls, P3, cells, sp, out, ctr, h->n_cu);
        return launch_se3_block(nl, var, n, ls, P3, cells, sp, out);
    };
    int launches = 0;
    HIPCHK(hipMemsetAsync(h->d_wave_ctr, 0, sizeof(unsigned) * NS, st));
    HIPCHK(hipEventRecord(h->ev0, st));
    // With side streams on, every bin launch goes to a stream the engine created itself (its own
    // stream and the side streams, created together so that the runtime spreads them over different
    // hardware queues); a caller's stream only forks and joins.  Launching a share of the bins on a
    // caller's stream made the overlap depend on which hardware queue that stream happened to be
    // mapped to (T700 on a torch pool stream: 51 ms instead of 32 ms).
    // Lanes for the bin launches: all seven side streams while the step is small (a shard of an 8-rank run, a thinned graph:
    // bins with fewer cells than the GPU has CUs only fill it side by side), two (SE2) / three (SE3) for large steps, whose
    // bins fill the GPU on their own and whose long-chain kernels lose to each other's traffic (C4, 3.2 M cells: 41.6 s with
    // three lanes, 43.6 s with seven; C2 is indifferent).
    const int n_side = h->side_forced || total < 300000 ? h->n_side : std::min(h->n_side, h->dim == 2 ? 2 : 3);
    hipStream_t st0 = n_side ? h->own_stream : st;
    if (n_side) {
        HIPCHK(hipEventRecord(h->ev_fork, st));
        if (st0 != st) HIPCHK(hipStreamWaitEvent(st0, h->ev_fork, 0));
        for (int k = 0; k < n_side; ++k) HIPCHK(hipStreamWaitEvent(h->side[k], h->ev_fork, 0));
    }
    for (int b = nb - 1; b >= 0; --b) {
        for (int nl = 2; nl >= 1; --nl) {
            const int s = (nl == 1 ? 0 : (kMaxBins + 1)) + b;
            if (!counts[s]) continue;
            CellOut out{h->d_chi + offsets[s], h->d_chitot + offsets[s], h->d_meta + offsets[s]};
            const int lane_q = launches % (n_side + 1);
            hipStream_t ls = lane_q == 0 ? st0 : h->side[lane_q - 1];
            const hipError_t e = launch_slot(b, nl, (int)counts[s], h->d_cells + offsets[s], out, ls, h->d_wave_ctr + s);
            if (e != hipSuccess) return fail(IPC_ERR_HIP, "cell kernel launch failed: %s", hipGetErrorString(e));
            ++launches;
        }
    }
    if (st0 != st) {
        HIPCHK(hipEventRecord(h->ev_join_own, st0));
        HIPCHK(hipStreamWaitEvent(st, h->ev_join_own, 0));
    }
    for (int k = 0; k < n_side; ++k) {
        HIPCHK(hipEventRecord(h->ev_join[k], h->side[k]));
        HIPCHK(hipStreamWaitEvent(st, h->ev_join[k], 0));
    }
    if (n_long) {
        if (int rc = solve_long_cells(h, st, nb, counts, offsets)) return rc;
    }
    HIPCHK(hipEventRecord(h->ev1, st));
    h->ev_valid = true;
    h->last_launches = launches;
    h->last_cells = (int)total;
    h->last_long_cells = (int)n_long;
    h->last_lm_cells = h->last_literal_cells = 0;
    const double band = h->term_eps > 0 ? (h->borderline_band >= 0 ? h->borderline_band : 4.0 * std::sqrt(h->term_eps)) : 0.0;
    if (total && (h->lm_retry || band > 0.0)) {
        // The cells to solve again: failed linear solves (Levenberg retry by the host-driven solver: degenerate information,
        // rare) and borderline cells (the literal trial loop, by the cell kernels themselves over compact per-slot lists).
        // ONE read-back of the counts -- the single host wait of a repeated step.
        if ((int)total > h->failed_cap) {
            HIPCHK(hipFree(h->d_failed));
            h->d_failed = nullptr;
            h->failed_cap = std::max(16384, (int)(total + total / 8));
            HIPCHK(hipMalloc(&h->d_failed, sizeof(int) * ((size_t)h->failed_cap + 1)));
        }
        HIPCHK(hipMemsetAsync(h->d_recount, 0, sizeof(int) * (NS + 1), st));
        hipLaunchKernelGGL(k_collect_failed, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (int)total, (const int4*)h->d_meta,
                           (const double*)h->d_chi, (const int2*)h->d_cells, h->prm.fast_reject_th, h->prm.slow_reject_th, band,
                           h->lm_retry, h->failed_cap, h->d_failed, h->d_recount, (const unsigned*)h->d_slot_off, NS, h->d_lit_cells, h->d_lit_idx, nb);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(h->h_recount, h->d_recount, sizeof(int) * (NS + 1), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        int n_lit = 0;
        for (int s = 0; s < NS; ++s) n_lit += h->h_recount[s];
        if (n_lit) {
            P.term_eps = 0.0; P3.term_eps = 0.0;                   // g2o's literal loop (Se2View::term_eps)
            HIPCHK(hipMemsetAsync(h->d_wave_ctr, 0, sizeof(unsigned) * NS, st));
            for (int b = nb - 1; b >= 0; --b) {
                for (int nl = 2; nl >= 1; --nl) {
                    const int s = (nl == 1 ? 0 : (kMaxBins + 1)) + b;
                    const int n = h->h_recount[s];
                    if (!n) continue;
                    CellOut out{h->d_lit_chi + offsets[s], h->d_lit_chitot + offsets[s], h->d_lit_meta + offsets[s]};
                    const hipError_t e = launch_slot(b, nl, n, h->d_lit_cells + offsets[s], out, st, h->d_wave_ctr + s);
                    if (e != hipSuccess) return fail(IPC_ERR_HIP, "cell kernel launch failed: %s", hipGetErrorString(e));
                }
            }
            // (borderline cells beyond every cell kernel were solved by the cluster solver with the engine's term_eps; their
            // slot is not re-solved: counts of the long slots are skipped above because b < nb)
            hipLaunchKernelGGL(k_scatter_literal, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (int)total,
                               (const unsigned*)h->d_slot_off, NS, (const int*)h->d_recount, (const int*)h->d_lit_idx, (const double*)h->d_lit_chi,
                               (const double*)h->d_lit_chitot, (const int4*)h->d_lit_meta, h->d_chi, h->d_chitot, h->d_meta);
            HIPCHK(hipGetLastError());
            h->last_literal_cells += n_lit;
        }
        if (h->h_recount[NS]) {
            if (int rc = resolve_failed_cells(h, h->h_recount[NS])) return rc;
        }
    }
    if (total)
        hipLaunchKernelGGL(k_scatter_bits, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (int)total,
                           h->d_cells, h->d_chi, h->prm.fast_reject_th, h->prm.slow_reject_th, rpr, (const int*)h->d_slot, words,
                           (unsigned long long*)d_upper);
    HIPCHK(hipGetLastError());
    if (total && !cached && phase == 0 && h->slow_first_iterations > 0) {
        // this step built the cell lists: its iteration counts order them for the steps to come (slow cells first)
        const unsigned nblk1 = (unsigned)((total + 1 + 255) / 256), nblk = (unsigned)((total + 255) / 256);
        hipLaunchKernelGGL(k_slow_flags, dim3(nblk1), dim3(256), 0, st, (int)total, (const int4*)h->d_meta, h->slow_first_iterations, h->d_lit_idx);
        hipLaunchKernelGGL(k_scan_int, dim3(1), dim3(1024), 0, st, h->d_lit_idx, (int)total + 1);
        hipLaunchKernelGGL(k_slow_first_permute, dim3(nblk), dim3(256), 0, st, (int)total, (const unsigned*)h->d_slot_off, NS, (const int*)h->d_lit_idx,
                           (const int2*)h->d_cells, (const double*)h->d_chi, (const double*)h->d_chitot, (const int4*)h->d_meta, h->d_lit_cells,
                           h->d_lit_chi, h->d_lit_chitot, h->d_lit_meta);
        HIPCHK(hipGetLastError());
        std::swap(h->d_cells, h->d_lit_cells); std::swap(h->d_chi, h->d_lit_chi);
        std::swap(h->d_chitot, h->d_lit_chitot); std::swap(h->d_meta, h->d_lit_meta);
    }
    return IPC_OK;
}

extern "C" int ipc_assemble_matrix(ipc_engine_t* h, const uint64_t* d_gathered, int world, uint64_t* d_bits,
                                   void* stream)
{
    if (!h || !d_gathered || !d_bits) return fail(IPC_ERR_ARG, "ipc_assemble_matrix: NULL argument");
    if (h->N <= 0) return fail(IPC_ERR_STATE, "ipc_assemble_matrix: no candidates set");
    if (world < 1) return fail(IPC_ERR_ARG, "ipc_assemble_matrix: world %d", world);
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->own_stream;
    const int N = h->N, words = (N + 63) / 64, rpr = ipc_rows_per_rank(N, world);
    if (int rc = matrix_mode_enter(h, st)) return rc;
    if (int rc = ensure_row_map(h, world)) return rc;
    hipLaunchKernelGGL(k_assemble, dim3(words, std::min((N + 255) / 256, 1024)), dim3(256), 0, st, N, words, (const int*)h->d_slot, h->d_lo,
                       h->d_hi, (const unsigned long long*)d_gathered, (unsigned long long*)d_bits);
    HIPCHK(hipGetLastError());
    return IPC_OK;
}

extern "C" int ipc_set_max(ipc_engine_t* h, const uint64_t* d_bits, uint8_t* d_accepted, void* stream)
{
    if (!h || !d_bits || !d_accepted) return fail(IPC_ERR_ARG, "ipc_set_max: NULL argument");
    if (h->N <= 0) return fail(IPC_ERR_STATE, "ipc_set_max: no candidates set");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->own_stream;
    const int N = h->N, words = (N + 63) / 64;
    const size_t shmem = sizeof(unsigned long long) * words;
    if (shmem > 60 * 1024) return fail(IPC_ERR_LIMIT, "ipc_set_max: N=%d exceeds the LDS-resident mask", N);
    if (int rc = matrix_mode_enter(h, st)) return rc;
    hipLaunchKernelGGL(k_set_max, dim3(1), dim3(1024), shmem, st, N, words, h->d_order,
                       (const unsigned long long*)d_bits, d_accepted, h->d_live);
    HIPCHK(hipGetLastError());
    return IPC_OK;
}

extern "C" int ipc_run(ipc_engine_t* h, uint64_t* bits_out, uint8_t* accepted_out)
{
    if (!h) return fail(IPC_ERR_ARG, "ipc_run: NULL handle");
    if (h->N <= 0) return fail(IPC_ERR_STATE, "ipc_run: no candidates set");
    HIPCHK(hipSetDevice(h->device));
    const int N = h->N, words = (N + 63) / 64;
    const size_t need = (size_t)N * words;
    if (need > h->run_cap) {
        hipFree(h->d_upper); hipFree(h->d_bits); hipFree(h->d_acc);
        h->d_upper = h->d_bits = nullptr; h->d_acc = nullptr;
        HIPCHK(hipMalloc(&h->d_upper, sizeof(uint64_t) * need));
        HIPCHK(hipMalloc(&h->d_bits, sizeof(uint64_t) * need));
        HIPCHK(hipMalloc(&h->d_acc, (size_t)N + 64));
        h->run_cap = need;
    }
    int rc = ipc_solve_rows(h, 0, 1, (uint64_t*)h->d_upper, h->own_stream);
    if (rc) return rc;
    rc = ipc_assemble_matrix(h, (const uint64_t*)h->d_upper, 1, (uint64_t*)h->d_bits, h->own_stream);
    if (rc) return rc;
    rc = ipc_set_max(h, (const uint64_t*)h->d_bits, h->d_acc, h->own_stream);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(h->own_stream));
    if (bits_out) HIPCHK(hipMemcpy(bits_out, h->d_bits, sizeof(uint64_t) * need, hipMemcpyDeviceToHost));
    if (accepted_out) HIPCHK(hipMemcpy(accepted_out, h->d_acc, (size_t)N, hipMemcpyDeviceToHost));
    return IPC_OK;
}

extern "C" int ipc_cell_count(ipc_engine_t* h, int* n_cells)
{
    if (!h || !n_cells) return fail(IPC_ERR_ARG, "ipc_cell_count: NULL argument");
    *n_cells = h->last_cells;
    return IPC_OK;
}

extern "C" int ipc_cell_info(ipc_engine_t* h, ipc_cell_info_t* out, int capacity)
{
    if (!h || !out) return fail(IPC_ERR_ARG, "ipc_cell_info: NULL argument");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipDeviceSynchronize());
    const int n = std::min(capacity, h->last_cells);
    if (n <= 0) return IPC_OK;
    std::vector<int2> cells(n);
    std::vector<double> chi(n), tot(n);
    std::vector<int4> meta(n);
    HIPCHK(hipMemcpy(cells.data(), h->d_cells, sizeof(int2) * n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(chi.data(), h->d_chi, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(tot.data(), h->d_chitot, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(meta.data(), h->d_meta, sizeof(int4) * n, hipMemcpyDeviceToHost));
    for (int c = 0; c < n; ++c) {
        ipc_cell_info_t& o = out[c];
        o.i = cells[c].x; o.j = cells[c].y;
        o.lo = std::min(h->h_lo[o.i], h->h_lo[o.j]);
        o.hi = std::max(h->h_hi[o.i], h->h_hi[o.j]);
        o.max_chi2 = chi[c]; o.chi2_total = tot[c];
        o.iterations = meta[c].x; o.tries = meta[c].y; o.flags = meta[c].z; o.evals = meta[c].w;
    }
    return IPC_OK;
}

// counts over the cell records of the last solve: [0] flags & 2 (linear solve failed), [1] not terminated
// (ran to the iteration cap), [2] NaN max chi2
__global__ void k_report(int n, const int4* meta, const double* chi, unsigned* out)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const int4 m = meta[c];
    if (m.z & 2) atomicAdd(&out[0], 1u);
    if (!(m.z & 1) && !(m.z & 2)) atomicAdd(&out[1], 1u);
    if (chi[c] != chi[c]) atomicAdd(&out[2], 1u);
}

extern "C" int ipc_solve_report(ipc_engine_t* h, ipc_solve_report_t* out)
{
    if (!h || !out) return fail(IPC_ERR_ARG, "ipc_solve_report: NULL argument");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipDeviceSynchronize());
    memset(out, 0, sizeof *out);
    out->cells = h->last_cells;
    out->long_cells = h->last_long_cells;
    if (h->last_cells <= 0) return IPC_OK;
    unsigned host[3] = {0, 0, 0};
    HIPCHK(hipMemsetAsync(h->d_counters, 0, sizeof(unsigned) * 3, h->own_stream));
    hipLaunchKernelGGL(k_report, dim3((h->last_cells + 255) / 256), dim3(256), 0, h->own_stream, h->last_cells, h->d_meta,
                       h->d_chi, h->d_counters);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(host, h->d_counters, sizeof host, hipMemcpyDeviceToHost, h->own_stream));
    HIPCHK(hipStreamSynchronize(h->own_stream));
    out->failed_cells = (int)host[0];
    out->capped_cells = (int)host[1];
    out->nan_cells = (int)host[2];
    out->damped_cells = h->last_lm_cells;
    out->literal_cells = h->last_literal_cells;
    return IPC_OK;
}

extern "C" int ipc_solver_time_ms(ipc_engine_t* h, double* ms, int* launches)
{
    if (!h || !ms) return fail(IPC_ERR_ARG, "ipc_solver_time_ms: NULL argument");
    if (!h->ev_valid) return fail(IPC_ERR_STATE, "ipc_solver_time_ms: no solve recorded");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipEventSynchronize(h->ev1));
    float t = 0;
    HIPCHK(hipEventElapsedTime(&t, h->ev0, h->ev1));
    *ms = t;
    if (launches) *launches = h->last_launches;
    return IPC_OK;
}

#if defined(IPC_PHASE_TIMING)
extern "C" int ipc_dbg_read(ipc_engine_t* h, double* out, int n)
{
    Se2View P = make_view(h);
    hipDeviceSynchronize();
    hipMemcpy(out, P.dbg, sizeof(double) * n, hipMemcpyDeviceToHost);
    return 0;
}
#endif
extern "C" int ipc_synchronize(ipc_engine_t* h)
{
    if (!h) return fail(IPC_ERR_ARG, "ipc_synchronize: NULL handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipDeviceSynchronize());
    return IPC_OK;
}

// ------------------------------------------------------------------------------------------
// faithful incremental mode + final map, SE2
// ------------------------------------------------------------------------------------------
__global__ void k_pose5_init(int V, const double* pose0, double* out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    const double th = pose0[2 * (size_t)V + i];
    double s, c;
    sincos_pi(th, s, c);
    out[i] = pose0[i]; out[(size_t)V + i] = pose0[(size_t)V + i]; out[2 * (size_t)V + i] = th;
    out[3 * (size_t)V + i] = c; out[4 * (size_t)V + i] = s;
}

// propagateCurrentGuess (reference src/consensus_utils.cpp:61-71): v[i] = v[i-1] * z[i-1] for
// i = start+1 .. V-1.  Sequential compose, one lane.
__global__ void k_se2_propagate_tail(int V, int start, const double* rec, int stride, double* cur)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double x = cur[start], y = cur[(size_t)V + start], th = cur[2 * (size_t)V + start];
    double c = cur[3 * (size_t)V + start], s = cur[4 * (size_t)V + start];
    for (int i = start + 1; i < V; ++i) {
        const double tx = rec[(size_t)F_TZX * stride + i - 1], ty = rec[(size_t)F_TZY * stride + i - 1];
        x += c * tx - s * ty;
        y += s * tx + c * ty;
        th = normalize_theta(th + rec[(size_t)F_THZ * stride + i - 1]);
        sincos_pi(th, s, c);
        cur[i] = x; cur[(size_t)V + i] = y; cur[2 * (size_t)V + i] = th;
        cur[3 * (size_t)V + i] = c; cur[4 * (size_t)V + i] = s;
    }
}

// SE3 propagateCurrentGuess: v[i] = v[i-1] * z[i-1] (Isometry3 product) for i = start+1 .. V-1
__global__ void k_se3_propagate_tail(int V, int start, const double* rec, int stride, double* cur)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double R[9], t[3];
    for (int q = 0; q < 9; ++q) R[q] = cur[(size_t)q * V + start];
    for (int q = 0; q < 3; ++q) t[q] = cur[(size_t)(9 + q) * V + start];
    for (int i = start + 1; i < V; ++i) {
        double Rz[9], tz[3], Rn[9], d[3];
        for (int q = 0; q < 9; ++q) Rz[q] = rec[(size_t)(G_RZ + q) * stride + i - 1];
        for (int q = 0; q < 3; ++q) tz[q] = rec[(size_t)(G_TZ + q) * stride + i - 1];
        m3_mul(R, Rz, Rn);
        m3_vec(R, tz, d);
        for (int q = 0; q < 3; ++q) t[q] += d[q];
        for (int q = 0; q < 9; ++q) R[q] = Rn[q];
        for (int q = 0; q < 9; ++q) cur[(size_t)q * V + i] = R[q];
        for (int q = 0; q < 3; ++q) cur[(size_t)(9 + q) * V + i] = t[q];
    }
}

// A candidate's OWN chi2 at a pose state ([5 | 12][V]), every candidate of the engine at once: what the speculative
// pipeline predicts verdicts from (spec_pump).  Scheduling only -- no decision ever depends on it.
__global__ void k_cand_own_chi2_se2(const double* poses, int V, const double* cand, int cstride, const int* from, const int* to,
                                    int n, double* out)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const int ia = from[c], ib = to[c];
    const Pose2 a{poses[ia], poses[V + ia], poses[2 * (size_t)V + ia], poses[3 * (size_t)V + ia], poses[4 * (size_t)V + ia]};
    const Pose2 b{poses[ib], poses[V + ib], poses[2 * (size_t)V + ib], poses[3 * (size_t)V + ib], poses[4 * (size_t)V + ib]};
    double e0, e1, e2;
    se2_error(a, b, cand[(size_t)F_TZX * cstride + c], cand[(size_t)F_TZY * cstride + c], cand[(size_t)F_CZ * cstride + c],
              cand[(size_t)F_SZ * cstride + c], cand[(size_t)F_THZ * cstride + c], e0, e1, e2);
    out[c] = gk_sym(cand, cstride, F_OM, c).quad(e0, e1, e2);
}
__global__ void k_cand_own_chi2_se3(const double* poses, int V, const double* cand, int cstride, const int* from, const int* to,
                                    int n, double* out)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    double Rz[9], tz[3], om[21];
    gk3_rz(cand, cstride, c, Rz, tz);
    Edge3 E;
    se3_edge(gk3_pose(poses, V, from[c]), gk3_pose(poses, V, to[c]), Rz, tz, E);
    gk3_sym(cand, cstride, G_OM, c, om);
    out[c] = sym6_quad(om, E.e);
}

// The accept branch of IPC::agreementCheck (src/consensus.cpp:69-71) in one launch: dst = parent outside the window, the
// optimised poses inside lo..hi, then propagateCurrentGuess (consensus_utils.cpp:61-71) down the tail.  The reference
// composes the tail pose by pose, v[i] = v[i-1] (+) z[i-1]; since every pose behind hi is then pure odometry on top of
// v[hi], v[i] = v[hi] (+) (z[hi] ... z[i-1]) = (v[hi] (+) P[hi]^-1) (+) P[i] with P the open-loop poses the engine computed
// once (propagateGuess from the origin, the same chain of compositions): ONE rigid transform applied to the open-loop
// pose, every tail pose in parallel instead of a serial chain of up to V composes on one lane (round 3: 128 us per accept
// on C2's trajectory, on the critical path of everything behind the accept).  The two forms differ by rounding only.
// NF = 5 (SE2: x, y, theta, cos, sin; open = [5][V]) or 12 (SE3: R row-major, t; open = [12][V]); X = the solver's
// arrays [NF][ldx], window index 0 = lo.
template <int NF>
__global__ __launch_bounds__(256) void k_apply_accept(int V, int lo, int hi, const double* parent, const double* X, int ldx,
                                                      const double* open, double* dst)
{
    const int tid = threadIdx.x;
    for (int q = tid; q < NF * V; q += 256) {
        const int f = q / V, i = q - f * V;
        if (i >= lo && i <= hi) dst[q] = X[(size_t)f * ldx + (i - lo)];
        else if (i < lo && dst != parent) dst[q] = parent[q];
    }
    if (hi + 1 >= V) return;
    const size_t Vs = (size_t)V;
    if (NF == 5) {
        // D = T (+) P^-1: theta_D = theta_T - theta_P, t_D = t_T - R_D t_P
        const double thT = X[2 * (size_t)ldx + (hi - lo)], thP = open[2 * Vs + hi];
        const double thD = normalize_theta(thT - thP);
        double sD, cD;
        sincos_pi(thD, sD, cD);
        const double xP = open[hi], yP = open[Vs + hi];
        const double xD = X[hi - lo] - (cD * xP - sD * yP), yD = X[(size_t)ldx + (hi - lo)] - (sD * xP + cD * yP);
        for (int i = hi + 1 + tid; i < V; i += 256) {
            const double x = open[i], y = open[Vs + i];
            const double th = normalize_theta(thD + open[2 * Vs + i]);
            double sn, cs;
            sincos_pi(th, sn, cs);
            dst[i] = xD + (cD * x - sD * y);
            dst[Vs + i] = yD + (sD * x + cD * y);
            dst[2 * Vs + i] = th;
            dst[3 * Vs + i] = cs;
            dst[4 * Vs + i] = sn;
        }
    } else {
        double RT[9], tT[3], RP[9], tP[3], RD[9], tD[3], d[3];
        for (int q = 0; q < 9; ++q) { RT[q] = X[(size_t)q * ldx + (hi - lo)]; RP[q] = open[(size_t)q * Vs + hi]; }
        for (int q = 0; q < 3; ++q) { tT[q] = X[(size_t)(9 + q) * ldx + (hi - lo)]; tP[q] = open[(size_t)(9 + q) * Vs + hi]; }
        for (int r = 0; r < 3; ++r)                             // R_D = R_T R_P^T
            for (int c = 0; c < 3; ++c) RD[3 * r + c] = RT[3 * r] * RP[3 * c] + RT[3 * r + 1] * RP[3 * c + 1] + RT[3 * r + 2] * RP[3 * c + 2];
        m3_vec(RD, tP, d);
        for (int q = 0; q < 3; ++q) tD[q] = tT[q] - d[q];
        for (int i = hi + 1 + tid; i < V; i += 256) {
            double Ri[9], ti[3], Rn[9];
            for (int q = 0; q < 9; ++q) Ri[q] = open[(size_t)q * Vs + i];
            for (int q = 0; q < 3; ++q) ti[q] = open[(size_t)(9 + q) * Vs + i];
            m3_mul(RD, Ri, Rn);
            m3_vec(RD, ti, d);
            for (int q = 0; q < 9; ++q) dst[(size_t)q * Vs + i] = Rn[q];
            for (int q = 0; q < 3; ++q) dst[(size_t)(9 + q) * Vs + i] = tD[q] + d[q];
        }
    }
}

static PoseArr pose_arr(double* base, int V)
{
    return PoseArr{base, base + (size_t)V, base + 2 * (size_t)V, base + 3 * (size_t)V, base + 4 * (size_t)V};
}

static int ensure_incremental(ipc_engine* h, const char* who)
{
    (void)who;
    HIPCHK(hipSetDevice(h->device));
    if (h->dim == 3) {
        if (!h->d_cur) {
            h->d_open = h->d_pose0;                   // same [12][V] layout; not owned twice, see ipc_destroy
            HIPCHK(hipMalloc(&h->d_cur, sizeof(double) * 12 * (size_t)h->V));
            HIPCHK(copy_d2d_now(h, h->d_cur, h->d_open, sizeof(double) * 12 * (size_t)h->V));
        }
        if (!h->cluster3) {
            h->cluster3 = new ClusterSolver3(); h->cluster3->term_eps = h->term_eps; h->cluster3->allow_damping = h->lm_retry;
            h->cluster3->literal_band_min_n = h->literal_band_min_n;
        }
        if (!h->persist3) {
            h->persist3 = new PersistSolver<PersistSe3>(h->knobs); h->persist3->term_eps = h->term_eps; h->persist3->d_prof = h->d_prof;
            if (h->max_helpers >= 0) h->persist3->max_helpers = h->max_helpers;
        }
        return IPC_OK;
    }
    if (!h->d_open) {
        HIPCHK(hipMalloc(&h->d_open, sizeof(double) * 5 * (size_t)h->V));
        HIPCHK(hipMalloc(&h->d_cur, sizeof(double) * 5 * (size_t)h->V));
        hipLaunchKernelGGL(k_pose5_init, dim3((h->V + 255) / 256), dim3(256), 0, h->own_stream, h->V, h->d_pose0, h->d_open);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(h->d_cur, h->d_open, sizeof(double) * 5 * (size_t)h->V, hipMemcpyDeviceToDevice, h->own_stream));
        HIPCHK(hipStreamSynchronize(h->own_stream));
    }
    if (!h->cluster) {
        h->cluster = new ClusterSolver2(); h->cluster->term_eps = h->term_eps; h->cluster->allow_damping = h->lm_retry;
        h->cluster->literal_band_min_n = h->literal_band_min_n;
    }
    if (!h->persist2) {
        h->persist2 = new PersistSolver<PersistSe2>(h->knobs); h->persist2->term_eps = h->term_eps; h->persist2->d_prof = h->d_prof;
        if (h->max_helpers >= 0) h->persist2->max_helpers = h->max_helpers;
    }
    return IPC_OK;
}

static void fill_info(ipc_check_info_t* info, int lo, int hi, int nloops, const ClusterOut& o)
{
    if (!info) return;
    info->lo = lo; info->hi = hi; info->n_cluster_loops = nloops;
    info->iterations = o.iterations; info->tries = o.tries; info->flags = o.flags;
    info->max_chi2 = o.max_chi2; info->chi2_total = o.chi2_total; info->chi2_initial = o.chi2_initial;
}

static int spec_ensure(ipc_engine* h);
extern "C" int ipc_incremental_prepare(ipc_engine_t* h)
{
    if (!h) return fail(IPC_ERR_ARG, "ipc_incremental_prepare: NULL handle");
    if (int rc = ensure_incremental(h, "ipc_incremental_prepare")) return rc;
    if (h->persist && h->spec_window > 1) { if (int rc = spec_ensure(h)) return rc; }
    return IPC_OK;
}

extern "C" int ipc_incremental_reset(ipc_engine_t* h)
{
    if (!h) return fail(IPC_ERR_ARG, "ipc_incremental_reset: NULL handle");
    if (int rc = ensure_incremental(h, "ipc_incremental_reset")) return rc;
    if (int rc = spec_quiesce(h, true)) return rc;
    HIPCHK(copy_d2d_now(h, h->d_cur, h->d_open, sizeof(double) * (h->dim == 2 ? 5 : 12) * (size_t)h->V));
    h->cns.clear(); h->cns_dups = false;
    std::fill(h->handed.begin(), h->handed.end(), 0);
    h->porder = h->order;
    for (int q = 0; q < h->N; ++q) h->ppos[h->porder[q]] = q;
    return IPC_OK;
}

// Resume the faithful loop from a saved state: the g2o vertex estimates and _max_consensus_set are ALL the state
// IPC::agreementCheck reads and writes (include/ipc/consensus.hpp:23-32), so (ipc_current_poses, ipc_consensus_set) taken
// at any point of a run and handed back here continue that run -- on this engine or another one of the same graph.
extern "C" int ipc_incremental_set_state(ipc_engine_t* h, const double* poses, const int* cns, int n_cns, int resume_position)
{
    if (!h || !poses || n_cns < 0 || (n_cns > 0 && !cns)) return fail(IPC_ERR_ARG, "ipc_incremental_set_state: bad argument");
    if (resume_position < 0 || resume_position > h->N) return fail(IPC_ERR_ARG, "ipc_incremental_set_state: resume position %d of %d", resume_position, h->N);
    for (int q = 0; q < n_cns; ++q)
        if (cns[q] < 0 || cns[q] >= h->N) return fail(IPC_ERR_ARG, "ipc_incremental_set_state: member %d of %d candidates", cns[q], h->N);
    if (int rc = ensure_incremental(h, "ipc_incremental_set_state")) return rc;
    if (int rc = spec_quiesce(h, true)) return rc;
    const size_t V = (size_t)h->V;
    const int nf = h->dim == 2 ? 3 : 12;
    std::vector<double> tmp((size_t)nf * V);
    for (size_t i = 0; i < V; ++i)
        for (int f = 0; f < nf; ++f) tmp[(size_t)f * V + i] = poses[(size_t)nf * i + f];
    if (h->dim == 3) {
        HIPCHK(hipMemcpyAsync(h->d_cur, tmp.data(), sizeof(double) * 12 * V, hipMemcpyHostToDevice, h->own_stream));
        HIPCHK(hipStreamSynchronize(h->own_stream));
    } else {
        double* d_tmp = nullptr;                           // [3][V] -> [5][V] (cos, sin as every pose of the engine carries them)
        HIPCHK(hipMalloc(&d_tmp, sizeof(double) * 3 * V));
        HIPCHK(hipMemcpyAsync(d_tmp, tmp.data(), sizeof(double) * 3 * V, hipMemcpyHostToDevice, h->own_stream));
        hipLaunchKernelGGL(k_pose5_init, dim3((h->V + 255) / 256), dim3(256), 0, h->own_stream, h->V, d_tmp, h->d_cur);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(h->own_stream));
        HIPCHK(hipFree(d_tmp));
    }
    h->cns.assign(cns, cns + n_cns);
    h->cns_dups = false;
    {
        std::vector<int> s(h->cns);
        std::sort(s.begin(), s.end());
        h->cns_dups = std::adjacent_find(s.begin(), s.end()) != s.end();
    }
    // the pipeline's prediction of the caller's order: the first resume_position candidates count as handed out
    h->porder = h->order;
    for (int q = 0; q < h->N; ++q) { h->ppos[h->porder[q]] = q; h->handed[h->porder[q]] = q < resume_position ? 1 : 0; }
    return IPC_OK;
}

// computeIndependentSubgraph (src/consensus.cpp:124-171) for candidate k against the current consensus set: absorb
// accepted edges whose id interval overlaps the growing extremes with positive length, until nothing new is found;
// then the threshold / iteration base of :50-52 and the x5 rule of consensus_utils.cpp:12-13.  members = the absorbed
// edges in the order they were found, then k (:56).
struct ClusterSpec { int lo, hi, nclu, iters; double th; std::vector<int> members; };
// The accepted edges (positions into lo / hi, the intervals of the consensus set in acceptance order) that
// computeIndependentSubgraph (src/consensus.cpp:124-171) absorbs for a candidate [klo, khi], and the hull they reach.
//   sweep = false: the reference's fixed point, literally -- re-scan the set until nothing new overlaps the growing hull with
//                  positive length (:140-168); members in the order they are found;
//   sweep = true : the same SET by one pass over the intervals sorted by first vertex.  The fixed point is the connected
//                  component of the candidate in the graph "intervals that overlap with positive length" (an edge overlaps the
//                  hull of a connected set iff it overlaps one of its members: the hull of a connected set has no gaps), and
//                  the re-scans cost one pass per step the hull grows (C5: 130 passes over 5 000 loops per check); members by
//                  first vertex.  tests/test_host_logic.py holds the two against each other on random interval sets.
static void absorbed_edges(int klo, int khi, int n, const int* lo, const int* hi, bool sweep, std::vector<int>& members, int& out_lo, int& out_hi)
{
    members.clear();
    out_lo = klo; out_hi = khi;
    if (sweep) {
        std::vector<int> idx(n + 1);
        for (int q = 0; q <= n; ++q) idx[q] = q;                             // (n = the candidate)
        auto lo_of = [&](int q) { return q == n ? klo : lo[q]; };
        auto hi_of = [&](int q) { return q == n ? khi : hi[q]; };
        std::sort(idx.begin(), idx.end(), [&](int x, int y) { return lo_of(x) != lo_of(y) ? lo_of(x) < lo_of(y) : x < y; });
        int start = 0, reach = hi_of(idx[0]), cs = 0, ce = n + 1;
        bool has_k = idx[0] == n;
        for (int p = 1; p <= n + 1; ++p) {
            if (p == n + 1 || lo_of(idx[p]) >= reach) {                       // component [start, p) ends
                if (has_k) { cs = start; ce = p; break; }
                if (p == n + 1) break;
                start = p; reach = hi_of(idx[p]); has_k = idx[p] == n;
            } else {
                reach = std::max(reach, hi_of(idx[p]));
                has_k = has_k || idx[p] == n;
            }
        }
        for (int p = cs; p < ce; ++p) {
            if (idx[p] == n) continue;
            out_lo = std::min(out_lo, lo[idx[p]]); out_hi = std::max(out_hi, hi[idx[p]]);
            members.push_back(idx[p]);
        }
        return;
    }
    std::vector<char> inc(n, 0);
    bool found = true;
    while (found) {
        found = false;
        for (int q = 0; q < n; ++q) {
            if (inc[q]) continue;
            if (std::min(hi[q], out_hi) - std::max(lo[q], out_lo) <= 0) continue;      // :157-159
            out_lo = std::min(out_lo, lo[q]); out_hi = std::max(out_hi, hi[q]);
            inc[q] = 1; found = true;
            members.push_back(q);
        }
    }
}
// (host code only: both forms of the cluster search on one interval set; members_out holds positions into lo / hi)
extern "C" int ipc_debug_absorbed_edges(int klo, int khi, int n, const int* lo, const int* hi, int sweep, int* members_out, int* n_members_out,
                                        int* lo_out, int* hi_out)
{
    if (n < 0 || (n > 0 && (!lo || !hi)) || !n_members_out || !lo_out || !hi_out) return fail(IPC_ERR_ARG, "ipc_debug_absorbed_edges: bad argument");
    std::vector<int> m;
    absorbed_edges(klo, khi, n, lo, hi, sweep != 0, m, *lo_out, *hi_out);
    *n_members_out = (int)m.size();
    if (members_out) std::copy(m.begin(), m.end(), members_out);
    return IPC_OK;
}
static ClusterSpec cluster_of(const ipc_engine* h, int k, const std::vector<int>& cns)
{
    ClusterSpec c;
    const int n = (int)cns.size();
    std::vector<int> clo(n), chi(n), pos;
    for (int q = 0; q < n; ++q) { clo[q] = h->h_lo[cns[q]]; chi[q] = h->h_hi[cns[q]]; }
    // (sets of 512 and more go through the sweep: their clusters go to the banded solver, which orders its loops itself;
    // smaller ones keep the discovery order of rounds 1 - 4, i.e. their bits)
    absorbed_edges(h->h_lo[k], h->h_hi[k], n, clo.data(), chi.data(), n >= 512, pos, c.lo, c.hi);
    c.members.reserve(pos.size() + 1);
    for (int q : pos) c.members.push_back(cns[q]);
    if (h->cns_dups) {                         // (an edge accepted twice sits in the set twice, :70, and enters the std::set once)
        std::vector<int> uniq;
        for (int e : c.members) if (std::find(uniq.begin(), uniq.end(), e) == uniq.end()) uniq.push_back(e);
        c.members.swap(uniq);
    }
    const bool intersection = !c.members.empty();                             // :50-52
    c.th = intersection ? h->prm.slow_reject_th : h->prm.fast_reject_th;
    c.iters = intersection ? h->prm.slow_reject_iter_base : h->prm.fast_reject_iter_base;
    c.nclu = (int)c.members.size();
    // :56 -- eset_independent is a std::set of edge pointers (:47-56): a candidate that is already in the consensus set (a
    // re-check) was absorbed above and is not inserted a second time
    if (std::find(c.members.begin(), c.members.end(), k) == c.members.end()) c.members.push_back(k);
    if ((c.hi - c.lo) + (int)c.members.size() > 100) c.iters *= 5;            // consensus_utils.cpp:12-13
    return c;
}
static ClusterSpec cluster_of(const ipc_engine* h, int k) { return cluster_of(h, k, h->cns); }

// IPC::agreementCheck's accept branch (src/consensus.cpp:69-71) on the pose buffer `dst` ([5 | 12][V]): the optimised window
// replaces the poses, the tail is re-propagated (propagateCurrentGuess, consensus_utils.cpp:61-71).  Enqueued on `st`.
static int apply_accept(ipc_engine* h, hipStream_t st, double* dst, const double* parent, int lo, int hi, const PoseArr* X2,
                        const double* X3, int ld3)
{
    if (h->dim == 3)
        hipLaunchKernelGGL(k_apply_accept<12>, dim3(1), dim3(256), 0, st, h->V, lo, hi, parent, X3, ld3, (const double*)h->d_open, dst);
    else if (X2->th - X2->y != X2->y - X2->x || X2->c - X2->th != X2->y - X2->x || X2->s - X2->c != X2->y - X2->x)
        return fail(IPC_ERR_STATE, "apply_accept: the solver's pose arrays are not rows of one block");
    else                                            // (x, y, th, c, s: rows of one block, ClusterSolver2 / PersistSe2::carve)
        hipLaunchKernelGGL(k_apply_accept<5>, dim3(1), dim3(256), 0, st, h->V, lo, hi, parent, X2->x, (int)(X2->y - X2->x),
                           (const double*)h->d_open, dst);
    HIPCHK(hipGetLastError());
    return IPC_OK;
}
// ... on the current poses; k joins the set.
static int commit_accept(ipc_engine* h, hipStream_t st, int k, int lo, int hi, const PoseArr* X2, const double* X3, int ld3)
{
    if (int rc = apply_accept(h, st, h->d_cur, h->d_cur, lo, hi, X2, X3, ld3)) return rc;
    if (std::find(h->cns.begin(), h->cns.end(), k) != h->cns.end()) h->cns_dups = true;
    h->cns.push_back(k);
    return IPC_OK;
}

// ---- speculative candidate pipeline (see ipc_engine::SpecSlot) ---------------------------------------------
__global__ void k_nap(unsigned long long ticks)             // (100 MHz wall clock)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// How many of `n` streams really run side by side: streams that share a hardware queue run their kernels one after the
// other, and how many queues the runtime has was decided at ITS initialisation (GPU_MAX_HW_QUEUES), which this library
// may or may not have been in time for.  One 1 ms nap per stream, all at once: the wall time says how many ran abreast
// (tools/microbench/stream_concurrency.cpp: 3.9 with 4 queues, 7.5 with 8, 13.8 of 16 with 16, 19.4 of 24 with 32 -- and a
// collapse to < 1 beyond ~24 streams, whatever the variable says).
static int probe_stream_concurrency(hipStream_t* st, int n)
{
    if (n <= 1) return n;
    const unsigned long long nap = 100000;                 // 1 ms
    for (int q = 0; q < n; ++q) hipLaunchKernelGGL(k_nap, dim3(1), dim3(64), 0, st[q], 100ull);    // (code object loaded, queues created)
    for (int q = 0; q < n; ++q) if (hipStreamSynchronize(st[q]) != hipSuccess) return 1;
    const auto t0 = std::chrono::steady_clock::now();
    for (int q = 0; q < n; ++q) hipLaunchKernelGGL(k_nap, dim3(1), dim3(64), 0, st[q], nap);
    for (int q = 0; q < n; ++q) if (hipStreamSynchronize(st[q]) != hipSuccess) return 1;
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return std::max(1, std::min(n, (int)std::lround(n * 1e-3 / std::max(wall - 0.1e-3, 1e-3))));
}

static int spec_ensure(ipc_engine* h)
{
    if (!h->slots.empty()) return IPC_OK;
    HIPCHK(hipHostMalloc(&h->h_abort, sizeof(int) * 64, hipHostMallocMapped));
    std::memset(h->h_abort, 0, sizeof(int) * 64);
    HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_abort), h->h_abort, 0));
    HIPCHK(hipEventCreateWithFlags(&h->ev_commit, hipEventDisableTiming));
    h->slots.resize(h->spec_window);
    h->spec_active = h->spec_window;
    if (h->max_helpers >= 0) h->helper_limit = std::min(h->helper_limit, h->max_helpers);
    // (workspaces for clusters of up to 256 loops up front, 9 MB (SE2) / 37 MB (SE3) per slot: growing them later means
    // hipFree, which waits for every solve in flight -- 0.36 s of a 2.3 s C2 run when the slots started at 48 loops)
    for (int q = 0; q < h->spec_window; ++q) {
        ipc_engine::SpecSlot& sl = h->slots[q];
        HIPCHK(pipeline_stream(h->device, q, &sl.st));
        HIPCHK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        if (h->dim == 3) {
            sl.s3 = new PersistSolver<PersistSe3>(h->knobs); sl.s3->term_eps = h->term_eps; sl.s3->d_prof = q == 0 ? h->d_prof : nullptr;
            sl.s3->d_abort_word = h->d_abort + q;
            HIPCHK(sl.s3->reserve(h->V - 1, std::max(256, std::min(h->N + 1, 16384)), 256));
        } else {
            sl.s2 = new PersistSolver<PersistSe2>(h->knobs); sl.s2->term_eps = h->term_eps; sl.s2->d_prof = q == 0 ? h->d_prof : nullptr;
            sl.s2->d_abort_word = h->d_abort + q;
            HIPCHK(sl.s2->reserve(h->V - 1, std::max(256, std::min(h->N + 1, 16384)), 256));
        }
    }
    if (!h->window_forced) {
        hipStream_t sts[64];
        const int n = std::min((int)h->slots.size(), 64);
        for (int q = 0; q < n; ++q) sts[q] = h->slots[q].st;
        // Reported with IPC_SPEC_STATS, not acted on: capping the solves in flight at the measured figure was tried and
        // LOSES (C2: 6 abreast measured, 588 candidates/s capped at 6 against 827 with all 10 in flight -- a solve queued
        // behind another on its hardware queue still starts the moment that one ends, without a host round trip)
        h->stream_concurrency = probe_stream_concurrency(sts, n);
        // ... except when the streams evidently share a handful of hardware queues (GPU_MAX_HW_QUEUES reached the environment
        // after HIP had initialised: the runtime's default is 4): an expected accept -- the serial chain -- then sits behind a
        // 13 ms reject on its queue.  C2 / C1 with 4 queues: 338 /s / 0.875 s with 16 slots, 398 /s / 0.700 s with 4.
        if (h->stream_concurrency * 2 <= n) h->spec_active = std::max(2, h->stream_concurrency);
    }
    {
        DevicePipeline& dp = device_pipeline(h->device);
        std::lock_guard<std::recursive_mutex> lk(dp.run_mu);
        if (std::find(dp.engines.begin(), dp.engines.end(), h) == dp.engines.end()) dp.engines.push_back(h);
    }
    // (a caller that never exported GPU_MAX_HW_QUEUES silently runs a quarter of the window -- say so, once)
    if (!getenv("GPU_MAX_HW_QUEUES")) {
        static std::once_flag warned;
        std::call_once(warned, [&] {
            fprintf(stderr, "[ipc_amd] GPU_MAX_HW_QUEUES is not set: the faithful mode keeps %d solves in flight instead of 16 (the HIP runtime's default is "
                            "4 hardware queues; export GPU_MAX_HW_QUEUES=24 before the first HIP call of the process, include/ipc_amd.h \"environment\")\n",
                    h->spec_active);
        });
    }
    return IPC_OK;
}

// a state object that nothing refers to any more (or a new one); its buffer is allocated once and kept
static int spec_alloc_state(ipc_engine* h, int& idx)
{
    idx = -1;
    for (size_t i = 0; i < h->spec_states.size(); ++i) {
        ipc_engine::SpecState& c = h->spec_states[i];
        if (c.live || c.users != 0 || !c.owned) continue;
        // (a dropped state's poses may still be on their way -- copies queued on the stream of the solve that made it:
        // they would land on top of the new owner's)
        if (c.has_ready && hipEventQuery(c.ready) != hipSuccess) continue;
        if (c.has_pred && !c.pred_ready && hipEventQuery(c.pred_ev) != hipSuccess) continue;     // (its prediction kernel still reads them)
        idx = (int)i;
        break;
    }
    if (idx < 0) {
        h->spec_states.emplace_back();
        idx = (int)h->spec_states.size() - 1;
        ipc_engine::SpecState& n = h->spec_states[idx];
        HIPCHK(hipMalloc(&n.d_poses, sizeof(double) * (h->dim == 2 ? 5 : 12) * (size_t)h->V));
        n.owned = true;
        HIPCHK(hipEventCreateWithFlags(&n.ready, hipEventDisableTiming));
    }
    ipc_engine::SpecState& n = h->spec_states[idx];
    n.live = true; n.users = 0; n.pos = -1; n.has_ready = false; n.has_pred = n.pred_ready = false; n.cns.clear();
    return IPC_OK;
}

// the pose state a solve of position p has to start from: the last tentative accept before p, else the committed state
static int spec_state_at(const ipc_engine* h, int p)
{
    int st = h->committed_state;
    for (int t : h->tent) { if (h->spec_states[t].pos < p) st = t; else break; }
    return st;
}

static void spec_abort_slot(ipc_engine* h, int q)
{
    ipc_engine::SpecSlot& sl = h->slots[q];
    if (sl.cand < 0) return;
    __atomic_store_n(&h->h_abort[q], sl.launch_id, __ATOMIC_RELEASE);
    --h->spec_states[sl.state].users;
    sl.cand = -1;
    ++h->spec_wasted;
}

// everything behind position p (solves in flight, parked results, tentative states) assumed that p rejects: forget it
static void spec_drop_after(ipc_engine* h, int p)
{
    for (size_t q = 0; q < h->slots.size(); ++q)
        if (h->slots[q].cand >= 0 && h->slots[q].pos > p) spec_abort_slot(h, (int)q);
    for (size_t r = (size_t)std::max(p + 1, 0); r < h->spec_res.size(); ++r) {
        if (h->spec_res[r].valid) ++h->spec_wasted;
        h->spec_res[r].valid = false;
    }
    while (!h->tent.empty() && h->spec_states[h->tent.back()].pos > p) {
        h->spec_states[h->tent.back()].live = false;
        h->tent.pop_back();
    }
}

// the committed state as an object of the pipeline: the current poses (d_cur) and the current set
static int spec_adopt_current(ipc_engine* h)
{
    if (h->committed_state >= 0) h->spec_states[h->committed_state].live = false;
    int idx = -1;
    for (size_t i = 0; i < h->spec_states.size(); ++i)
        if (!h->spec_states[i].owned) { idx = (int)i; break; }
    if (idx < 0) {
        h->spec_states.emplace_back();
        idx = (int)h->spec_states.size() - 1;
        h->spec_states[idx].d_poses = h->d_cur;
        h->spec_states[idx].owned = false;
    }
    ipc_engine::SpecState& c = h->spec_states[idx];
    c.live = true; c.pos = -1; c.cns = h->cns; c.has_ready = false;      // (its readers wait for ev_commit instead)
    c.has_pred = c.pred_ready = false;
    h->committed_state = idx;
    return IPC_OK;
}

// Every candidate's own chi2 at the poses of state `si`, enqueued on `st` (behind whatever completes those poses), written
// straight into host-mapped memory; S.pred_ev says when they are there.
static int spec_predict_state(ipc_engine* h, int si, hipStream_t st)
{
    ipc_engine::SpecState& S = h->spec_states[si];
    S.has_pred = S.pred_ready = false;
    if (!(h->pred_k > 0.0) || h->N == 0) return IPC_OK;
    if (S.pred_cap < h->N) {                       // (candidates appended since: a kernel of the state's last life may still write the old array)
        if (S.h_pred && S.pred_ev) HIPCHK(hipEventSynchronize(S.pred_ev));
        if (S.h_pred) HIPCHK(hipHostFree(S.h_pred));
        S.h_pred = nullptr;
        S.pred_cap = std::max(1024, 2 * h->N);
        HIPCHK(hipHostMalloc(&S.h_pred, sizeof(double) * S.pred_cap, hipHostMallocMapped));
        HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&S.d_pred), S.h_pred, 0));
    }
    if (!S.pred_ev) HIPCHK(hipEventCreateWithFlags(&S.pred_ev, hipEventDisableTiming));
    if (h->cand_event) HIPCHK(hipStreamWaitEvent(st, h->ev_cand, 0));
    S.pred_n = h->N;
    if (h->dim == 3)
        hipLaunchKernelGGL(k_cand_own_chi2_se3, dim3((h->N + 127) / 128), dim3(128), 0, st, (const double*)S.d_poses, h->V,
                           (const double*)h->d_cand, h->cstride, (const int*)h->d_from, (const int*)h->d_to, h->N, S.d_pred);
    else
        hipLaunchKernelGGL(k_cand_own_chi2_se2, dim3((h->N + 127) / 128), dim3(128), 0, st, (const double*)S.d_poses, h->V,
                           (const double*)h->d_cand, h->cstride, (const int*)h->d_from, (const int*)h->d_to, h->N, S.d_pred);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(S.pred_ev, st));
    S.has_pred = true;
    return IPC_OK;
}

struct SpecTimer {
    double& acc; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit SpecTimer(double& a) : acc(a) {}
    ~SpecTimer() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
static int spec_launch(ipc_engine* h, int q, int p, int helpers, bool expect_reject, int expect, double pred_ratio = -1.0)
{
    SpecTimer tm(h->spec_t_launch);
    ipc_engine::SpecSlot& sl = h->slots[q];
    const int k = h->porder[p], si = spec_state_at(h, p);
    ipc_engine::SpecState& S = h->spec_states[si];
    if (h->cand_event) HIPCHK(hipStreamWaitEvent(sl.st, h->ev_cand, 0));  // (records ipc_append_candidate wrote on own_stream)
    const ClusterSpec c = cluster_of(h, k, S.cns);
    sl.cand = k; sl.pos = p; sl.state = si; sl.lo = c.lo; sl.hi = c.hi; sl.nclu = c.nclu; sl.th = c.th; sl.expect = expect; sl.pred_ratio = pred_ratio;
    // (ids only grow, per slot too: the kernels give up when their slot's word has reached their id, so an abort also
    // reaches a launch that was still queued behind another aborted one when the word moved on)
    sl.launch_id = h->next_launch_id++;
    if (h->next_launch_id >= 0x7ffffff0) return fail(IPC_ERR_LIMIT, "ipc_agreement_check: 2^31 solves launched on one engine");
    if (S.has_ready) HIPCHK(hipStreamWaitEvent(sl.st, S.ready, 0));
    else if (!S.owned && h->commit_count) HIPCHK(hipStreamWaitEvent(sl.st, h->ev_commit, 0));
    ++S.users;
    ++h->spec_launches;
    if (expect_reject) {
        // an expected reject gets a share of the helpers its system could use (one per two 64 x 64 tiles of the first trailing
        // update is what an expected accept gets): 8 of 18 on C2's 480 unknowns, the full 39 on an SE3 cluster of 1 464
        const int n = (h->dim == 2 ? 3 : 6) * (int)c.members.size(), nt = (n + 63) / 64, wanted = (nt * (nt + 1) / 2 + kPSG - 1) / kPSG;
        helpers = std::min(helpers, std::max(h->helper_limit_reject, (int)(0.45 * wanted)));
    }
    if (h->dim == 3) sl.s3->max_helpers = helpers; else sl.s2->max_helpers = helpers;
    if (h->dim == 3) sl.s3->economy = expect_reject; else sl.s2->economy = expect_reject;
    if (h->dim == 3) {
        sl.s3->launch_id = sl.launch_id;
        HIPCHK(sl.s3->launch(sl.st, h->d_chain, h->estride, h->d_cand, h->cstride, S.d_poses, h->V, c.lo, c.hi, c.members,
                             h->h_from.data(), h->h_to.data(), c.iters));
    } else {
        sl.s2->launch_id = sl.launch_id;
        HIPCHK(sl.s2->launch(sl.st, h->d_chain, h->estride, h->d_cand, h->cstride, S.d_poses, h->V, c.lo, c.hi, c.members,
                             h->h_from.data(), h->h_to.data(), c.iters));
    }
    HIPCHK(hipEventRecord(sl.done, sl.st));
    sl.t_launch = std::chrono::steady_clock::now();
    // (a kernel this slot was told to give up may still be running in front of the new one, never beside it)
    sl.busy_wgs = std::max(sl.busy_wgs, h->dim == 3 ? sl.s3->workgroups() : sl.s2->workgroups());
    return IPC_OK;
}

// The accept at position p (result parked, solver buffers of slot q still hold its poses) becomes a tentative state:
// its parent's poses with the optimised window and the re-propagated tail, written on the slot's stream.
static int spec_make_tentative(ipc_engine* h, int p, int q)
{
    SpecTimer tm(h->spec_t_tent);
    ipc_engine::SpecResult& R = h->spec_res[p];
    spec_drop_after(h, p);
    int t = -1;
    if (int rc = spec_alloc_state(h, t)) return rc;
    ipc_engine::SpecSlot& sl = h->slots[q];
    ipc_engine::SpecState& P = h->spec_states[R.state];
    ipc_engine::SpecState& T = h->spec_states[t];
    if (P.has_ready) HIPCHK(hipStreamWaitEvent(sl.st, P.ready, 0));
    else if (!P.owned && h->commit_count) HIPCHK(hipStreamWaitEvent(sl.st, h->ev_commit, 0));
    if (h->dim == 3) {
        const double* res = sl.s3->result_in_second() ? sl.s3->dev().Xn : sl.s3->dev().X;
        if (int rc = apply_accept(h, sl.st, T.d_poses, P.d_poses, R.lo, R.hi, nullptr, res, sl.s3->ld())) return rc;
    } else {
        const PoseArr X = sl.s2->result_in_second() ? sl.s2->dev().Xn : sl.s2->dev().X;
        if (int rc = apply_accept(h, sl.st, T.d_poses, P.d_poses, R.lo, R.hi, &X, nullptr, 0)) return rc;
    }
    HIPCHK(hipEventRecord(T.ready, sl.st));
    T.has_ready = true;
    // (on a stream of its own: the next solve on the new state often lands on this very slot's stream, and the prediction
    // kernel -- 35 small workgroups on a GPU full of persistent ones -- took 2.2 ms on average on C4 in front of it)
    if (!h->pred_stream) HIPCHK(pipeline_stream(h->device, (int)h->slots.size(), &h->pred_stream));
    HIPCHK(hipStreamWaitEvent(h->pred_stream, T.ready, 0));
    if (int rc = spec_predict_state(h, t, h->pred_stream)) return rc;
    T.cns = P.cns;
    if (std::find(T.cns.begin(), T.cns.end(), h->porder[p]) != T.cns.end()) h->cns_dups = true;
    T.cns.push_back(h->porder[p]);
    T.pos = p;
    if (h->spec_log) fprintf(h->spec_log, "tentative,%.0f,%d,%d\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h->spec_log_t0).count(), p, h->porder[p]);
    h->tent.push_back(t);
    R.child = t;
    ++h->spec_tentative;
    return IPC_OK;
}

// moves the candidate at position `from` to position `to` of the prediction; everything keyed by position follows
static void spec_move_position(ipc_engine* h, int from, int to)
{
    if (from == to) return;
    auto remap = [&](int q) {
        if (q == from) return to;
        if (from > to) return (q >= to && q < from) ? q + 1 : q;
        return (q > from && q <= to) ? q - 1 : q;
    };
    if (from > to) {
        std::rotate(h->porder.begin() + to, h->porder.begin() + from, h->porder.begin() + from + 1);
        std::rotate(h->spec_res.begin() + to, h->spec_res.begin() + from, h->spec_res.begin() + from + 1);
    } else {
        std::rotate(h->porder.begin() + from, h->porder.begin() + from + 1, h->porder.begin() + to + 1);
        std::rotate(h->spec_res.begin() + from, h->spec_res.begin() + from + 1, h->spec_res.begin() + to + 1);
    }
    for (int q = std::min(from, to); q <= std::max(from, to); ++q) h->ppos[h->porder[q]] = q;
    for (auto& sl : h->slots) if (sl.cand >= 0) sl.pos = remap(sl.pos);
    for (int t : h->tent) h->spec_states[t].pos = remap(h->spec_states[t].pos);
    std::sort(h->tent.begin(), h->tent.end(), [&](int a, int b) { return h->spec_states[a].pos < h->spec_states[b].pos; });
}

// The caller asks for candidate k, which the prediction has elsewhere than at the head.  k moves to the head.  Every solve
// in flight and every parked result assumed that the candidates in front of it reject (or built on a tentative accept in
// front of it) -- with k in front of them that is still all they assume, so they stay; k's own work survives only if it
// started from the committed state, and what was built on a tentative accept of k goes.
static void spec_move_to_head(ipc_engine* h, int k)
{
    const int p = h->ppos[k], head = h->spec_head;
    if (p < head) {                                    // handed out before (a re-check): back in front of the head, nothing of it is in flight
        spec_move_position(h, p, head - 1);
        h->spec_head = head - 1;
        h->spec_res[h->spec_head] = ipc_engine::SpecResult{};
        return;
    }
    ipc_engine::SpecResult& R = h->spec_res[p];
    const bool from_committed = spec_state_at(h, p) == h->committed_state;
    if (R.valid && R.agree && R.child >= 0) {
        // k's accept is already a tentative state.  From the committed state: it stays and everything else in flight is
        // redone on top of it; otherwise it goes with everything behind it.
        if (from_committed) {
            spec_move_position(h, p, head);
            spec_drop_after(h, head);
            return;
        }
        spec_drop_after(h, p - 1);
    } else if (!from_committed) {
        if (R.valid) { R.valid = false; ++h->spec_wasted; }
        for (size_t q = 0; q < h->slots.size(); ++q)
            if (h->slots[q].cand >= 0 && h->slots[q].pos == p) spec_abort_slot(h, (int)q);
    }
    spec_move_position(h, p, head);
}

// ipc_append_candidate while the pipeline is up: candidate k (the last index) gets a position in the prediction -- behind
// the not yet handed out candidates that end at or before its later vertex (cmpTime) -- and everything behind it shifts
static void spec_insert_position(ipc_engine* h, int k)
{
    const int n = (int)h->porder.size();               // == k
    int q = n;
    if (h->spec_head >= 0) {
        q = std::max(h->spec_head, 0);
        while (q < n && h->h_hi[h->porder[q]] <= h->h_hi[k]) ++q;
    } else {
        while (q > 0 && h->h_hi[h->porder[q - 1]] > h->h_hi[k]) --q;
    }
    h->porder.push_back(k);
    h->ppos.push_back(n);
    h->handed.push_back(0);
    if (h->spec_head >= 0) h->spec_res.emplace_back();
    if (h->spec_head >= 0) spec_move_position(h, n, q);
    else {
        std::rotate(h->porder.begin() + q, h->porder.begin() + n, h->porder.end());
        for (int i = q; i <= n; ++i) h->ppos[h->porder[i]] = i;
    }
}

// One turn of the pipeline: collect the solves that have ended, let the accepts among them (earliest first) move the
// tip, start solves on the free slots.
static int spec_pump(ipc_engine* h)
{
    const int B = std::min((int)h->slots.size(), h->spec_active > 0 ? h->spec_active : (int)h->slots.size());
    int fin_pos[64], fin_slot[64], nfin = 0;
    for (int q = 0; q < B; ++q) {
        ipc_engine::SpecSlot& sl = h->slots[q];
        if (sl.cand < 0 && sl.busy_wgs == 0) continue;
        const hipError_t e = hipEventQuery(sl.done);
        if (e == hipErrorNotReady) continue;
        HIPCHK(e);
        sl.busy_wgs = 0;
        if (sl.cand < 0) continue;                                            // (a solve that was told to give up has left the GPU)
        ClusterOut o;
        HIPCHK(h->dim == 3 ? sl.s3->fetch(o) : sl.s2->fetch(o));
        const bool aborted = h->dim == 3 ? sl.s3->aborted() : sl.s2->aborted();
        const bool lost = h->dim == 3 ? sl.s3->timed_out() : sl.s2->timed_out();
        const int p = sl.pos;
        --h->spec_states[sl.state].users;
        const bool stale = aborted || p < h->spec_head || sl.state != spec_state_at(h, p);
        if (h->spec_log) {
            const auto us = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::micro>(t - h->spec_log_t0).count(); };
            fprintf(h->spec_log, "solve,%.0f,%.0f,%d,%d,%d,%d,%d,%d,%d,%d,%.0f,%.4g\n", us(sl.t_launch), us(std::chrono::steady_clock::now()), p, sl.cand, sl.expect,
                    h->spec_states[sl.state].pos, stale ? (aborted ? 2 : 1) : 0, !(o.max_chi2 > sl.th) ? 1 : 0, o.iterations, sl.nclu,
                    h->dim == 3 ? sl.s3->device_us() : sl.s2->device_us(), sl.pred_ratio);
        }
        sl.cand = -1;
        if (stale) { ++h->spec_wasted; continue; }
        ipc_engine::SpecResult& R = h->spec_res[p];
        R = ipc_engine::SpecResult{};
        R.valid = true; R.state = sl.state; R.lo = sl.lo; R.hi = sl.hi; R.nclu = sl.nclu; R.o = o;
        if (lost) ++h->persist_timeouts;
        R.retry_host = lost || ((o.flags & 2) && h->lm_retry);                // (decided when its turn comes: launched again alone, cluster_solve)
        R.agree = !R.retry_host && !(o.max_chi2 > sl.th);                     // consensus_utils.cpp:17-21
        ++h->pred_conf[sl.expect + 1][R.agree ? 1 : 0];
        {
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - sl.t_launch).count();
            const double dev = 1e-6 * (h->dim == 3 ? sl.s3->device_us() : sl.s2->device_us());
            if (R.agree) { h->st_acc_s += dt; h->st_acc_dev_s += dev; h->st_acc_it += o.iterations; ++h->st_acc_n; }
            else { h->st_rej_s += dt; h->st_rej_dev_s += dev; h->st_rej_it += o.iterations; ++h->st_rej_n; }
        }
        fin_pos[nfin] = p; fin_slot[nfin] = q; ++nfin;
    }
    for (int a = 0; a < nfin; ++a)                                            // by position (a handful at most)
        for (int b = a + 1; b < nfin; ++b)
            if (fin_pos[b] < fin_pos[a]) { std::swap(fin_pos[a], fin_pos[b]); std::swap(fin_slot[a], fin_slot[b]); }
    for (int a = 0; a < nfin; ++a) {
        const ipc_engine::SpecResult& R = h->spec_res[fin_pos[a]];
        if (!R.valid || !R.agree) continue;                                   // (dropped by an earlier accept of this turn)
        if (int rc = spec_make_tentative(h, fin_pos[a], fin_slot[a])) return rc;
    }
    // How many solves to keep in flight: the j-th one beyond the first unknown verdict is of use only if the j before it
    // all reject -- (1 - accept rate)^j; below 5 % it is not started (3 in flight at 70 % accepts, the whole window at 13 %).
    const bool file_pred = !h->pred_accept.empty();
    const bool predicting = file_pred || h->pred_k > 0.0;
    const double pred_th = h->pred_k * h->prm.slow_reject_th;
    // the predictions that hold at a position: those of the state a solve there starts from, or of the nearest state in
    // front of it whose predictions have arrived (a candidate's own chi2 moves little from one state to the next)
    auto preds_of = [&](int si, int& n) -> const double* {
        ipc_engine::SpecState& S = h->spec_states[si];
        if (!S.has_pred) return nullptr;
        if (!S.pred_ready) { if (hipEventQuery(S.pred_ev) != hipSuccess) return nullptr; S.pred_ready = true; }
        n = S.pred_n;
        return S.h_pred;
    };
    // the solves of the OTHER engines of this device that are still on the GPU (their owners are not pumping while this call
    // holds the device's lock; a finished solve they have not collected yet no longer counts)
    int foreign_busy = 0, foreign_xcd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (ipc_engine* o : device_pipeline(h->device).engines) {
        if (o == h) continue;
        for (ipc_engine::SpecSlot& sl : o->slots) {
            if (sl.busy_wgs == 0) continue;
            // (a finished solve is asked about ONCE: its owner still collects it by sl.cand -- with dozens of idle engines alive, as
            // in the test suite's process, sixteen event queries per engine and pump made the host loop the bottleneck: 9.3 s for C2)
            if (hipEventQuery(sl.done) != hipErrorNotReady) { sl.busy_wgs = 0; continue; }
            foreign_busy += sl.busy_wgs;
            for (int x = 0; x < 8; ++x) if (sl.busy_wgs > x) foreign_xcd[x] += (sl.busy_wgs - x + 7) / 8;
        }
    }
    int target = B;
    if (!predicting && h->accept_rate > 0.01) {
        const double r = std::log(0.05) / std::log(std::max(1e-9, 1.0 - std::min(h->accept_rate, 0.999)));
        target = std::max(2, std::min(target, 1 + (int)r));
    }
    const double release_ms = h->gate_release_ms > 0.0 ? h->gate_release_ms
                              : (h->st_acc_n >= 8 ? std::max(1.0, 2.5e3 * h->st_acc_s / (double)h->st_acc_n) : 8.0);
    const auto now = std::chrono::steady_clock::now();
    // the positions to launch: lowest first, the ones from the head on that have neither a parked result nor a solve in flight
    const int end = std::min(h->N, h->spec_head + h->spec_ahead);
    int why = 1;                                       // why the scan ended (idle_why)
    // Two scans.  The first starts nothing but the next expected accept in line, if none is running: it is the serial chain
    // of the run, and each of the (on C2: seven) expected rejects in front of it costs 20 - 30 us of this thread to launch.
    for (int pa_only = predicting ? 1 : 0; pa_only >= 0; --pa_only) {
    why = 1;
    int cur_n = 0;
    const double* cur_pred = file_pred || !predicting ? nullptr : preds_of(h->committed_state, cur_n);
    size_t ti = 0;
    bool gated = false;                                // a predicted accept with its verdict still out lies in front of lp
    int gate_hi = -1;                                  // its later vertex: the candidates that END there too may be asked for before it (cmpTime
                                                       // leaves their order open, src/utils.cpp:379-389), so they are not "behind" it
    int behind = 0;                                    // solves in flight behind it
    int chain_out = 0;                                 // (first scan) expected accepts in flight, gate and hedges
    int toss_out = 0;                                  // (first scan) of them the coin tosses (uncertain_from)
    const int hedge = h->spec_hedge >= 0 ? h->spec_hedge : (h->st_acc_n >= 8 && h->st_acc_s >= 3e-3 * (double)h->st_acc_n ? 1 : 0);
    for (int lp = h->spec_head; lp < end; ++lp) {
        if (h->spec_res[lp].valid) continue;
        int q = -1, running = 0, busy = 0, at = -1;
        for (int i = 0; i < B; ++i) {
            running += h->slots[i].cand >= 0;
            if (h->slots[i].cand >= 0 && h->slots[i].pos == lp) at = i;
            if (h->slots[i].cand < 0 && (q < 0 || (h->slots[q].busy_wgs && !h->slots[i].busy_wgs))) q = i;   // (an empty stream first)
        }
        while (!file_pred && predicting && ti < h->tent.size() && h->spec_states[h->tent[ti]].pos < lp) {
            int n2 = 0;
            if (const double* p2 = preds_of(h->tent[ti], n2)) { cur_pred = p2; cur_n = n2; }
            ++ti;
        }
        const int cand_lp = h->porder[lp];
        const bool pa = file_pred ? (cand_lp < (int)h->pred_accept.size() && h->pred_accept[cand_lp] != 0)
                                   : (cur_pred && cand_lp < cur_n && cur_pred[cand_lp] <= pred_th);
        const bool tied = gated && h->h_hi[cand_lp] == gate_hi;
        if (at >= 0) {
            if (gated) behind += !tied;
            else if (pa && std::chrono::duration<double, std::milli>(now - h->slots[at].t_launch).count() < release_ms) { gated = true; gate_hi = h->h_hi[cand_lp]; }
            if (pa_only && gated && pa) {                // (the chain is running)
                if (h->uncertain_from > 0.0 && h->slots[at].pred_ratio > h->uncertain_from) { if (++toss_out > 2) break; }
                else if (++chain_out > hedge) break;
            }
            continue;
        }
        if (pa_only && !pa) continue;
        if (q < 0 || running >= target) { why = -1; break; }
        if (gated && !tied && behind >= h->spec_behind && !(pa_only && (hedge > 0 || toss_out > 0))) { why = 0; break; }     // (a hedge is not one of the `behind`)
        // every workgroup of every solve on the GPU must be resident (they meet at grid barriers) and one workgroup fills
        // a CU's register file: the workgroups in flight may not exceed the CUs -- less a few, so that the copies and the
        // tail propagation of an accept (on the critical path of everything behind it) never wait for a solve to end
        for (int i = 0; i < B; ++i) if (i != q) busy += h->slots[i].busy_wgs;
        busy += foreign_busy;
        // (an expected reject with next to nothing beside it -- a caller that appends one candidate per check -- is the critical path too)
        const bool expect_reject = (cur_pred || file_pred) && !pa && running >= 4;
        // ... and per XCD: workgroup b of a launch goes to XCD b % 8 (observed placement, MI355X guide: used for speed only -- a wrong
        // guess costs time, never a result), so a launch of 18 puts 3 on XCDs 0 and 1 and 2 on the others, and sixteen of them ask
        // XCD 0 for 48 CUs of its 32: the last workgroups of a launch then wait for a CU while the others spin at the first
        // barrier (IPC_PERSIST_PROF "rest": a quarter of the leader's time on C4 with the global count alone)
        int gmax = 1 << 30;
        {
            const int per_xcd = h->xcd_cus > 0 ? h->xcd_cus : h->n_cu / 8 - 1;
            for (int x = 0; x < 8; ++x) {
                int load = 0;
                for (int i = 0; i < B; ++i) if (i != q && h->slots[i].busy_wgs > x) load += (h->slots[i].busy_wgs - x + 7) / 8;
                load += foreign_xcd[x];
                gmax = std::min(gmax, 8 * std::max(0, per_xcd - load) + x);
            }
        }
        const int helpers = std::min(std::min(h->helper_limit, gmax - 1), h->n_cu - 8 - busy - 1);
        if (helpers < std::min(8, h->helper_limit) && running > 0) { why = 2; break; }        // (wait for a solve to leave)
        const int tip = spec_state_at(h, lp);
        if (!PersistSolver<PersistSe2>::fits(h->V, (int)h->spec_states[tip].cns.size() + 1)) { why = 3; break; }
        if (int rc = spec_launch(h, q, lp, std::max(0, helpers), expect_reject, (cur_pred || file_pred) ? (pa ? 1 : 0) : -1,
                                  cur_pred && cand_lp < cur_n ? cur_pred[cand_lp] / h->prm.slow_reject_th : -1.0)) return rc;
        if (gated) behind += !tied;
        else if (pa) { gated = true; gate_hi = h->h_hi[cand_lp]; }
        if (pa_only) break;
    }
    }
    {
        int idle = 0;
        for (int i = 0; i < B; ++i) idle += h->slots[i].cand < 0;
        h->idle_why[4] += (unsigned long long)B;
        if (idle > 0 && why >= 0) h->idle_why[why] += (unsigned long long)idle;
    }
    return IPC_OK;
}

// stop everything in flight and forget every result and tentative state (the poses / the set change from outside, or the
// caller leaves the processing order)
static int spec_reset(ipc_engine* h)
{
    for (size_t q = 0; q < h->slots.size(); ++q) spec_abort_slot(h, (int)q);
    for (auto& sl : h->slots) { HIPCHK(hipStreamSynchronize(sl.st)); sl.busy_wgs = 0; }      // (nothing of this engine is on the GPU any more)
    if (h->pred_stream) HIPCHK(hipStreamSynchronize(h->pred_stream));      // (the prediction kernels read the candidates and the states' poses)
    for (auto& R : h->spec_res) R.valid = false;
    for (int t : h->tent) h->spec_states[t].live = false;
    h->tent.clear();
    if (h->commit_count) HIPCHK(hipEventSynchronize(h->ev_commit));
    h->spec_head = -1;
    return IPC_OK;
}

static int agreement_check_speculative(ipc_engine* h, int k, int* agrees, ipc_check_info_t* info)
{
    DevicePipeline& dp = device_pipeline(h->device);
    std::lock_guard<std::recursive_mutex> run_lk(dp.run_mu);
    if (int rc = spec_ensure(h)) return rc;
    dp.active = h;                                      // (other engines of this device keep their look-ahead: spec_pump's budget counts their solves)
    SpecTimer tm(h->spec_t_total);
    if (h->spec_head < 0) {                            // first call (or after a reset): the pipeline starts from the poses as they are
        if (int rc = spec_reset(h)) return rc;
        if (int rc = spec_adopt_current(h)) return rc;
        if (int rc = spec_predict_state(h, h->committed_state, h->own_stream)) return rc;     // (own_stream: behind the last commit)
        if (h->pred_k > 0.0) HIPCHK(hipStreamSynchronize(h->own_stream));
        h->spec_res.assign(h->N, ipc_engine::SpecResult{});
        if (const char* pf = getenv("IPC_SPEC_PREDICT_FILE")) {          // (experiments: the verdicts of a recorded run as the prediction)
            if (FILE* f = *pf ? fopen(pf, "r") : nullptr) {
                h->pred_accept.assign(h->N, 0);
                for (int c = 0; c < h->N; ++c) { const int ch = fgetc(f); if (ch == EOF) break; h->pred_accept[c] = ch == '1'; }
                fclose(f);
            }
        }
        // the candidates already handed out in front of the head, the others behind it, both in the order they have
        std::stable_partition(h->porder.begin(), h->porder.end(), [&](int c) { return h->handed[c] != 0; });
        h->spec_head = 0;
        for (int q = 0; q < h->N; ++q) { h->ppos[h->porder[q]] = q; h->spec_head += h->handed[h->porder[q]] != 0; }
    }
    if (h->ppos[k] != h->spec_head) spec_move_to_head(h, k);          // a caller off the predicted order
    const int p = h->spec_head;
    for (unsigned spin = 0;; ++spin) {
        if (int rc = spec_pump(h)) return rc;
        if (h->spec_res[p].valid) break;
        bool running = false;
        for (auto& sl : h->slots) running = running || sl.cand >= 0;
        if (!running) return fail(IPC_ERR_STATE, "ipc_agreement_check: the pipeline lost candidate %d", k);
        if (spin > 64) std::this_thread::yield();
    }
    ipc_engine::SpecResult R = h->spec_res[p];
    h->spec_res[p].valid = false;
    ++h->spec_hits;
    if (h->spec_log) fprintf(h->spec_log, "verdict,%.0f,%d,%d,%d\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h->spec_log_t0).count(), p, k, R.agree ? 1 : 0);
    ClusterOut o = R.o;
    const double th = R.nclu ? h->prm.slow_reject_th : h->prm.fast_reject_th;
    bool agree = R.agree;
    if (R.retry_host) {
        // the capacitance factorisation met a non-positive pivot: redo this check with the host-driven solver, which
        // retries with Levenberg damping as g2o does (rare: degenerate information matrices, NaN poses)
        if (int rc = spec_reset(h)) return rc;
        const ClusterSpec c = cluster_of(h, k);
        HIPCHK(cluster_solve(h, h->d_chain, h->d_cur, c.lo, c.hi, c.members, c.iters, o));
        agree = !(o.max_chi2 > th);
        if (agree) {
            int rld = 0;
            if (h->dim == 3) {
                const double* res = cluster_result3(h, rld);
                if (int rc = commit_accept(h, h->own_stream, k, R.lo, R.hi, nullptr, res, rld)) return rc;
            } else {
                const PoseArr X = cluster_result2(h);
                if (int rc = commit_accept(h, h->own_stream, k, R.lo, R.hi, &X, nullptr, 0)) return rc;
            }
            HIPCHK(hipEventRecord(h->ev_commit, h->own_stream));
            ++h->commit_count;
        }
        h->handed[k] = 1;
        // (spec_head is -1: the next call starts the pipeline again from the poses as they are now)
    } else {
        if (agree) {                                                          // :69-71 -- the tentative state becomes THE state
            if (h->tent.empty() || h->tent.front() != R.child || R.child < 0)
                return fail(IPC_ERR_STATE, "ipc_agreement_check: accept of candidate %d without its state", k);
            ipc_engine::SpecState& T = h->spec_states[R.child];
            h->spec_states[h->committed_state].live = false;
            h->committed_state = R.child;
            h->tent.erase(h->tent.begin());
            h->cns = T.cns;
            // the poses everyone outside the pipeline reads
            HIPCHK(hipStreamWaitEvent(h->own_stream, T.ready, 0));
            HIPCHK(hipMemcpyAsync(h->d_cur, T.d_poses, sizeof(double) * (h->dim == 2 ? 5 : 12) * (size_t)h->V, hipMemcpyDeviceToDevice,
                                  h->own_stream));
            HIPCHK(hipEventRecord(h->ev_commit, h->own_stream));
            ++h->commit_count;
            ++h->spec_promoted;
        }
        h->spec_head = p + 1;
        h->handed[k] = 1;
        h->accept_rate += 0.08 * ((agree ? 1.0 : 0.0) - h->accept_rate);
        if (h->spec_head >= h->N) { if (int rc = spec_reset(h)) return rc; }
    }
    *agrees = agree ? 1 : 0;
    fill_info(info, R.lo, R.hi, R.nclu, o);
    return IPC_OK;
}

// the poses / the set are about to be read or edited from outside the pipeline: nothing in flight may outlive that
static int spec_quiesce(ipc_engine* h, bool state_changes)
{
    std::lock_guard<std::recursive_mutex> run_lk(device_pipeline(h->device).run_mu);
    if (h->slots.empty()) return IPC_OK;
    if (state_changes) return spec_reset(h);
    if (h->commit_count) HIPCHK(hipEventSynchronize(h->ev_commit));
    return IPC_OK;
}

extern "C" int ipc_agreement_check(ipc_engine_t* h, int k, int* agrees, ipc_check_info_t* info)
{
    if (!h || !agrees) return fail(IPC_ERR_ARG, "ipc_agreement_check: NULL argument");
    if (h->N == 0) return fail(IPC_ERR_STATE, "ipc_agreement_check: no candidates set");
    if (k < 0 || k >= h->N) return fail(IPC_ERR_ARG, "ipc_agreement_check: candidate %d of %d", k, h->N);
    if (int rc = ensure_incremental(h, "ipc_agreement_check")) return rc;
    if (h->persist && h->spec_window > 1 && PersistSolver<PersistSe2>::fits(h->V, (int)h->cns.size() + 1))
        return agreement_check_speculative(h, k, agrees, info);
    if (int rc = spec_quiesce(h, true)) return rc;
    const ClusterSpec c = cluster_of(h, k);
    ClusterOut o;
    HIPCHK(cluster_solve(h, h->d_chain, h->d_cur, c.lo, c.hi, c.members, c.iters, o));
    const bool agree = !(o.max_chi2 > c.th);                                  // consensus_utils.cpp:17-21
    if (agree) {                                                              // :69-71
        int rld = 0;
        if (h->dim == 3) {
            const double* res = cluster_result3(h, rld);
            if (int rc = commit_accept(h, h->own_stream, k, c.lo, c.hi, nullptr, res, rld)) return rc;
        } else {
            const PoseArr X = cluster_result2(h);
            if (int rc = commit_accept(h, h->own_stream, k, c.lo, c.hi, &X, nullptr, 0)) return rc;
        }
        HIPCHK(hipStreamSynchronize(h->own_stream));
    }
    h->handed[k] = 1;
    *agrees = agree ? 1 : 0;
    fill_info(info, c.lo, c.hi, c.nclu, o);
    return IPC_OK;
}

extern "C" int ipc_incremental_counters(ipc_engine_t* h, ipc_incremental_counters_t* out)
{
    if (!h || !out) return fail(IPC_ERR_ARG, "ipc_incremental_counters: NULL argument");
    out->host_solver_fallbacks = h->lm_fallbacks;
    out->lost_launches = h->persist_timeouts;
    out->relaunches = h->persist_relaunches;
    out->literal_band_solves = (h->cluster ? h->cluster->literal_band_solves() : 0) + (h->cluster3 ? h->cluster3->literal_band_solves() : 0);
    return IPC_OK;
}

extern "C" int ipc_consensus_size(ipc_engine_t* h, int* n)
{
    if (!h || !n) return fail(IPC_ERR_ARG, "ipc_consensus_size: NULL argument");
    *n = (int)h->cns.size();
    return IPC_OK;
}

extern "C" int ipc_consensus_set(ipc_engine_t* h, int* out)
{
    if (!h || (!out && !h->cns.empty())) return fail(IPC_ERR_ARG, "ipc_consensus_set: NULL argument");
    std::copy(h->cns.begin(), h->cns.end(), out);
    return IPC_OK;
}

extern "C" int ipc_remove_from_consensus(ipc_engine_t* h, int k, int* removed)
{
    if (!h || !removed) return fail(IPC_ERR_ARG, "ipc_remove_from_consensus: NULL argument");
    if (k < 0 || k >= h->N) return fail(IPC_ERR_ARG, "ipc_remove_from_consensus: candidate %d of %d", k, h->N);
    if (int rc = spec_quiesce(h, true)) return rc;
    *removed = 0;
    for (auto it = h->cns.begin(); it != h->cns.end(); ++it) {
        if (h->h_lo[*it] != h->h_lo[k] || h->h_hi[*it] != h->h_hi[k]) continue;
        h->cns.erase(it);
        *removed = 1;
        break;
    }
    return IPC_OK;
}

extern "C" int ipc_add_to_consensus(ipc_engine_t* h, int k)
{
    if (!h) return fail(IPC_ERR_ARG, "ipc_add_to_consensus: NULL handle");
    if (k < 0 || k >= h->N) return fail(IPC_ERR_ARG, "ipc_add_to_consensus: candidate %d of %d", k, h->N);
    if (int rc = spec_quiesce(h, true)) return rc;
    for (int e : h->cns)
        if (h->h_lo[e] == h->h_lo[k] && h->h_hi[e] == h->h_hi[k]) return IPC_OK;
    h->cns.push_back(k);
    // cmpEdgesTime (src/utils.cpp:371-377); std::sort leaves ties unspecified, the build keeps them stable
    std::stable_sort(h->cns.begin(), h->cns.end(), [&](int a, int b) { return h->h_hi[a] < h->h_hi[b]; });
    return IPC_OK;
}

static int download_poses(ipc_engine* h, const PoseArr& X, int n, double* poses_out)
{
    std::vector<double> tmp(3 * (size_t)n);
    HIPCHK(hipMemcpy(tmp.data(), X.x, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(tmp.data() + n, X.y, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(tmp.data() + 2 * (size_t)n, X.th, sizeof(double) * n, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) {
        poses_out[3 * (size_t)i] = tmp[i]; poses_out[3 * (size_t)i + 1] = tmp[(size_t)n + i];
        poses_out[3 * (size_t)i + 2] = tmp[2 * (size_t)n + i];
    }
    return IPC_OK;
}

// SE3: device [12][ld] (first n columns) -> host [n][12]
static int download_poses3(ipc_engine* h, const double* d, int ld, int n, double* poses_out)
{
    std::vector<double> tmp(12 * (size_t)n);
    HIPCHK(hipMemcpy2D(tmp.data(), sizeof(double) * n, d, sizeof(double) * ld, sizeof(double) * n, 12, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i)
        for (int f = 0; f < 12; ++f) poses_out[12 * (size_t)i + f] = tmp[(size_t)f * n + i];
    return IPC_OK;
}

extern "C" int ipc_current_poses(ipc_engine_t* h, double* poses_out)
{
    if (!h || !poses_out) return fail(IPC_ERR_ARG, "ipc_current_poses: NULL argument");
    if (int rc = ensure_incremental(h, "ipc_current_poses")) return rc;
    if (int rc = spec_quiesce(h, false)) return rc;
    if (h->dim == 3) return download_poses3(h, h->d_cur, h->V, h->V, poses_out);
    return download_poses(h, pose_arr(h->d_cur, h->V), h->V, poses_out);
}

extern "C" int ipc_final_optimize(ipc_engine_t* h, const uint8_t* accepted, int iterations, double* poses_out,
                                  ipc_check_info_t* info)
{
    if (!h || (h->N > 0 && !accepted)) return fail(IPC_ERR_ARG, "ipc_final_optimize: NULL argument");
    if (iterations < 0) return fail(IPC_ERR_ARG, "ipc_final_optimize: iterations %d", iterations);
    if (int rc = ensure_incremental(h, "ipc_final_optimize")) return rc;
    const int E = h->V - 1;
    if (!h->d_chain1) {
        double *d_m = nullptr, *d_i = nullptr;
        const int ms = h->dim == 2 ? 3 : 7, is = h->dim == 2 ? 6 : 21;
        const size_t nf = h->dim == 2 ? (size_t)F_NFIELDS : (size_t)G_NFIELDS;
        HIPCHK(hipMalloc(&h->d_chain1, sizeof(double) * (nf * h->estride + 64)));
        HIPCHK(hipMalloc(&d_m, sizeof(double) * ms * E));
        HIPCHK(hipMalloc(&d_i, sizeof(double) * is * E));
        HIPCHK(hipMemcpy(d_m, h->h_odom_meas.data(), sizeof(double) * ms * E, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_i, h->h_odom_info.data(), sizeof(double) * is * E, hipMemcpyHostToDevice));
        HIPCHK(hipStreamSynchronize(nullptr));                     // (NULL-stream copies are not ordered against the engine's non-blocking streams: copy_d2d_now)
        HIPCHK(hipMemsetAsync(h->d_chain1, 0, sizeof(double) * (nf * h->estride + 64), h->own_stream));
        if (h->dim == 2)
            hipLaunchKernelGGL(k_se2_prep, dim3((E + 255) / 256), dim3(256), 0, h->own_stream, E, d_m, d_i,
                               h->prm.s_factor, h->d_chain1, h->estride, h->prm.s_factor);
        else
            hipLaunchKernelGGL(k_se3_prep, dim3((E + 63) / 64), dim3(64), 0, h->own_stream, E, d_m, d_i,
                               h->prm.s_factor, h->d_chain1, h->estride, h->prm.s_factor);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(h->own_stream));
        HIPCHK(hipFree(d_m));
        HIPCHK(hipFree(d_i));
    }
    std::vector<int> members;
    for (int k : h->order) if (accepted[k]) members.push_back(k);
    if (members.empty()) {
        // pure odometry: the open-loop guess already has zero error, optimize() leaves it alone
        ClusterOut o;
        fill_info(info, 0, h->V - 1, 0, o);
        if (poses_out && h->dim == 3) return download_poses3(h, h->d_open, h->V, h->V, poses_out);
        if (poses_out) return download_poses(h, pose_arr(h->d_open, h->V), h->V, poses_out);
        return IPC_OK;
    }
    ClusterOut o;
    if (h->dim == 3) {
        HIPCHK(cluster_solve(h, h->d_chain1, h->d_open, 0, h->V - 1, members, iterations, o));
        fill_info(info, 0, h->V - 1, (int)members.size(), o);
        int rld = 0;
        const double* res = cluster_result3(h, rld);
        if (poses_out) return download_poses3(h, res, rld, h->V, poses_out);
        return IPC_OK;
    }
    HIPCHK(cluster_solve(h, h->d_chain1, h->d_open, 0, h->V - 1, members, iterations, o));
    fill_info(info, 0, h->V - 1, (int)members.size(), o);
    if (poses_out) return download_poses(h, cluster_result2(h), h->V, poses_out);
    return IPC_OK;
}

// Diagnostic: solve the dense SPD system of a cluster's capacitance matrix as the incremental mode does.
// system: (n+1) x n column major, lower triangle of S in rows 0..n-1, right-hand side in row n.
// mode 0: dense_chol.hpp (one launch per block column), mode 1: the persistent orchestration with `workgroups`.
extern "C" int ipc_debug_dense_solve(int n, const double* system, int mode, int workgroups, double* x_out, int* info_out)
{
    if (n < 1 || !system || !x_out || !info_out) return fail(IPC_ERR_ARG, "ipc_debug_dense_solve: bad argument");
    const size_t m = (size_t)(n + 1) * n;
    double *dA = nullptr, *dx = nullptr, *ddinv = nullptr;
    int* dinfo = nullptr;
    PersistCtl* dctl = nullptr;
    HIPCHK(hipMalloc(&dA, sizeof(double) * 2 * m));
    HIPCHK(hipMalloc(&dx, sizeof(double) * (n + 64)));
    HIPCHK(hipMalloc(&ddinv, sizeof(double) * (n + 64)));
    HIPCHK(hipMalloc(&dinfo, sizeof(int)));
    HIPCHK(hipMalloc(&dctl, sizeof(PersistCtl)));
    HIPCHK(hipMemcpy(dA, system, sizeof(double) * m, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(dA + m, 0, sizeof(double) * m));
    HIPCHK(hipMemset(dinfo, 0, sizeof(int)));
    HIPCHK(hipMemset(dctl, 0, sizeof(PersistCtl)));
    HIPCHK(hipStreamSynchronize(nullptr));                     // (NULL-stream copies are not ordered against the engine's non-blocking streams: copy_d2d_now)
    if (mode == 0) {
        HIPCHK(chol_solve_device(dA, dA + m, n, dx, dinfo, nullptr));
    } else {
        if (workgroups < 1 || workgroups > 200) return fail(IPC_ERR_ARG, "ipc_debug_dense_solve: workgroups %d", workgroups);
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pchol_test_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(sizeof(double) * kLdsTotal)));
        hipLaunchKernelGGL(pchol_test_kernel, dim3(workgroups), dim3(kPT), sizeof(double) * kLdsTotal, nullptr, dA, dA + m, ddinv, n,
                           dx, dctl, dinfo);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(x_out, dx, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(info_out, dinfo, sizeof(int), hipMemcpyDeviceToHost));
    hipFree(dA); hipFree(dx); hipFree(ddinv); hipFree(dinfo); hipFree(dctl);
    return IPC_OK;
}

// The banded factorisation of the large-cluster kernel (cluster_band.hpp) on its own: `system` is the lower triangle in
// the banded layout (column j: the W band rows j .. j+W-1, then the m dense rows; nb band columns, m - 1 dense columns),
// the right-hand side in the last dense row.
extern "C" int ipc_debug_band_solve(int nb, int m, int W, const double* system, int workgroups, double* x_out, int* info_out)
{
    if (nb < 0 || m < 1 || W < 64 || !system || !x_out || !info_out) return fail(IPC_ERR_ARG, "ipc_debug_band_solve: bad argument");
    if (workgroups < 1 || workgroups > 200) return fail(IPC_ERR_ARG, "ipc_debug_band_solve: workgroups %d", workgroups);
    BandLayout B;
    B.nb = nb; B.m = m; B.W = W; B.ldb = W + m; B.n = nb + m - 1;
    if (B.n < 1) return fail(IPC_ERR_ARG, "ipc_debug_band_solve: empty system");
    const size_t sz = B.doubles();
    double *dA = nullptr, *dx = nullptr, *ddinv = nullptr;
    int* dinfo = nullptr;
    PersistCtl* dctl = nullptr;
    HIPCHK(hipMalloc(&dA, sizeof(double) * 2 * sz));
    HIPCHK(hipMalloc(&dx, sizeof(double) * (B.n + 64)));
    HIPCHK(hipMalloc(&ddinv, sizeof(double) * (B.n + 64)));
    HIPCHK(hipMalloc(&dinfo, sizeof(int)));
    HIPCHK(hipMalloc(&dctl, sizeof(PersistCtl)));
    HIPCHK(hipMemset(dA, 0, sizeof(double) * 2 * sz));
    HIPCHK(hipMemcpy(dA, system, sizeof(double) * (size_t)B.n * B.ldb, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(dinfo, 0, sizeof(int)));
    HIPCHK(hipMemset(dctl, 0, sizeof(PersistCtl)));
    HIPCHK(hipStreamSynchronize(nullptr));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&bband_test_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(sizeof(double) * kLdsTotal)));
    HIPCHK(hipMemset(ddinv + B.n + 32, 0, sizeof(double)));                 // (the word that holds 0.0)
    BandArgs Q{B, dA, dA + sz, ddinv, nullptr, nullptr, 0, 0, nullptr, ddinv + B.n + 32, -1, BandLayout{}, nullptr, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL(bband_test_kernel, dim3(workgroups), dim3(kPT), sizeof(double) * kLdsTotal, nullptr, Q, dx, dctl, dinfo);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    if (const char* reps_env = getenv("IPC_BAND_SOLVE_REPS")) {          // (tuning: the factorisation + back substitution alone, timed)
        const int reps = std::max(1, atoi(reps_env));
        double* dA0 = nullptr;
        HIPCHK(hipMalloc(&dA0, sizeof(double) * sz));
        HIPCHK(hipMemset(dA0, 0, sizeof(double) * sz));
        HIPCHK(hipMemcpy(dA0, system, sizeof(double) * (size_t)B.n * B.ldb, hipMemcpyHostToDevice));
        hipEvent_t e0, e1;
        HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
        float total_ms = 0.f;
        for (int r = 0; r < reps; ++r) {
            HIPCHK(hipMemcpyAsync(dA, dA0, sizeof(double) * sz, hipMemcpyDeviceToDevice, nullptr));
            HIPCHK(hipMemsetAsync(dctl, 0, sizeof(PersistCtl), nullptr));
            HIPCHK(hipEventRecord(e0, nullptr));
            hipLaunchKernelGGL(bband_test_kernel, dim3(workgroups), dim3(kPT), sizeof(double) * kLdsTotal, nullptr, Q, dx, dctl, dinfo);
            HIPCHK(hipEventRecord(e1, nullptr));
            HIPCHK(hipEventSynchronize(e1));
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, e0, e1));
            total_ms += ms;
        }
        fprintf(stderr, "[band solve] n %d W %d m %d workgroups %d: %.1f us per factorisation + back substitution (%d block columns, %.2f us each)\n",
                B.n, B.W, B.m, workgroups, 1e3 * total_ms / reps, (B.n + kCB - 1) / kCB, 1e3 * total_ms / reps / ((B.n + kCB - 1) / kCB));
        hipEventDestroy(e0); hipEventDestroy(e1); hipFree(dA0);
    }
    HIPCHK(hipMemcpy(x_out, dx, sizeof(double) * B.n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(info_out, dinfo, sizeof(int), hipMemcpyDeviceToHost));
    hipFree(dA); hipFree(dx); hipFree(ddinv); hipFree(dinfo); hipFree(dctl);
    return IPC_OK;
}

// The band structure the large-cluster solver finds for a set of loops (host code only: no GPU needed).  a / b: first /
// last vertex per loop.  order_out[q] = the loop at position q (band loops by first vertex, then the wide ones).
extern "C" int ipc_debug_band_plan(int d, int nl, const int* a, const int* b, int min_n, int* use_out, int* nlb_out, int* bwb_out,
                                   int* order_out)
{
    if ((d != 3 && d != 6) || nl < 0 || (nl > 0 && (!a || !b)) || !use_out || !nlb_out || !bwb_out)
        return fail(IPC_ERR_ARG, "ipc_debug_band_plan: bad argument");
    const BandPlan P = band_plan(d, std::vector<int>(a, a + nl), std::vector<int>(b, b + nl), min_n);
    *use_out = P.use ? 1 : 0; *nlb_out = P.nlb; *bwb_out = P.bwb;
    if (order_out && P.use) std::copy(P.order.begin(), P.order.end(), order_out);
    return IPC_OK;
}

// Set-only mode (one GPU): the consensus set of the batched formulation without the cells the set-max never reads.
// ipc_set_max only ever tests candidates whose own (diagonal) cell passed, and for those only their bits against each
// other: so the diagonal cells are solved first, then the pair cells among the candidates that passed.  Same accepted
// set as ipc_run by construction (tests hold them against each other); the bit matrix it builds on the way is NOT the
// consistency matrix (pairs with a failed candidate stay unsolved) and is not returned.
extern "C" int ipc_run_set_only(ipc_engine_t* h, uint8_t* accepted_out, int* solved_cells_out)
{
    if (!h) return fail(IPC_ERR_ARG, "ipc_run_set_only: NULL handle");
    if (h->N <= 0) return fail(IPC_ERR_STATE, "ipc_run_set_only: no candidates set");
    HIPCHK(hipSetDevice(h->device));
    const int N = h->N, words = (N + 63) / 64;
    const size_t need = (size_t)N * words;
    if (need > h->run_cap) {
        hipFree(h->d_upper); hipFree(h->d_bits); hipFree(h->d_acc);
        h->d_upper = h->d_bits = nullptr; h->d_acc = nullptr;
        HIPCHK(hipMalloc(&h->d_upper, sizeof(uint64_t) * need));
        HIPCHK(hipMalloc(&h->d_bits, sizeof(uint64_t) * need));
        HIPCHK(hipMalloc(&h->d_acc, (size_t)N + 64));
        h->run_cap = need;
    }
    int rc = solve_rows_impl(h, 0, 1, (uint64_t*)h->d_upper, h->own_stream, 1);
    const int diag_cells = h->last_cells;
    if (!rc) rc = solve_rows_impl(h, 0, 1, (uint64_t*)h->d_upper, h->own_stream, 2);
    if (!rc) rc = ipc_assemble_matrix(h, (const uint64_t*)h->d_upper, 1, (uint64_t*)h->d_bits, h->own_stream);
    if (!rc) rc = ipc_set_max(h, (const uint64_t*)h->d_bits, h->d_acc, h->own_stream);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(h->own_stream));
    if (solved_cells_out) *solved_cells_out = diag_cells + h->last_cells;
    if (accepted_out) HIPCHK(hipMemcpy(accepted_out, h->d_acc, (size_t)N, hipMemcpyDeviceToHost));
    return IPC_OK;
}

// Matrix mode over several GPUs of one node from ONE process (the C++ testers' IPC_AMD_DEVICES): engines[r] hold the
// same chain and candidate list on their own devices and act as rank r of world n_engines.  Every engine solves the
// rows ipc_row_assignment gives it, concurrently (one host thread per device); the shards are copied into the gathered
// layout on engines[0]'s device (hipMemcpyPeerAsync: xGMI peer copies, no collective library -- there is one gather,
// not a reduction), which assembles the matrix and runs the set-max.  Outputs as ipc_run.
extern "C" int ipc_run_sharded(ipc_engine_t** engines, int n_engines, uint64_t* bits_out, uint8_t* accepted_out)
{
    if (!engines || n_engines < 1) return fail(IPC_ERR_ARG, "ipc_run_sharded: no engines");
    for (int r = 0; r < n_engines; ++r) {
        if (!engines[r]) return fail(IPC_ERR_ARG, "ipc_run_sharded: engine %d is NULL", r);
        if (engines[r]->N != engines[0]->N || engines[r]->V != engines[0]->V || engines[r]->dim != engines[0]->dim)
            return fail(IPC_ERR_ARG, "ipc_run_sharded: engine %d holds a different problem", r);
        if (engines[r]->h_cand_ids != engines[0]->h_cand_ids) return fail(IPC_ERR_ARG, "ipc_run_sharded: engine %d holds other candidates", r);
    }
    ipc_engine* h0 = engines[0];
    if (h0->N <= 0) return fail(IPC_ERR_STATE, "ipc_run_sharded: no candidates set");
    if (n_engines == 1) return ipc_run(h0, bits_out, accepted_out);
    const int N = h0->N, words = (N + 63) / 64, world = n_engines, rpr = ipc_rows_per_rank(N, world);
    const size_t shard = (size_t)rpr * words, need = (size_t)N * words;
    HIPCHK(hipSetDevice(h0->device));
    if (need > h0->run_cap) {
        hipFree(h0->d_upper); hipFree(h0->d_bits); hipFree(h0->d_acc);
        h0->d_upper = h0->d_bits = nullptr; h0->d_acc = nullptr;
        HIPCHK(hipMalloc(&h0->d_upper, sizeof(uint64_t) * need));
        HIPCHK(hipMalloc(&h0->d_bits, sizeof(uint64_t) * need));
        HIPCHK(hipMalloc(&h0->d_acc, (size_t)N + 64));
        h0->run_cap = need;
    }
    unsigned long long* d_gathered = nullptr;
    HIPCHK(hipMalloc(&d_gathered, sizeof(uint64_t) * shard * world));
    std::vector<int> rcs(world, IPC_OK);
    std::vector<std::string> errs(world);
    std::vector<std::thread> workers;
    for (int r = 0; r < world; ++r) {
        workers.emplace_back([&, r]() {
            ipc_engine* h = engines[r];
            auto bad = [&](hipError_t e, const char* what) {
                if (e == hipSuccess) return false;
                rcs[r] = IPC_ERR_HIP; errs[r] = std::string(what) + ": " + hipGetErrorString(e);
                return true;
            };
            if (bad(hipSetDevice(h->device), "hipSetDevice")) return;
            unsigned long long* d_shard = nullptr;
            if (bad(hipMalloc(&d_shard, sizeof(uint64_t) * shard), "hipMalloc shard")) return;
            const int rc = ipc_solve_rows(h, r, world, (uint64_t*)d_shard, h->own_stream);
            if (rc) { rcs[r] = rc; errs[r] = ipc_last_error(); hipFree(d_shard); return; }
            if (!bad(hipMemcpyPeerAsync(d_gathered + (size_t)r * shard, h0->device, d_shard, h->device, sizeof(uint64_t) * shard, h->own_stream),
                     "hipMemcpyPeerAsync"))
                bad(hipStreamSynchronize(h->own_stream), "hipStreamSynchronize");
            hipFree(d_shard);
        });
    }
    for (auto& w : workers) w.join();
    HIPCHK(hipSetDevice(h0->device));
    for (int r = 0; r < world; ++r)
        if (rcs[r]) { hipFree(d_gathered); return fail((ipc_status)rcs[r], "ipc_run_sharded: rank %d: %s", r, errs[r].c_str()); }
    int rc = ipc_assemble_matrix(h0, (const uint64_t*)d_gathered, world, (uint64_t*)h0->d_bits, h0->own_stream);
    if (!rc) rc = ipc_set_max(h0, (const uint64_t*)h0->d_bits, h0->d_acc, h0->own_stream);
    if (rc) { hipFree(d_gathered); return rc; }
    HIPCHK(hipStreamSynchronize(h0->own_stream));
    HIPCHK(hipFree(d_gathered));
    if (bits_out) HIPCHK(hipMemcpy(bits_out, h0->d_bits, sizeof(uint64_t) * need, hipMemcpyDeviceToHost));
    if (accepted_out) HIPCHK(hipMemcpy(accepted_out, h0->d_acc, (size_t)N, hipMemcpyDeviceToHost));
    return IPC_OK;
}

