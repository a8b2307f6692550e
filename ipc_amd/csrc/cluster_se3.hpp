// General SE(3) cluster solve: the SE(3) counterpart of cluster_se2.hpp (faithful incremental mode
// and final map for VertexSE3 / EdgeSE3 graphs).  Same structure -- grid kernels over HBM arrays,
// prefix sums of Psi_j / w_j for the loop ranges, dense Cholesky of the (6 nl)^2 capacitance
// system, host-side dog-leg -- with the pose algebra of se3_cell.hpp:
//   Phi_j   = T(X~_j^-1) D(E_j)^-1 = [[U, K],[0, Vq]],  U = R_j RE_j^T, Vq = R_j Q_j^-1, K = 2 [t~_j]x Vq
//   Gamma_l = sigma_l D(E_l) T(X~_to) = sigma_l [[RE_l R_to^T, -2 RE_l R_to^T [t~_to]x],[0, Q_l R_to^T]]
//   h_j = rho_j + Ad_j h_{j-1}  <=>  omega_j = omega_{j-1} + R_j rho_q,
//                                    tau_j = tau_{j-1} + R_j rho_t + 2 omega_{j-1} x (t_j - t_{j-1})
#pragma once
#include "cluster_common.hpp"
#include "se3_cell.hpp"

namespace ipc {

struct ClusterDev3 {
    const double* chain; int estride; int lo; int L; int nl; int ld;   // ld = L + 2
    const double* cand; int cstride;
    const int *lfrom, *lto, *lcand;
    double *X, *Xn;                          // [12][ld]  R row-major (9), t (3)
    double *e, *en;                          // [6][ld]
    double *le, *len;                        // [6][nl]
    double *g, *m;                           // [6][ld]
    double *lg, *lm;                         // [6][nl]
    double *b, *h;                           // [6][ld]
    double *ps;                              // [27][ld]  prefix sums of Psi (21) and w (6)
    double *gam;                             // [36][nl]
    double *S; int ldS;                      // (NS+1) x NS column major; row NS holds the rhs
    double *rhs;                             // [NS]
    double *nu;                              // [6][nl]
    double *nd;                              // [6][ld]
    double *sc;                              // [6][ld]   R rho_q (3) -> omega, then R rho_t / term (3) -> tau
    const int *adj_ptr, *adj_item, *ev_ptr, *ev_item;
    double* partial;
    double* scal;
    double* chi_edges;
};

__device__ __forceinline__ Pose3 gk3_pose(const double* P, int ld, int i)
{
    Pose3 p;
#pragma unroll
    for (int k = 0; k < 9; ++k) p.R[k] = P[(size_t)k * ld + i];
#pragma unroll
    for (int k = 0; k < 3; ++k) p.t[k] = P[(size_t)(9 + k) * ld + i];
    return p;
}
__device__ __forceinline__ void gk3_rz(const double* rec, int stride, int idx, double* Rz, double* tz)
{
#pragma unroll
    for (int k = 0; k < 9; ++k) Rz[k] = rec[(size_t)(G_RZ + k) * stride + idx];
#pragma unroll
    for (int k = 0; k < 3; ++k) tz[k] = rec[(size_t)(G_TZ + k) * stride + idx];
}
__device__ __forceinline__ void gk3_sym(const double* rec, int stride, int field0, int idx, double* s)
{
#pragma unroll
    for (int k = 0; k < 21; ++k) s[k] = rec[(size_t)(field0 + k) * stride + idx];
}
__device__ __forceinline__ void gk3_ld6(const double* a, int ld, int i, double* v)
{
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = a[(size_t)k * ld + i];
}
__device__ __forceinline__ void gk3_st6(double* a, int ld, int i, const double* v)
{
#pragma unroll
    for (int k = 0; k < 6; ++k) a[(size_t)k * ld + i] = v[k];
}
// Phi_j blocks at pose X_j for edge E_j: U, Vq (3x3 each) and tt = t_j - o
__device__ __forceinline__ void gk3_phi(const Pose3& X, const Edge3& E, const double* o, double* U, double* Vq, double* tt)
{
    double Qi[9];
    m3_mult(X.R, E.RE, U);
    const double iw = 1.0 / E.qw;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) Qi[3 * i + k] = E.qv[i] * E.qv[k] * iw + (i == k ? E.qw : 0.0);
    Qi[1] += E.qv[2]; Qi[2] -= E.qv[1];      // - [v]x
    Qi[3] -= E.qv[2]; Qi[5] += E.qv[0];
    Qi[6] += E.qv[1]; Qi[7] -= E.qv[0];
    m3_mul(X.R, Qi, Vq);
#pragma unroll
    for (int k = 0; k < 3; ++k) tt[k] = X.t[k] - o[k];
}

// errors + chi2 of the poses Y; edges 1..L then loops
__device__ __forceinline__ void gk3_eval_at(const ClusterDev3& D, const double* Y, double* eo, double* leo, int i, double (&v)[1])
{
    if (i >= 1 && i <= D.L) {
        const int k = D.lo + i - 1;
        double Rz[9], tz[3], om[21];
        gk3_rz(D.chain, D.estride, k, Rz, tz);
        Edge3 E;
        se3_edge(gk3_pose(Y, D.ld, i - 1), gk3_pose(Y, D.ld, i), Rz, tz, E);
        gk3_st6(eo, D.ld, i, E.e);
        gk3_sym(D.chain, D.estride, G_OM, k, om);
        v[0] = sym6_quad(om, E.e);
    } else if (i > D.L && i <= D.L + D.nl) {
        const int l = i - D.L - 1, c = D.lcand[l];
        double Rz[9], tz[3], om[21];
        gk3_rz(D.cand, D.cstride, c, Rz, tz);
        Edge3 E;
        se3_edge(gk3_pose(Y, D.ld, D.lfrom[l]), gk3_pose(Y, D.ld, D.lto[l]), Rz, tz, E);
        gk3_st6(leo, D.nl, l, E.e);
        gk3_sym(D.cand, D.cstride, G_OM, c, om);
        v[0] = sym6_quad(om, E.e);
    }
}
__global__ void gk3_eval(ClusterDev3 D, const double* Y, double* eo, double* leo)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double v[1] = {0.0};
    gk3_eval_at(D, Y, eo, leo, i, v);
    gk_block_reduce_store<1>(v, D.partial + blockIdx.x * 4);
}

__device__ __forceinline__ void gk3_chi_edges_at(const ClusterDev3& D, int i)
{
    double om[21], e[6];
    if (i >= 1 && i <= D.L) {
        gk3_sym(D.chain, D.estride, G_OM, D.lo + i - 1, om);
        gk3_ld6(D.e, D.ld, i, e);
        D.chi_edges[i - 1] = sym6_quad(om, e);
    } else if (i > D.L && i <= D.L + D.nl) {
        const int l = i - D.L - 1;
        gk3_sym(D.cand, D.cstride, G_OM, D.lcand[l], om);
        gk3_ld6(D.le, D.nl, l, e);
        D.chi_edges[D.L + l] = sym6_quad(om, e);
    }
}
__global__ void gk3_chi_edges(ClusterDev3 D)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    gk3_chi_edges_at(D, i);
}

// g = D^T Om e, m = Ad^T g (odometry and loops); Gamma_l
__device__ __forceinline__ void gk3_force_at(const ClusterDev3& D, int i)
{
    if (i >= 1 && i <= D.L) {
        const int k = D.lo + i - 1;
        double Rz[9], tz[3], om[21], e[6], qo[6], g[6], m[6];
        gk3_rz(D.chain, D.estride, k, Rz, tz);
        Edge3 E;
        se3_edge(gk3_pose(D.X, D.ld, i - 1), gk3_pose(D.X, D.ld, i), Rz, tz, E);
        gk3_sym(D.chain, D.estride, G_OM, k, om);
        gk3_ld6(D.e, D.ld, i, e);
        sym6_mul(om, e, qo);
        se3_Dt(E, qo, g);
        se3_Adt(E, g, m);
        gk3_st6(D.g, D.ld, i, g);
        gk3_st6(D.m, D.ld, i, m);
    } else if (i > D.L && i <= D.L + D.nl) {
        const int l = i - D.L - 1, c = D.lcand[l];
        double Rz[9], tz[3], om[21], e[6], qo[6], g[6], m[6];
        gk3_rz(D.cand, D.cstride, c, Rz, tz);
        const Pose3 pt = gk3_pose(D.X, D.ld, D.lto[l]);
        Edge3 E;
        se3_edge(gk3_pose(D.X, D.ld, D.lfrom[l]), pt, Rz, tz, E);
        gk3_sym(D.cand, D.cstride, G_OM, c, om);
        gk3_ld6(D.le, D.nl, l, e);
        sym6_mul(om, e, qo);
        se3_Dt(E, qo, g);
        se3_Adt(E, g, m);
        gk3_st6(D.lg, D.nl, l, g);
        gk3_st6(D.lm, D.nl, l, m);
        // Gamma_l = sigma [[P, -2 P [tt]x],[0, Q Rto^T]],  P = RE Rto^T, Q = w I + [v]x
        const double sg = D.lto[l] > D.lfrom[l] ? 1.0 : -1.0;
        const double tt[3] = {pt.t[0] - D.X[(size_t)9 * D.ld], pt.t[1] - D.X[(size_t)10 * D.ld], pt.t[2] - D.X[(size_t)11 * D.ld]};
        double P[9], QR[9];
        m3_mult(E.RE, pt.R, P);
        const double Q[9] = {E.qw, -E.qv[2], E.qv[1], E.qv[2], E.qw, -E.qv[0], -E.qv[1], E.qv[0], E.qw};
        m3_mult(Q, pt.R, QR);
        const double TX[9] = {0.0, -tt[2], tt[1], tt[2], 0.0, -tt[0], -tt[1], tt[0], 0.0};
        double PT[9];
        m3_mul(P, TX, PT);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                D.gam[(size_t)(6 * r + cc) * D.nl + l] = sg * P[3 * r + cc];
                D.gam[(size_t)(6 * r + 3 + cc) * D.nl + l] = sg * (-2.0 * PT[3 * r + cc]);
                D.gam[(size_t)(6 * (r + 3) + cc) * D.nl + l] = 0.0;
                D.gam[(size_t)(6 * (r + 3) + 3 + cc) * D.nl + l] = sg * QR[3 * r + cc];
            }
    }
}
__global__ void gk3_force(ClusterDev3 D)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    gk3_force_at(D, i);
}

// b_j = m_{j+1} - g_j + loop terms; partial b^T b
__device__ __forceinline__ void gk3_b_at(const ClusterDev3& D, int j, double (&v)[1])
{
    if (j >= 1 && j <= D.L) {
        double b[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            b[k] = -D.g[(size_t)k * D.ld + j];
            if (j < D.L) b[k] += D.m[(size_t)k * D.ld + j + 1];
        }
        for (int q = D.adj_ptr[j]; q < D.adj_ptr[j + 1]; ++q) {
            const int it = D.adj_item[q], l = it >> 1;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                if (it & 1) b[k] -= D.lg[(size_t)k * D.nl + l];
                else b[k] += D.lm[(size_t)k * D.nl + l];
            }
        }
        gk3_st6(D.b, D.ld, j, b);
#pragma unroll
        for (int k = 0; k < 6; ++k) v[0] += b[k] * b[k];
    }
}
__global__ void gk3_b(ClusterDev3 D)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    double v[1] = {0.0};
    gk3_b_at(D, j, v);
    gk_block_reduce_store<1>(v, D.partial + blockIdx.x * 4);
}

// partial b^T H b; Psi_j / w_j before the prefix sums
__device__ __forceinline__ void gk3_bHb_psi_at(const ClusterDev3& D, int i, double (&v)[1])
{
    if (i >= 1 && i <= D.L) {
        const int k = D.lo + i - 1;
        double Rz[9], tz[3], om[21], sg[21], va[6] = {0, 0, 0, 0, 0, 0}, vb[6], w[6], e[6];
        gk3_rz(D.chain, D.estride, k, Rz, tz);
        const Pose3 X = gk3_pose(D.X, D.ld, i);
        Edge3 E;
        se3_edge(gk3_pose(D.X, D.ld, i - 1), X, Rz, tz, E);
        if (i > 1) gk3_ld6(D.b, D.ld, i - 1, va);
        gk3_ld6(D.b, D.ld, i, vb);
        se3_apply_J(E, va, vb, w);
        gk3_sym(D.chain, D.estride, G_OM, k, om);
        v[0] = sym6_quad(om, w);
        gk3_sym(D.chain, D.estride, G_SG, k, sg);
        gk3_ld6(D.e, D.ld, i, e);
        const double o[3] = {D.X[(size_t)9 * D.ld], D.X[(size_t)10 * D.ld], D.X[(size_t)11 * D.ld]};
        double U[9], Vq[9], tt[3], K[9];
        gk3_phi(X, E, o, U, Vq, tt);
#pragma unroll
        for (int c = 0; c < 3; ++c) {                 // K[:,c] = 2 tt x Vq[:,c]
            K[0 + c] = 2 * (tt[1] * Vq[6 + c] - tt[2] * Vq[3 + c]);
            K[3 + c] = 2 * (tt[2] * Vq[0 + c] - tt[0] * Vq[6 + c]);
            K[6 + c] = 2 * (tt[0] * Vq[3 + c] - tt[1] * Vq[0 + c]);
        }
        auto phi = [&](int r, int q) -> double {
            if (r < 3) return q < 3 ? U[3 * r + q] : K[3 * r + (q - 3)];
            return q < 3 ? 0.0 : Vq[3 * (r - 3) + (q - 3)];
        };
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            double ps[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double acc = 0.0;
#pragma unroll
                for (int a = (r < 3 ? 0 : 3); a < 6; ++a) acc += phi(r, a) * sg[sym6_idx(a, c)];
                ps[c] = acc;
            }
#pragma unroll
            for (int c = r; c < 6; ++c) {
                double acc = 0.0;
#pragma unroll
                for (int q = (c < 3 ? 0 : 3); q < 6; ++q) acc += ps[q] * phi(c, q);
                D.ps[(size_t)sym6_idx(r, c) * D.ld + i] = acc;
            }
            double acc = 0.0;
#pragma unroll
            for (int a = (r < 3 ? 0 : 3); a < 6; ++a) acc += phi(r, a) * e[a];
            D.ps[(size_t)(21 + r) * D.ld + i] = acc;
        }
    } else if (i > D.L && i <= D.L + D.nl) {
        const int l = i - D.L - 1, c = D.lcand[l];
        const int f = D.lfrom[l], t = D.lto[l];
        double Rz[9], tz[3], om[21], va[6] = {0, 0, 0, 0, 0, 0}, vb[6] = {0, 0, 0, 0, 0, 0}, w[6];
        gk3_rz(D.cand, D.cstride, c, Rz, tz);
        Edge3 E;
        se3_edge(gk3_pose(D.X, D.ld, f), gk3_pose(D.X, D.ld, t), Rz, tz, E);
        if (f > 0) gk3_ld6(D.b, D.ld, f, va);
        if (t > 0) gk3_ld6(D.b, D.ld, t, vb);
        se3_apply_J(E, va, vb, w);
        gk3_sym(D.cand, D.cstride, G_OM, c, om);
        v[0] = sym6_quad(om, w);
    } else if (i == 0) {
#pragma unroll
        for (int k = 0; k < 27; ++k) D.ps[(size_t)k * D.ld] = 0.0;
    }
}
__global__ void gk3_bHb_psi(ClusterDev3 D)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double v[1] = {0.0};
    gk3_bHb_psi_at(D, i, v);
    gk_block_reduce_store<1>(v, D.partial + blockIdx.x * 4);
}

// capacitance system: S (lower triangle of 6x6 blocks) and rhs; put / put_rhs as in gk_assemble_core (cluster_se2.hpp)
template <class Put, class PutRhs>
__device__ __forceinline__ void gk3_assemble_core(const ClusterDev3& D, int l1, int l2, Put put, PutRhs put_rhs)     // row block l1, column block l2
{
    if (l2 >= D.nl || l1 >= D.nl || l2 > l1) return;
    const int lo1 = min(D.lfrom[l1], D.lto[l1]), hi1 = max(D.lfrom[l1], D.lto[l1]);
    const int lo2 = min(D.lfrom[l2], D.lto[l2]), hi2 = max(D.lfrom[l2], D.lto[l2]);
    const int a = max(lo1, lo2), bq = min(hi1, hi2);
    double Mm[21];
#pragma unroll
    for (int k = 0; k < 21; ++k) Mm[k] = bq > a ? D.ps[(size_t)k * D.ld + bq] - D.ps[(size_t)k * D.ld + a] : 0.0;
    double sgl[21];
    if (l1 == l2) gk3_sym(D.cand, D.cstride, G_SG, D.lcand[l1], sgl);
#pragma unroll 1
    for (int r = 0; r < 6; ++r) {
        double g1[6], T[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) g1[q] = D.gam[(size_t)(6 * r + q) * D.nl + l1];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            double acc = 0.0;
#pragma unroll
            for (int p = 0; p < 6; ++p) acc += g1[p] * Mm[sym6_idx(p, q)];
            T[q] = acc;
        }
#pragma unroll 1
        for (int c = 0; c < 6; ++c) {
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < 6; ++q) acc += T[q] * D.gam[(size_t)(6 * c + q) * D.nl + l2];
            if (l1 == l2) acc += sgl[sym6_idx(r, c)];
            put(6 * l1 + r, 6 * l2 + c, acc);
        }
        if (l1 == l2) {
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < 6; ++q)
                acc += g1[q] * (D.ps[(size_t)(21 + q) * D.ld + hi1] - D.ps[(size_t)(21 + q) * D.ld + lo1]);
            put_rhs(6 * l1 + r, D.le[(size_t)r * D.nl + l1] - acc);
        }
    }
}
// One ROW r of the same block (the banded kernel gives every block row a thread of its own: six times the threads, and the
// operands of a row -- Gamma_l2 whole, row r of Gamma_l1, the prefix-sum differences -- are requested together, two trips
// to L2 per row instead of seven per row in the rolled form above, whose threads each walked through a dozen block pairs:
// 17 % of an iteration).  The arithmetic and its order are gk3_assemble_core's: same bits.
template <class Put, class PutRhs>
__device__ __forceinline__ void gk3_assemble_row(const ClusterDev3& D, int l1, int l2, int r, Put put, PutRhs put_rhs)
{
    if (l2 >= D.nl || l1 >= D.nl || l2 > l1) return;
    const int lo1 = min(D.lfrom[l1], D.lto[l1]), hi1 = max(D.lfrom[l1], D.lto[l1]);
    const int lo2 = min(D.lfrom[l2], D.lto[l2]), hi2 = max(D.lfrom[l2], D.lto[l2]);
    const int a = max(lo1, lo2), bq = min(hi1, hi2);
    const bool ov = bq > a, diag = l1 == l2;
    const int ia = ov ? a : 0, ib = ov ? bq : 0;             // (index 0 of every row is a valid slot: no predicated loads)
    double Pa[21], Pb[21], g1[6], G2[36];
#pragma unroll
    for (int k = 0; k < 21; ++k) { Pb[k] = D.ps[(size_t)k * D.ld + ib]; Pa[k] = D.ps[(size_t)k * D.ld + ia]; }
#pragma unroll
    for (int q = 0; q < 6; ++q) g1[q] = D.gam[(size_t)(6 * r + q) * D.nl + l1];
#pragma unroll
    for (int k = 0; k < 36; ++k) G2[k] = D.gam[(size_t)k * D.nl + l2];
    __builtin_amdgcn_sched_barrier(0);
    double Mm[21], T[6];
#pragma unroll
    for (int k = 0; k < 21; ++k) Mm[k] = ov ? Pb[k] - Pa[k] : 0.0;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        double acc = 0.0;
#pragma unroll
        for (int p = 0; p < 6; ++p) acc += g1[p] * Mm[sym6_idx(p, q)];
        T[q] = acc;
    }
    double sgl[21];
    if (diag) gk3_sym(D.cand, D.cstride, G_SG, D.lcand[l1], sgl);
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < 6; ++q) acc += T[q] * G2[6 * c + q];
        if (diag) acc += sgl[sym6_idx(r, c)];
        put(6 * l1 + r, 6 * l2 + c, acc);
    }
    if (diag) {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < 6; ++q)
            acc += g1[q] * (D.ps[(size_t)(21 + q) * D.ld + hi1] - D.ps[(size_t)(21 + q) * D.ld + lo1]);
        put_rhs(6 * l1 + r, D.le[(size_t)r * D.nl + l1] - acc);
    }
}
__device__ __forceinline__ void gk3_assemble_at(const ClusterDev3& D, int l1, int l2)
{
    const int NS = 6 * D.nl;
    gk3_assemble_core(D, l1, l2, [&](int row, int col, double v) { st_shared(&D.S[(size_t)col * D.ldS + row], v); },
                      [&](int col, double v) { st_shared(&D.S[(size_t)col * D.ldS + NS], v); });
}
__global__ void gk3_assemble(ClusterDev3 D)
{
    gk3_assemble_at(D, blockIdx.y, blockIdx.x * blockDim.x + threadIdx.x);
}

// nu_l = Gamma_l^T mu_l
__device__ __forceinline__ void gk3_nu_at(const ClusterDev3& D, int l)
{
    if (l >= D.nl) return;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < 6; ++r) t += D.gam[(size_t)(6 * r + c) * D.nl + l] * D.rhs[6 * l + r];
        D.nu[(size_t)c * D.nl + l] = t;
    }
}
__global__ void gk3_nu(ClusterDev3 D)
{
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    gk3_nu_at(D, l);
}

__device__ __forceinline__ void gk3_events_at(const ClusterDev3& D, int j)
{
    if (j > D.L + 1) return;
    double a[6] = {0, 0, 0, 0, 0, 0};
    if (j >= 1) {
        for (int q = D.ev_ptr[j]; q < D.ev_ptr[j + 1]; ++q) {
            const int it = D.ev_item[q], l = it >> 1;
            const double sgn = (it & 1) ? -1.0 : 1.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) a[k] += sgn * D.nu[(size_t)k * D.nl + l];
        }
    }
    gk3_st6(D.nd, D.ld, j, a);
}
__global__ void gk3_events(ClusterDev3 D)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    gk3_events_at(D, j);
}

// u_j = -Cov Phi^T n_j - e_j, rho = D^-1 u;  R_j rho_q -> sc[0..2], R_j rho_t -> sc[3..5]
__device__ __forceinline__ void gk3_rho_at(const ClusterDev3& D, int i)
{
    if (i < 1 || i > D.L) return;
    const int k = D.lo + i - 1;
    double Rz[9], tz[3], sg[21], nn[6], e[6];
    gk3_rz(D.chain, D.estride, k, Rz, tz);
    const Pose3 X = gk3_pose(D.X, D.ld, i);
    Edge3 E;
    se3_edge(gk3_pose(D.X, D.ld, i - 1), X, Rz, tz, E);
    gk3_ld6(D.nd, D.ld, i, nn);
    gk3_ld6(D.e, D.ld, i, e);
    const double o[3] = {D.X[(size_t)9 * D.ld], D.X[(size_t)10 * D.ld], D.X[(size_t)11 * D.ld]};
    double U[9], Vq[9], tt[3], wv[6], cr[3], y3[3];
    gk3_phi(X, E, o, U, Vq, tt);
    cross3(tt, nn, cr);
#pragma unroll
    for (int q = 0; q < 3; ++q) y3[q] = nn[3 + q] - 2 * cr[q];
    m3_tvec(U, nn, wv);
    m3_tvec(Vq, y3, wv + 3);
    gk3_sym(D.chain, D.estride, G_SG, k, sg);
    double v[6], u[6], rho[6], rq[3], rt[3];
    sym6_mul(sg, wv, v);
#pragma unroll
    for (int q = 0; q < 6; ++q) u[q] = -v[q] - e[q];
    se3_Dinv(E, u, rho);
    m3_vec(X.R, rho + 3, rq);
    m3_vec(X.R, rho, rt);
#pragma unroll
    for (int q = 0; q < 3; ++q) { D.sc[(size_t)q * D.ld + i] = rq[q]; D.sc[(size_t)(3 + q) * D.ld + i] = rt[q]; }
}
__global__ void gk3_rho(ClusterDev3 D)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    gk3_rho_at(D, i);
}
// term = R rho_t + 2 omega_{j-1} x (t_j - t_{j-1})   (sc[0..2] already holds the inclusive omega prefix)
__device__ __forceinline__ void gk3_term_at(const ClusterDev3& D, int i)
{
    if (i < 1 || i > D.L) return;
    double op[3] = {0, 0, 0}, d3[3], c[3];
    if (i > 1) {
#pragma unroll
        for (int q = 0; q < 3; ++q) op[q] = D.sc[(size_t)q * D.ld + i - 1];
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) d3[q] = D.X[(size_t)(9 + q) * D.ld + i] - D.X[(size_t)(9 + q) * D.ld + i - 1];
    cross3(op, d3, c);
#pragma unroll
    for (int q = 0; q < 3; ++q) D.sc[(size_t)(3 + q) * D.ld + i] += 2 * c[q];
}
__global__ void gk3_term(ClusterDev3 D)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    gk3_term_at(D, i);
}
// h = (R^T tau, R^T omega); partial |h|^2, b.h
__device__ __forceinline__ void gk3_h_at(const ClusterDev3& D, int i, double (&v)[2])
{
    if (i >= 1 && i <= D.L) {
        double R[9], om3[3], ta3[3], h[6], b[6];
#pragma unroll
        for (int q = 0; q < 9; ++q) R[q] = D.X[(size_t)q * D.ld + i];
#pragma unroll
        for (int q = 0; q < 3; ++q) { om3[q] = D.sc[(size_t)q * D.ld + i]; ta3[q] = D.sc[(size_t)(3 + q) * D.ld + i]; }
        m3_tvec(R, ta3, h);
        m3_tvec(R, om3, h + 3);
        gk3_st6(D.h, D.ld, i, h);
        gk3_ld6(D.b, D.ld, i, b);
#pragma unroll
        for (int q = 0; q < 6; ++q) { v[0] += h[q] * h[q]; v[1] += b[q] * h[q]; }
    }
}
__global__ void gk3_h(ClusterDev3 D)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double v[2] = {0.0, 0.0};
    gk3_h_at(D, i, v);
    gk_block_reduce_store<2>(v, D.partial + blockIdx.x * 4);
}
__device__ __forceinline__ void gk3_blend_at(const ClusterDev3& D, double alpha, int i, double (&v)[2])
{
    if (i >= 1 && i <= D.L) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double sk = alpha * D.b[(size_t)k * D.ld + i], ak = D.h[(size_t)k * D.ld + i] - sk;
            v[0] += sk * ak;
            v[1] += ak * ak;
        }
    }
}
__global__ void gk3_blend(ClusterDev3 D, double alpha)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double v[2] = {0.0, 0.0};
    gk3_blend_at(D, alpha, i, v);
    gk_block_reduce_store<2>(v, D.partial + blockIdx.x * 4);
}
// trial poses Xn = X * fromVectorMQT(p b + q h) (VertexSE3::oplusImpl); partial "changed" count
__device__ __forceinline__ void gk3_update_at(const ClusterDev3& D, double p, double q, int i, double (&v)[1])
{
    if (i == 0) {
#pragma unroll
        for (int k = 0; k < 12; ++k) D.Xn[(size_t)k * D.ld] = D.X[(size_t)k * D.ld];
    }
    if (i >= 1 && i <= D.L) {
        const Pose3 X = gk3_pose(D.X, D.ld, i);
        double dl[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) dl[k] = fma(p, D.b[(size_t)k * D.ld + i], q * D.h[(size_t)k * D.ld + i]);
        double wq = 1.0 - (dl[3] * dl[3] + dl[4] * dl[4] + dl[5] * dl[5]);
        double dR[9], Rn[9], rt3[3];
        if (wq < 0) R_from_quat(1.0, 0.0, 0.0, 0.0, dR);
        else { wq = sqrt(wq); R_from_quat(wq, dl[3], dl[4], dl[5], dR); }
        m3_mul(X.R, dR, Rn);
        m3_vec(X.R, dl, rt3);
        bool chg = false;
#pragma unroll
        for (int k = 0; k < 9; ++k) { D.Xn[(size_t)k * D.ld + i] = Rn[k]; chg |= Rn[k] != X.R[k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double tn = X.t[k] + rt3[k];
            D.Xn[(size_t)(9 + k) * D.ld + i] = tn;
            chg |= tn != X.t[k];
        }
        v[0] = chg ? 1.0 : 0.0;
    }
}
__global__ void gk3_update(ClusterDev3 D, double p, double q)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double v[1] = {0.0};
    gk3_update_at(D, p, q, i, v);
    gk_block_reduce_store<1>(v, D.partial + blockIdx.x * 4);
}

// ---- literal normal equations (Levenberg retry, cluster_common.hpp::cluster_dogleg): SE(3) counterpart of
// cluster_se2.hpp's gk_dense_H / gk_h_from_dense / gk_quad_bh (6 x 6 blocks, n = 6 L unknowns) ------------------
__device__ __forceinline__ void gk3_edge_J(const Edge3& E, double (&Ja)[6][6], double (&Jb)[6][6])
{
    const double z[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        double u[6] = {0, 0, 0, 0, 0, 0}, w[6];
        u[k] = 1.0;
        se3_apply_J(E, u, z, w);
#pragma unroll
        for (int r = 0; r < 6; ++r) Ja[r][k] = w[r];
        se3_apply_J(E, z, u, w);
#pragma unroll
        for (int r = 0; r < 6; ++r) Jb[r][k] = w[r];
    }
}
// out += A^T Om B
__device__ __forceinline__ void gk3_atob(const double (&A)[6][6], const double* om, const double (&B)[6][6], double (&out)[6][6])
{
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        double col[6], oc[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) col[r] = B[r][c];
        sym6_mul(om, col, oc);
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < 6; ++q) acc += A[q][r] * oc[q];
            out[r][c] += acc;
        }
    }
}
template <class St>
__global__ void gk3_literal_H(ClusterDev3 D, St S, double lambda)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < 1 || p > D.L) return;
    double dg[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) dg[r][c] = r == c ? lambda : 0.0;
    auto put = [&](int prow, int pcol, const double (&B)[6][6], bool add) {
        const int ui0 = S.unk(prow), uj0 = S.unk(pcol);
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const int ui = ui0 + r, uj = uj0 + c;         // (cluster_se2.hpp::gk_literal_H)
                double* q = ui >= uj ? S.lower(ui, uj) : S.lower(uj, ui);
                *q = add ? *q + B[r][c] : B[r][c];
            }
    };
    double Rz[9], tz[3], om[21];
    {
        const int k = D.lo + p - 1;
        gk3_rz(D.chain, D.estride, k, Rz, tz);
        Edge3 E;
        se3_edge(gk3_pose(D.X, D.ld, p - 1), gk3_pose(D.X, D.ld, p), Rz, tz, E);
        double Ja[6][6], Jb[6][6];
        gk3_edge_J(E, Ja, Jb);
        gk3_sym(D.chain, D.estride, G_OM, k, om);
        gk3_atob(Jb, om, Jb, dg);
        if (p >= 2) {
            double off[6][6] = {};
            gk3_atob(Jb, om, Ja, off);
            put(p, p - 1, off, false);
        }
    }
    if (p < D.L) {
        const int k = D.lo + p;
        gk3_rz(D.chain, D.estride, k, Rz, tz);
        Edge3 E;
        se3_edge(gk3_pose(D.X, D.ld, p), gk3_pose(D.X, D.ld, p + 1), Rz, tz, E);
        double Ja[6][6], Jb[6][6];
        gk3_edge_J(E, Ja, Jb);
        gk3_sym(D.chain, D.estride, G_OM, k, om);
        gk3_atob(Ja, om, Ja, dg);
    }
    for (int q = D.adj_ptr[p]; q < D.adj_ptr[p + 1]; ++q) {
        const int it = D.adj_item[q], l = it >> 1, c = D.lcand[l];
        const int f = D.lfrom[l], t = D.lto[l];
        gk3_rz(D.cand, D.cstride, c, Rz, tz);
        Edge3 E;
        se3_edge(gk3_pose(D.X, D.ld, f), gk3_pose(D.X, D.ld, t), Rz, tz, E);
        double Ja[6][6], Jb[6][6];
        gk3_edge_J(E, Ja, Jb);
        gk3_sym(D.cand, D.cstride, G_OM, c, om);
        if (it & 1) gk3_atob(Jb, om, Jb, dg); else gk3_atob(Ja, om, Ja, dg);
        const int other = (it & 1) ? f : t;
        if (other >= 1 && other < p) {
            double off[6][6] = {};
            if (it & 1) gk3_atob(Jb, om, Ja, off); else gk3_atob(Ja, om, Jb, off);
            put(p, other, off, true);
        }
    }
    const int u0 = S.unk(p);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
#pragma unroll
        for (int c = 0; c <= r; ++c) *S.lower(u0 + r, u0 + c) = dg[r][c];
        *S.lower(S.n(), u0 + r) = D.b[(size_t)r * D.ld + p];
    }
}
__global__ void gk3_h_from_dense(ClusterDev3 D, const double* x)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double v[2] = {0.0, 0.0};
    if (i >= 1 && i <= D.L) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double hk = x[6 * (size_t)(i - 1) + k];
            D.h[(size_t)k * D.ld + i] = hk;
            v[0] += hk * hk;
            v[1] += D.b[(size_t)k * D.ld + i] * hk;
        }
    }
    gk_block_reduce_store<2>(v, D.partial + blockIdx.x * 4);
}
__global__ void gk3_quad_bh(ClusterDev3 D)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double v[2] = {0.0, 0.0};
    int f = -1, t = -1;
    double Rz[9], tz[3], om[21];
    if (i >= 1 && i <= D.L) {
        const int k = D.lo + i - 1;
        f = i - 1; t = i;
        gk3_rz(D.chain, D.estride, k, Rz, tz);
        gk3_sym(D.chain, D.estride, G_OM, k, om);
    } else if (i > D.L && i <= D.L + D.nl) {
        const int l = i - D.L - 1, c = D.lcand[l];
        f = D.lfrom[l]; t = D.lto[l];
        gk3_rz(D.cand, D.cstride, c, Rz, tz);
        gk3_sym(D.cand, D.cstride, G_OM, c, om);
    }
    if (f >= 0) {
        Edge3 E;
        se3_edge(gk3_pose(D.X, D.ld, f), gk3_pose(D.X, D.ld, t), Rz, tz, E);
        double va[6] = {0, 0, 0, 0, 0, 0}, vb[6] = {0, 0, 0, 0, 0, 0}, wb[6], wh[6], o[6];
        if (f > 0) gk3_ld6(D.b, D.ld, f, va);
        if (t > 0) gk3_ld6(D.b, D.ld, t, vb);
        se3_apply_J(E, va, vb, wb);
#pragma unroll
        for (int k = 0; k < 6; ++k) { va[k] = 0.0; vb[k] = 0.0; }
        if (f > 0) gk3_ld6(D.h, D.ld, f, va);
        if (t > 0) gk3_ld6(D.h, D.ld, t, vb);
        se3_apply_J(E, va, vb, wh);
        sym6_mul(om, wh, o);
#pragma unroll
        for (int k = 0; k < 6; ++k) { v[0] += wb[k] * o[k]; v[1] += wh[k] * o[k]; }
    }
    gk_block_reduce_store<2>(v, D.partial + blockIdx.x * 4);
}

// ------------------------------------------------------------------------------------------
// host driver
// ------------------------------------------------------------------------------------------
class ClusterSolver3 {
public:
    ~ClusterSolver3() { release(); }
    double term_eps = 0.0;             // convergence shortcut of the trial loop (Se2View::term_eps)

    // src: poses [12][src_ld] with global indexing; the optimised poses stay in result() ([12][ld()],
    // local indexing 0..hi-lo).
    hipError_t solve(hipStream_t st, const double* chain, int estride, const double* cand, int cstride,
                     const double* src, int src_ld, int lo, int hi, const std::vector<int>& members, const int* from,
                     const int* to, int iterations, ClusterOut& out, std::vector<double>* chi_host);
    const double* result() const { return dev_.X; }
    int ld() const { return dev_.ld; }

    // ---- Ops of cluster_dogleg ----
    hipError_t evaluate_committed(double& chi) { return evaluate(dev_.X, dev_.e, dev_.le, chi); }
    hipError_t linearize(double& bb, double& bHb, double& hh, double& bh, int& info);
    hipError_t blend(double alpha, double& c, double& bma);
    hipError_t trial(double p, double q, double& newChi, bool& anyChanged);
    void commit() { std::swap(dev_.X, dev_.Xn); std::swap(dev_.e, dev_.en); std::swap(dev_.le, dev_.len); }
    hipError_t max_edge_chi2(double& mx);
    hipError_t damped_solve(double lambda, bool& ok, double& hh, double& bh, double& bHh, double& hHh);
    bool allow_damping = true;          // Levenberg retry of a failed linear solve on the literal normal equations: banded + bordered store
                                        // (cluster_literal_band.hpp) where the loops form a band, dense up to kMaxDenseN unknowns otherwise
    static constexpr int kMaxDenseN = 24000;
    static constexpr int kD = 6;
    int literal_band_min_n = 3072;      // IPC_LITERAL_BAND_MIN_N: smaller systems keep the dense store (bit for bit as in rounds 3 - 5); < 0: never banded
    LiteralBand* lband = nullptr; bool lband_stale = true;
    void want_plain_solve(bool w) { want_plain_ = w; }
    bool want_plain_ = true;
    long literal_band_solves() const;
    const LoopTables& tables() const { return tab_; }
    hipStream_t stream() const { return st_; }
    template <class St> void launch_literal_H(const St& S, double lambda)
    {
        auto& D = dev_;
        hipLaunchKernelGGL(gk3_literal_H<St>, dim3((D.L + 1 + 63) / 64), dim3(64), 0, st_, D, S, lambda);
    }

private:
    double* d_H_ = nullptr; size_t capH_ = 0;
    ClusterDev3 dev_{};
    hipStream_t st_ = nullptr;
    int nblk_ = 1;
    std::vector<double>* chi_host_ = nullptr;
    int capL_ = 0, capNl_ = 0;
    double *d_edge_ = nullptr, *d_loop_ = nullptr, *d_S_ = nullptr, *d_partial_ = nullptr, *d_scal_ = nullptr;
    int *d_int_ = nullptr, *d_info_ = nullptr;
    double* h_scal_ = nullptr;
    LoopTables tab_;

    void release()
    {
        if (lband) { literal_band_free(lband); lband = nullptr; }
        hipFree(d_edge_); hipFree(d_loop_); hipFree(d_S_); hipFree(d_partial_); hipFree(d_scal_);
        hipFree(d_int_); hipFree(d_H_); d_H_ = nullptr; capH_ = 0;
        if (h_scal_) hipHostFree(h_scal_);
        d_edge_ = d_loop_ = d_S_ = d_partial_ = d_scal_ = nullptr; d_int_ = d_info_ = nullptr; h_scal_ = nullptr;
        capL_ = capNl_ = 0;
    }
    hipError_t ensure(int L, int nl);
    hipError_t fetch(int n)
    {
        (void)n;                     // scalars [0, 12) and the solver's info word at [12] travel in one copy
        IPC_CL_CHK(hipMemcpyAsync(h_scal_, d_scal_, sizeof(double) * 13, hipMemcpyDeviceToHost, st_));
        return hipStreamSynchronize(st_);
    }
    void sum_partials(int K, int off)
    {
        hipLaunchKernelGGL(gk_sum, dim3(1), dim3(64), 0, st_, (const double*)dev_.partial, nblk_, K, dev_.scal, off);
    }
    hipError_t evaluate(const double* Y, double* eo, double* leo, double& chi)
    {
        hipLaunchKernelGGL(gk3_eval, dim3(nblk_), dim3(kGB), 0, st_, dev_, Y, eo, leo);
        sum_partials(1, 7);
        IPC_CL_CHK(fetch(8));
        chi = h_scal_[7];
        return hipSuccess;
    }
};

inline hipError_t ClusterSolver3::ensure(int L, int nl)
{
    if (!h_scal_) {
        IPC_CL_CHK(hipHostMalloc(&h_scal_, sizeof(double) * 16));
        IPC_CL_CHK(hipMalloc(&d_scal_, sizeof(double) * 16));
        d_info_ = reinterpret_cast<int*>(d_scal_ + 12);
    }
    if (L > capL_ || nl > capNl_) {
        const int nL = std::max(L, capL_), nN = std::max(nl, capNl_);
        hipFree(d_edge_); hipFree(d_loop_); hipFree(d_S_); hipFree(d_partial_); hipFree(d_int_);
        d_edge_ = d_loop_ = d_S_ = d_partial_ = nullptr; d_int_ = nullptr;
        capL_ = capNl_ = 0;
        const size_t ld = (size_t)nL + 2;
        IPC_CL_CHK(hipMalloc(&d_edge_, sizeof(double) * (99 * ld + ld + nN)));
        IPC_CL_CHK(hipMalloc(&d_loop_, sizeof(double) * (72 * (size_t)nN + 8)));
        IPC_CL_CHK(hipMalloc(&d_S_, sizeof(double) * 2 * (6 * (size_t)nN + 1) * (6 * (size_t)nN)));   // system + factor
        IPC_CL_CHK(hipMalloc(&d_partial_, sizeof(double) * 4 * ((nL + nN + 1 + kGB) / kGB + 1)));
        IPC_CL_CHK(hipMalloc(&d_int_, sizeof(int) * LoopTables::capacity(nL, nN)));
        capL_ = nL; capNl_ = nN;
    }
    return hipSuccess;
}

inline hipError_t ClusterSolver3::linearize(double& bb, double& bHb, double& hh, double& bh, int& info)
{
    ClusterDev3& D = dev_;
    const dim3 grid(nblk_), block(kGB);
    const int L = D.L, nl = D.nl, ld = D.ld;
    hipLaunchKernelGGL(gk3_force, grid, block, 0, st_, D);
    hipLaunchKernelGGL(gk3_b, grid, block, 0, st_, D);
    sum_partials(1, 0);
    hipLaunchKernelGGL(gk3_bHb_psi, grid, block, 0, st_, D);
    sum_partials(1, 1);
    if (!want_plain_) {                              // cluster_dogleg: only b, b^T b and b^T H b are needed, a damped solve follows
        IPC_CL_CHK(hipGetLastError());
        IPC_CL_CHK(fetch(4));
        bb = h_scal_[0]; bHb = h_scal_[1]; hh = 0.0; bh = 0.0; info = 1;
        return hipSuccess;
    }
    hipLaunchKernelGGL(gk_scan, dim3(27), dim3(1024), 0, st_, D.ps, 27, L, ld);
    hipLaunchKernelGGL(gk3_assemble, dim3((nl + 63) / 64, nl), dim3(64), 0, st_, D);
    IPC_CL_CHK(chol_solve_device(D.S, D.S + (size_t)(6 * nl + 1) * (6 * nl), 6 * nl, D.rhs, d_info_, st_));
    hipLaunchKernelGGL(gk3_nu, dim3((nl + 63) / 64), dim3(64), 0, st_, D);
    hipLaunchKernelGGL(gk3_events, dim3((L + 2 + kGB - 1) / kGB), block, 0, st_, D);
    hipLaunchKernelGGL(gk_scan, dim3(6), dim3(1024), 0, st_, D.nd, 6, L, ld);
    hipLaunchKernelGGL(gk3_rho, grid, block, 0, st_, D);
    hipLaunchKernelGGL(gk_scan, dim3(3), dim3(1024), 0, st_, D.sc, 3, L, ld);
    hipLaunchKernelGGL(gk3_term, grid, block, 0, st_, D);
    hipLaunchKernelGGL(gk_scan, dim3(3), dim3(1024), 0, st_, D.sc + 3 * (size_t)ld, 3, L, ld);
    hipLaunchKernelGGL(gk3_h, grid, block, 0, st_, D);
    sum_partials(2, 2);
    IPC_CL_CHK(hipGetLastError());
    IPC_CL_CHK(fetch(4));
    std::memcpy(&info, h_scal_ + 12, sizeof(int));
    bb = h_scal_[0]; bHb = h_scal_[1]; hh = h_scal_[2]; bh = h_scal_[3];
    return hipSuccess;
}

inline hipError_t ClusterSolver3::damped_solve(double lambda, bool& ok, double& hh, double& bh, double& bHh, double& hHh)
{
    ClusterDev3& D = dev_;
    const int n = 6 * D.L;
    ok = false;
    IPC_CL_CHK(hipMemsetAsync(d_info_, 0, sizeof(int), st_));
    bool banded = false;
    if (literal_band_min_n >= 0 && n >= literal_band_min_n)
        IPC_CL_CHK(literal_band_damped(*this, lambda, banded, D.sc, d_info_));   // (solution in the scan workspace: 6 ld doubles)
    if (!banded) {
        if (n > kMaxDenseN) return hipSuccess;                   // (no band and too large for the dense store: the solve reports Fail)
        const size_t m = (size_t)(n + 1) * n;
        if (2 * m > capH_) {
            hipFree(d_H_); d_H_ = nullptr; capH_ = 0;
            IPC_CL_CHK(hipMalloc(&d_H_, sizeof(double) * 2 * m));
            capH_ = 2 * m;
        }
        IPC_CL_CHK(hipMemsetAsync(d_H_, 0, sizeof(double) * m, st_));
        hipLaunchKernelGGL(gk3_literal_H<DenseStore>, dim3((D.L + 1 + 63) / 64), dim3(64), 0, st_, D, DenseStore{d_H_, n, 6}, lambda);
        IPC_CL_CHK(chol_solve_device(d_H_, d_H_ + m, n, D.sc, d_info_, st_));
    }
    hipLaunchKernelGGL(gk3_h_from_dense, dim3(nblk_), dim3(kGB), 0, st_, D, (const double*)D.sc);
    sum_partials(2, 2);
    hipLaunchKernelGGL(gk3_quad_bh, dim3(nblk_), dim3(kGB), 0, st_, D);
    sum_partials(2, 4);
    IPC_CL_CHK(hipGetLastError());
    IPC_CL_CHK(fetch(6));
    int info;
    std::memcpy(&info, h_scal_ + 12, sizeof(int));
    hh = h_scal_[2]; bh = h_scal_[3]; bHh = h_scal_[4]; hHh = h_scal_[5];
    ok = info == 0 && hh == hh;
    IPC_CL_CHK(hipMemsetAsync(d_info_, 0, sizeof(int), st_));
    return hipSuccess;
}

inline hipError_t ClusterSolver3::blend(double alpha, double& c, double& bma)
{
    hipLaunchKernelGGL(gk3_blend, dim3(nblk_), dim3(kGB), 0, st_, dev_, alpha);
    sum_partials(2, 4);
    IPC_CL_CHK(fetch(6));
    c = h_scal_[4]; bma = h_scal_[5];
    return hipSuccess;
}

inline hipError_t ClusterSolver3::trial(double p, double q, double& newChi, bool& anyChanged)
{
    hipLaunchKernelGGL(gk3_update, dim3(nblk_), dim3(kGB), 0, st_, dev_, p, q);
    sum_partials(1, 6);
    IPC_CL_CHK(evaluate(dev_.Xn, dev_.en, dev_.len, newChi));
    anyChanged = h_scal_[6] != 0.0;
    return hipSuccess;
}

inline hipError_t ClusterSolver3::max_edge_chi2(double& mx_out)
{
    ClusterDev3& D = dev_;
    hipLaunchKernelGGL(gk3_chi_edges, dim3(nblk_), dim3(kGB), 0, st_, D);
    IPC_CL_CHK(hipGetLastError());
    std::vector<double> local;
    std::vector<double>& chi = chi_host_ ? *chi_host_ : local;
    chi.resize((size_t)D.L + D.nl);
    IPC_CL_CHK(hipMemcpyAsync(chi.data(), D.chi_edges, sizeof(double) * chi.size(), hipMemcpyDeviceToHost, st_));
    IPC_CL_CHK(hipStreamSynchronize(st_));
    double mx = 0.0;
    bool nan = false;
    for (double c : chi) { if (c != c) nan = true; else mx = std::max(mx, c); }
    mx_out = (nan && !(mx > 0.0)) ? std::nan("") : mx;      // (max over the edges that have a number: se2_cell.hpp)
    return hipSuccess;
}

inline hipError_t ClusterSolver3::solve(hipStream_t st, const double* chain, int estride, const double* cand,
                                        int cstride, const double* src, int src_ld, int lo, int hi,
                                        const std::vector<int>& members, const int* from, const int* to,
                                        int iterations, ClusterOut& out, std::vector<double>* chi_host)
{
    const int L = hi - lo, nl = (int)members.size(), NS = 6 * nl;
    IPC_CL_CHK(ensure(L, nl));
    const int ld = L + 2;
    st_ = st; chi_host_ = chi_host;
    ClusterDev3& D = dev_;
    D.chain = chain; D.estride = estride; D.lo = lo; D.L = L; D.nl = nl; D.ld = ld;
    D.cand = cand; D.cstride = cstride;
    {
        double* p = d_edge_;
        auto take = [&](size_t n) { double* q = p; p += n; return q; };
        D.X = take(12 * (size_t)ld); D.Xn = take(12 * (size_t)ld);
        D.e = take(6 * (size_t)ld); D.en = take(6 * (size_t)ld); D.g = take(6 * (size_t)ld); D.m = take(6 * (size_t)ld);
        D.b = take(6 * (size_t)ld); D.h = take(6 * (size_t)ld); D.ps = take(27 * (size_t)ld);
        D.nd = take(6 * (size_t)ld); D.sc = take(6 * (size_t)ld);
        D.chi_edges = take((size_t)ld + nl);
        double* q = d_loop_;
        auto takel = [&](size_t n) { double* r = q; q += n; return r; };
        D.le = takel(6 * (size_t)nl); D.len = takel(6 * (size_t)nl); D.lg = takel(6 * (size_t)nl); D.lm = takel(6 * (size_t)nl);
        D.gam = takel(36 * (size_t)nl); D.nu = takel(6 * (size_t)nl); D.rhs = takel(6 * (size_t)nl);
        D.S = d_S_; D.ldS = NS + 1;
        D.partial = d_partial_; D.scal = d_scal_;
    }
    tab_.build(lo, hi, members, from, to);
    lband_stale = true;
    IPC_CL_CHK(hipMemcpyAsync(d_int_, tab_.host.data(), sizeof(int) * tab_.size(), hipMemcpyHostToDevice, st));
    IPC_CL_CHK(hipStreamSynchronize(st));
    D.lfrom = tab_.lfrom(d_int_); D.lto = tab_.lto(d_int_); D.lcand = tab_.lcand(d_int_);
    D.adj_ptr = tab_.adj_ptr(d_int_); D.adj_item = tab_.adj_item(d_int_);
    D.ev_ptr = tab_.ev_ptr(d_int_); D.ev_item = tab_.ev_item(d_int_);
    IPC_CL_CHK(hipMemcpy2DAsync(D.X, sizeof(double) * ld, src + lo, sizeof(double) * src_ld, sizeof(double) * (L + 1), 12,
                                hipMemcpyDeviceToDevice, st));
    nblk_ = (L + nl + 1 + kGB - 1) / kGB;
    IPC_CL_CHK(hipMemsetAsync(d_info_, 0, sizeof(int), st));
    return cluster_dogleg(*this, iterations, out, term_eps, tab_.L + tab_.nl, allow_damping);
}

}  // namespace ipc
