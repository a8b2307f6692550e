// General SE(2) cluster solve: odometry chain lo..hi + ANY number of loop edges, one problem at a
// time, host-driven dog-leg over small grid-wide kernels.  Used by
//   * the faithful incremental mode  (IPC::agreementCheck, reference src/consensus.cpp:43-75:
//     cluster of the candidate + every transitively overlapping accepted edge), and
//   * the final map optimisation     (reference src/simulation.cpp:50-65: optimize(1000) over the
//     un-scaled odometry + every accepted loop).
// Same formulation as se2_cell.hpp (chain closed form, G = Gamma_l Phi_j, capacitance system),
// but the capacitance matrix is (3 nl) x (3 nl) and is factored by dense_chol.hpp; per-edge state lives in HBM arrays instead of registers, loop ranges are handled with prefix
// sums of Psi_j / w_j, and the dog-leg control flow runs on the host (a few scalars per trial
// cross PCIe).  This mode is sequential by nature -- it is the reference's algorithm, not the
// throughput path.
#pragma once
#include "cluster_common.hpp"
#include "se2_cell.hpp"

namespace ipc {

struct PoseArr { double *x, *y, *th, *c, *s; };

struct ClusterDev {
    const double* chain; int estride; int lo; int L; int nl; int ld;   // ld = L + 2 (row length of per-edge arrays)
    const double* cand; int cstride;
    const int *lfrom, *lto, *lcand;          // [nl] local pose indices / candidate record index
    PoseArr X, Xn;                           // [L+1]
    double *e, *en;                          // [3][ld]  odometry errors (index j = edge j-1 -> j)
    double *le, *len;                        // [3][nl]  loop errors
    double *g, *m;                           // [3][ld]
    double *lg, *lm;                         // [3][nl]
    double *b, *h;                           // [3][ld]
    double *ps;                              // [9][ld]  prefix sums of Psi (6) and w (3)
    double *gam;                             // [9][nl]
    double *S; int ldS;                      // (NS+1) x NS column major, ldS = NS+1; row NS holds the rhs
    double *rhs;                             // [NS] solution mu of the capacitance system
    double *nu;                              // [3][nl]
    double *nd;                              // [3][ld]  event sums -> n_j
    double *sc;                              // [3][ld]  scan workspace
    const int *adj_ptr, *adj_item;           // per pose p: items l*2 + role (0 = from, 1 = to)
    const int *ev_ptr, *ev_item;             // per index j: items l*2 + kind (0 = start, 1 = end)
    double* partial;                         // [nblocks][4]
    double* scal;                            // [8] reduced scalars
    double* chi_edges;                       // [L + nl] per-edge chi2 (output)
};

__device__ __forceinline__ Pose2 gk_pose(const PoseArr& A, int p)
{
    return Pose2{gptr(A.x)[p], gptr(A.y)[p], gptr(A.th)[p], gptr(A.c)[p], gptr(A.s)[p]};
}
__device__ __forceinline__ Sym3 gk_sym(const double* base_generic, int stride, int field0, int idx)
{
    auto base = gptr(base_generic);
    Sym3 s;
    s.a00 = base[(size_t)(field0 + 0) * stride + idx];
    s.a01 = base[(size_t)(field0 + 1) * stride + idx];
    s.a02 = base[(size_t)(field0 + 2) * stride + idx];
    s.a11 = base[(size_t)(field0 + 3) * stride + idx];
    s.a12 = base[(size_t)(field0 + 4) * stride + idx];
    s.a22 = base[(size_t)(field0 + 5) * stride + idx];
    return s;
}

// GK_OPERANDS_FIRST: the per-index bodies below read all their operands into locals first and put a scheduling barrier
// behind the reads.  Left alone the scheduler sinks each load next to its use; with hundreds of waves per CU (the
// one-kernel-per-phase path) that is harmless, but the persistent kernel's leader runs these bodies with two waves per
// SIMD and then pays one L2 round trip per operand instead of one per body.
// errors + chi2 of the poses Y (trial or committed); edges 1..L then loops
__device__ __forceinline__ void gk_eval_at(const ClusterDev& D, const PoseArr& Y, double* eo, double* leo, int i, double (&v)[1])
{
    if (i >= 1 && i <= D.L) {
        const int k = D.lo + i - 1;
        const Pose2 a = gk_pose(Y, i - 1), b = gk_pose(Y, i);
        const double tzx = gptr(D.chain)[(size_t)F_TZX * D.estride + k], tzy = gptr(D.chain)[(size_t)F_TZY * D.estride + k];
        const double cz = gptr(D.chain)[(size_t)F_CZ * D.estride + k], sz = gptr(D.chain)[(size_t)F_SZ * D.estride + k];
        const double thz = gptr(D.chain)[(size_t)F_THZ * D.estride + k];
        const Sym3 om = gk_sym(D.chain, D.estride, F_OM, k);
        __builtin_amdgcn_sched_barrier(0);          // every operand requested before the first is used (see GK_OPERANDS_FIRST)
        double e0, e1, e2;
        se2_error(a, b, tzx, tzy, cz, sz, thz, e0, e1, e2);
        gptr(eo)[i] = e0; gptr(eo)[D.ld + i] = e1; gptr(eo)[2 * D.ld + i] = e2;
        v[0] = om.quad(e0, e1, e2);
    } else if (i > D.L && i <= D.L + D.nl) {
        const int l = i - D.L - 1, c = gptr(D.lcand)[l];
        const Pose2 a = gk_pose(Y, gptr(D.lfrom)[l]), b = gk_pose(Y, gptr(D.lto)[l]);
        double e0, e1, e2;
        se2_error(a, b, gptr(D.cand)[(size_t)F_TZX * D.cstride + c], gptr(D.cand)[(size_t)F_TZY * D.cstride + c],
                  gptr(D.cand)[(size_t)F_CZ * D.cstride + c], gptr(D.cand)[(size_t)F_SZ * D.cstride + c],
                  gptr(D.cand)[(size_t)F_THZ * D.cstride + c], e0, e1, e2);
        gptr(leo)[l] = e0; gptr(leo)[D.nl + l] = e1; gptr(leo)[2 * D.nl + l] = e2;
        v[0] = gk_sym(D.cand, D.cstride, F_OM, c).quad(e0, e1, e2);
    }
}
__global__ void gk_eval(ClusterDev D, PoseArr Y, double* eo, double* leo)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double v[1] = {0.0};
    gk_eval_at(D, Y, eo, leo, i, v);
    gk_block_reduce_store<1>(v, D.partial + blockIdx.x * 4);
}

// per-edge chi2 of the committed state (output for the per-edge threshold test)
__device__ __forceinline__ void gk_chi_edges_at(const ClusterDev& D, int i)
{
    if (i >= 1 && i <= D.L) {
        const int k = D.lo + i - 1;
        gptr(D.chi_edges)[i - 1] = gk_sym(D.chain, D.estride, F_OM, k).quad(gptr(D.e)[i], gptr(D.e)[D.ld + i], gptr(D.e)[2 * D.ld + i]);
    } else if (i > D.L && i <= D.L + D.nl) {
        const int l = i - D.L - 1;
        gptr(D.chi_edges)[D.L + l] = gk_sym(D.cand, D.cstride, F_OM, gptr(D.lcand)[l]).quad(gptr(D.le)[l], gptr(D.le)[D.nl + l], gptr(D.le)[2 * D.nl + l]);
    }
}
__global__ void gk_chi_edges(ClusterDev D)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    gk_chi_edges_at(D, i);
}

// forces g = (P q_t, q_th), hand-backs m = (g_t, g_th + (J dt).g_t), Gamma_l
__device__ __forceinline__ void gk_force_at(const ClusterDev& D, int i)
{
    if (i >= 1 && i <= D.L) {
        const int k = D.lo + i - 1;
        const Pose2 a = gk_pose(D.X, i - 1), b = gk_pose(D.X, i);
        const Sym3 om = gk_sym(D.chain, D.estride, F_OM, k);
        const double e0 = gptr(D.e)[i], e1 = gptr(D.e)[D.ld + i], e2 = gptr(D.e)[2 * D.ld + i];
        const double cz = gptr(D.chain)[(size_t)F_CZ * D.estride + k], sz = gptr(D.chain)[(size_t)F_SZ * D.estride + k];
        __builtin_amdgcn_sched_barrier(0);          // every operand requested before the first is used (see GK_OPERANDS_FIRST)
        double q0, q1, q2;
        om.mul(e0, e1, e2, q0, q1, q2);
        const double cP = a.c * cz - a.s * sz, sP = a.s * cz + a.c * sz;
        const double gx = cP * q0 - sP * q1, gy = sP * q0 + cP * q1;
        const double dx = b.x - a.x, dy = b.y - a.y;
        gptr(D.g)[i] = gx; gptr(D.g)[D.ld + i] = gy; gptr(D.g)[2 * D.ld + i] = q2;
        gptr(D.m)[i] = gx; gptr(D.m)[D.ld + i] = gy; gptr(D.m)[2 * D.ld + i] = q2 + (-dy * gx + dx * gy);
    } else if (i > D.L && i <= D.L + D.nl) {
        const int l = i - D.L - 1, c = gptr(D.lcand)[l];
        const Pose2 a = gk_pose(D.X, gptr(D.lfrom)[l]), b = gk_pose(D.X, gptr(D.lto)[l]);
        double q0, q1, q2;
        gk_sym(D.cand, D.cstride, F_OM, c).mul(gptr(D.le)[l], gptr(D.le)[D.nl + l], gptr(D.le)[2 * D.nl + l], q0, q1, q2);
        const double cz = gptr(D.cand)[(size_t)F_CZ * D.cstride + c], sz = gptr(D.cand)[(size_t)F_SZ * D.cstride + c];
        const double cP = a.c * cz - a.s * sz, sP = a.s * cz + a.c * sz;
        const double gx = cP * q0 - sP * q1, gy = sP * q0 + cP * q1;
        const double dx = b.x - a.x, dy = b.y - a.y;
        gptr(D.lg)[l] = gx; gptr(D.lg)[D.nl + l] = gy; gptr(D.lg)[2 * D.nl + l] = q2;
        gptr(D.lm)[l] = gx; gptr(D.lm)[D.nl + l] = gy; gptr(D.lm)[2 * D.nl + l] = q2 + (-dy * gx + dx * gy);
        // Gamma_l = sigma [[Lam, Lam K],[0,1]],  Lam = R(-(th_f + thz)),  K = J (t_to - o), o = pose 0
        const double A = cP, B = sP;          // cos / sin of (th_f + thz)
        const double Kx = -(b.y - gptr(D.X.y)[0]), Ky = b.x - gptr(D.X.x)[0];
        const double sg = gptr(D.lto)[l] > gptr(D.lfrom)[l] ? 1.0 : -1.0;
        auto G = gptr(D.gam);
        G[0 * D.nl + l] = sg * A;  G[1 * D.nl + l] = sg * B; G[2 * D.nl + l] = sg * (A * Kx + B * Ky);
        G[3 * D.nl + l] = -sg * B; G[4 * D.nl + l] = sg * A; G[5 * D.nl + l] = sg * (-B * Kx + A * Ky);
        G[6 * D.nl + l] = 0.0;     G[7 * D.nl + l] = 0.0;    G[8 * D.nl + l] = sg;
    }
}
__global__ void gk_force(ClusterDev D)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    gk_force_at(D, i);
}

// b_j = m_{j+1} - g_j + loop terms; partial b^T b
__device__ __forceinline__ void gk_b_at(const ClusterDev& D, int j, double (&v)[1])
{
    if (j >= 1 && j <= D.L) {
        double b0 = -gptr(D.g)[j], b1 = -gptr(D.g)[D.ld + j], b2 = -gptr(D.g)[2 * D.ld + j];
        if (j < D.L) { b0 += gptr(D.m)[j + 1]; b1 += gptr(D.m)[D.ld + j + 1]; b2 += gptr(D.m)[2 * D.ld + j + 1]; }
        for (int q = gptr(D.adj_ptr)[j]; q < gptr(D.adj_ptr)[j + 1]; ++q) {
            const int it = gptr(D.adj_item)[q], l = it >> 1;
            if (it & 1) { b0 -= gptr(D.lg)[l]; b1 -= gptr(D.lg)[D.nl + l]; b2 -= gptr(D.lg)[2 * D.nl + l]; }
            else        { b0 += gptr(D.lm)[l]; b1 += gptr(D.lm)[D.nl + l]; b2 += gptr(D.lm)[2 * D.nl + l]; }
        }
        gptr(D.b)[j] = b0; gptr(D.b)[D.ld + j] = b1; gptr(D.b)[2 * D.ld + j] = b2;
        v[0] = b0 * b0 + b1 * b1 + b2 * b2;
    }
}
__global__ void gk_b(ClusterDev D)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    double v[1] = {0.0};
    gk_b_at(D, j, v);
    gk_block_reduce_store<1>(v, D.partial + blockIdx.x * 4);
}

// partial b^T H b = sum_e |J_e b|^2_Om ; also Psi_j / w_j (before the prefix sums)
__device__ __forceinline__ void gk_bHb_psi_at(const ClusterDev& D, int i, double (&v)[1])
{
    if (i >= 1 && i <= D.L) {
        const int k = D.lo + i - 1;
        const Pose2 a = gk_pose(D.X, i - 1), b = gk_pose(D.X, i);
        const double cz = gptr(D.chain)[(size_t)F_CZ * D.estride + k], sz = gptr(D.chain)[(size_t)F_SZ * D.estride + k];
        const int ia = i > 1 ? i - 1 : i;               // (row 0 of b carries nothing: read a valid slot, select afterwards)
        const double ra0 = gptr(D.b)[ia], ra1 = gptr(D.b)[D.ld + ia], ra2 = gptr(D.b)[2 * D.ld + ia];
        const double vb0 = gptr(D.b)[i], vb1 = gptr(D.b)[D.ld + i], vb2 = gptr(D.b)[2 * D.ld + i];
        const Sym3 om = gk_sym(D.chain, D.estride, F_OM, k), sg = gk_sym(D.chain, D.estride, F_SG, k);
        const double ox = gptr(D.X.x)[0], oy = gptr(D.X.y)[0];
        const double ee0 = gptr(D.e)[i], ee1 = gptr(D.e)[D.ld + i], ee2 = gptr(D.e)[2 * D.ld + i];
        __builtin_amdgcn_sched_barrier(0);          // every operand requested before the first is used (see GK_OPERANDS_FIRST)
        const double va0 = i > 1 ? ra0 : 0.0, va1 = i > 1 ? ra1 : 0.0, va2 = i > 1 ? ra2 : 0.0;
        double wx, wy, wth;
        se2_apply_J(a, b, cz, sz, va0, va1, va2, vb0, vb1, vb2, wx, wy, wth);
        v[0] = om.quad(wx, wy, wth);
        // Psi_j = Phi Cov Phi^T, w_j = Phi e_j,  Phi = [[P, -kappa],[0,1]], kappa = J (t_j - o)
        const double c = a.c * cz - a.s * sz, sn = a.s * cz + a.c * sz;
        const double kx = -(b.y - oy), ky = b.x - ox;
        const double cc = c * c, ss = sn * sn, cs = c * sn;
        const double C00 = cc * sg.a00 - 2 * cs * sg.a01 + ss * sg.a11;
        const double C01 = cs * (sg.a00 - sg.a11) + (cc - ss) * sg.a01;
        const double C11 = ss * sg.a00 + 2 * cs * sg.a01 + cc * sg.a11;
        const double c0 = c * sg.a02 - sn * sg.a12, c1 = sn * sg.a02 + c * sg.a12;
        const double sth = sg.a22;
        const double p02 = c0 - sth * kx, p12 = c1 - sth * ky;
        gptr(D.ps)[0 * D.ld + i] = C00 - kx * c0 - kx * p02;
        gptr(D.ps)[1 * D.ld + i] = C01 - kx * c1 - ky * p02;
        gptr(D.ps)[2 * D.ld + i] = p02;
        gptr(D.ps)[3 * D.ld + i] = C11 - ky * c1 - ky * p12;
        gptr(D.ps)[4 * D.ld + i] = p12;
        gptr(D.ps)[5 * D.ld + i] = sth;
        gptr(D.ps)[6 * D.ld + i] = c * ee0 - sn * ee1 - kx * ee2;
        gptr(D.ps)[7 * D.ld + i] = sn * ee0 + c * ee1 - ky * ee2;
        gptr(D.ps)[8 * D.ld + i] = ee2;
    } else if (i > D.L && i <= D.L + D.nl) {
        const int l = i - D.L - 1, c = gptr(D.lcand)[l];
        const int f = gptr(D.lfrom)[l], t = gptr(D.lto)[l];
        const Pose2 a = gk_pose(D.X, f), b = gk_pose(D.X, t);
        const double va0 = f > 0 ? gptr(D.b)[f] : 0.0, va1 = f > 0 ? gptr(D.b)[D.ld + f] : 0.0, va2 = f > 0 ? gptr(D.b)[2 * D.ld + f] : 0.0;
        const double vb0 = t > 0 ? gptr(D.b)[t] : 0.0, vb1 = t > 0 ? gptr(D.b)[D.ld + t] : 0.0, vb2 = t > 0 ? gptr(D.b)[2 * D.ld + t] : 0.0;
        double wx, wy, wth;
        se2_apply_J(a, b, gptr(D.cand)[(size_t)F_CZ * D.cstride + c], gptr(D.cand)[(size_t)F_SZ * D.cstride + c], va0, va1, va2,
                    vb0, vb1, vb2, wx, wy, wth);
        v[0] = gk_sym(D.cand, D.cstride, F_OM, c).quad(wx, wy, wth);
    } else if (i == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) gptr(D.ps)[k * D.ld] = 0.0;
    }
}
__global__ void gk_bHb_psi(ClusterDev D)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double v[1] = {0.0};
    gk_bHb_psi_at(D, i, v);
    gk_block_reduce_store<1>(v, D.partial + blockIdx.x * 4);
}

// capacitance system: S (lower triangle) and rhs.  put(row, col, value) stores one entry of the lower triangle,
// put_rhs(col, value) one entry of the right-hand-side row: the dense column-major array (gk_assemble_at) or the banded
// layout of cluster_band.hpp.
template <class Put, class PutRhs>
__device__ __forceinline__ void gk_assemble_core(const ClusterDev& D, int l1, int l2, Put put, PutRhs put_rhs)     // row block l1, column block l2
{
    if (l2 >= D.nl || l1 >= D.nl || l2 > l1) return;
    const int lo1 = min(gptr(D.lfrom)[l1], gptr(D.lto)[l1]), hi1 = max(gptr(D.lfrom)[l1], gptr(D.lto)[l1]);
    const int lo2 = min(gptr(D.lfrom)[l2], gptr(D.lto)[l2]), hi2 = max(gptr(D.lfrom)[l2], gptr(D.lto)[l2]);
    const int a = max(lo1, lo2), bq = min(hi1, hi2);
    double Mm[6] = {0, 0, 0, 0, 0, 0};
    if (bq > a) {
#pragma unroll
        for (int k = 0; k < 6; ++k) Mm[k] = gptr(D.ps)[k * D.ld + bq] - gptr(D.ps)[k * D.ld + a];
    }
    double G1[3][3], G2[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) { G1[r][c] = gptr(D.gam)[(3 * r + c) * D.nl + l1]; G2[r][c] = gptr(D.gam)[(3 * r + c) * D.nl + l2]; }
    const double M3[3][3] = {{Mm[0], Mm[1], Mm[2]}, {Mm[1], Mm[3], Mm[4]}, {Mm[2], Mm[4], Mm[5]}};
    double GM[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) GM[r][c] = G1[r][0] * M3[0][c] + G1[r][1] * M3[1][c] + G1[r][2] * M3[2][c];
    const Sym3 sgl = gk_sym(D.cand, D.cstride, F_SG, gptr(D.lcand)[l1]);
    const double Sg[3][3] = {{sgl.a00, sgl.a01, sgl.a02}, {sgl.a01, sgl.a11, sgl.a12}, {sgl.a02, sgl.a12, sgl.a22}};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double t = GM[r][0] * G2[c][0] + GM[r][1] * G2[c][1] + GM[r][2] * G2[c][2];
            if (l1 == l2) t += Sg[r][c];
            put(3 * l1 + r, 3 * l2 + c, t);
        }
    if (l1 == l2) {
        const double W0 = gptr(D.ps)[6 * D.ld + hi1] - gptr(D.ps)[6 * D.ld + lo1];
        const double W1 = gptr(D.ps)[7 * D.ld + hi1] - gptr(D.ps)[7 * D.ld + lo1];
        const double W2 = gptr(D.ps)[8 * D.ld + hi1] - gptr(D.ps)[8 * D.ld + lo1];
#pragma unroll
        for (int r = 0; r < 3; ++r)
            put_rhs(3 * l1 + r, gptr(D.le)[r * D.nl + l1] - (G1[r][0] * W0 + G1[r][1] * W1 + G1[r][2] * W2));
    }
}
__device__ __forceinline__ void gk_assemble_at(const ClusterDev& D, int l1, int l2)
{
    const int NS = 3 * D.nl;
    gk_assemble_core(D, l1, l2, [&](int row, int col, double v) { st_shared(&D.S[(size_t)col * D.ldS + row], v); },
                     [&](int col, double v) { st_shared(&D.S[(size_t)col * D.ldS + NS], v); });
}
__global__ void gk_assemble(ClusterDev D)
{
    gk_assemble_at(D, blockIdx.y, blockIdx.x * blockDim.x + threadIdx.x);
}

// nu_l = Gamma_l^T mu_l
__device__ __forceinline__ void gk_nu_at(const ClusterDev& D, int l)
{
    if (l >= D.nl) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < 3; ++r) t += gptr(D.gam)[(3 * r + c) * D.nl + l] * gptr(D.rhs)[3 * l + r];
        gptr(D.nu)[c * D.nl + l] = t;
    }
}
__global__ void gk_nu(ClusterDev D)
{
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    gk_nu_at(D, l);
}

// nd[j] = sum of the events at index j (+nu at the start of a loop range, -nu one past its end)
__device__ __forceinline__ void gk_events_at(const ClusterDev& D, int j)
{
    if (j > D.L + 1) return;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    if (j >= 1) {
        for (int q = gptr(D.ev_ptr)[j]; q < gptr(D.ev_ptr)[j + 1]; ++q) {
            const int it = gptr(D.ev_item)[q], l = it >> 1;
            const double sgn = (it & 1) ? -1.0 : 1.0;
            a0 += sgn * gptr(D.nu)[l]; a1 += sgn * gptr(D.nu)[D.nl + l]; a2 += sgn * gptr(D.nu)[2 * D.nl + l];
        }
    }
    gptr(D.nd)[j] = a0; gptr(D.nd)[D.ld + j] = a1; gptr(D.nd)[2 * D.ld + j] = a2;
}
__global__ void gk_events(ClusterDev D)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    gk_events_at(D, j);
}

// u_j = -Cov Phi^T n_j - e_j ; rho = (P u_t, u_th): rho_th -> sc[0], rho_t -> sc[1], sc[2]
__device__ __forceinline__ void gk_rho_at(const ClusterDev& D, int i)
{
    if (i < 1 || i > D.L) return;
    const int k = D.lo + i - 1;
    const Pose2 a = gk_pose(D.X, i - 1), b = gk_pose(D.X, i);
    const double cz = gptr(D.chain)[(size_t)F_CZ * D.estride + k], sz = gptr(D.chain)[(size_t)F_SZ * D.estride + k];
    const double ox = gptr(D.X.x)[0], oy = gptr(D.X.y)[0];
    const double n0 = gptr(D.nd)[i], n1 = gptr(D.nd)[D.ld + i], n2 = gptr(D.nd)[2 * D.ld + i];
    const Sym3 sg = gk_sym(D.chain, D.estride, F_SG, k);
    const double e0 = gptr(D.e)[i], e1 = gptr(D.e)[D.ld + i], e2 = gptr(D.e)[2 * D.ld + i];
    __builtin_amdgcn_sched_barrier(0);              // GK_OPERANDS_FIRST
    const double c = a.c * cz - a.s * sz, sn = a.s * cz + a.c * sz;
    const double kx = -(b.y - oy), ky = b.x - ox;
    const double wx = c * n0 + sn * n1, wy = -sn * n0 + c * n1, wth = -(kx * n0 + ky * n1) + n2;
    double vx, vy, vth;
    sg.mul(wx, wy, wth, vx, vy, vth);
    const double ux = -vx - e0, uy = -vy - e1, uth = -vth - e2;
    gptr(D.sc)[i] = uth;
    gptr(D.sc)[D.ld + i] = c * ux - sn * uy;
    gptr(D.sc)[2 * D.ld + i] = sn * ux + c * uy;
}
__global__ void gk_rho(ClusterDev D)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    gk_rho_at(D, i);
}
// term = rho_t + J dt h_theta(j-1)  (sc[0] already holds the inclusive theta prefix)
__device__ __forceinline__ void gk_term_at(const ClusterDev& D, int i)
{
    if (i < 1 || i > D.L) return;
    const double thPrev = i > 1 ? gptr(D.sc)[i - 1] : 0.0;
    const double dx = gptr(D.X.x)[i] - gptr(D.X.x)[i - 1], dy = gptr(D.X.y)[i] - gptr(D.X.y)[i - 1];
    gptr(D.sc)[D.ld + i] += -dy * thPrev;
    gptr(D.sc)[2 * D.ld + i] += dx * thPrev;
}
__global__ void gk_term(ClusterDev D)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    gk_term_at(D, i);
}
// h = (sc1, sc2, sc0); partial |h|^2, b.h
__device__ __forceinline__ void gk_h_at(const ClusterDev& D, int i, double (&v)[2])
{
    if (i >= 1 && i <= D.L) {
        const double h0 = gptr(D.sc)[D.ld + i], h1 = gptr(D.sc)[2 * D.ld + i], h2 = gptr(D.sc)[i];
        gptr(D.h)[i] = h0; gptr(D.h)[D.ld + i] = h1; gptr(D.h)[2 * D.ld + i] = h2;
        v[0] = h0 * h0 + h1 * h1 + h2 * h2;
        v[1] = gptr(D.b)[i] * h0 + gptr(D.b)[D.ld + i] * h1 + gptr(D.b)[2 * D.ld + i] * h2;
    }
}
__global__ void gk_h(ClusterDev D)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double v[2] = {0.0, 0.0};
    gk_h_at(D, i, v);
    gk_block_reduce_store<2>(v, D.partial + blockIdx.x * 4);
}
// c = hsd.(hgn - hsd), |hgn - hsd|^2 for the dog-leg blend
__device__ __forceinline__ void gk_blend_at(const ClusterDev& D, double alpha, int i, double (&v)[2])
{
    if (i >= 1 && i <= D.L) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double sk = alpha * gptr(D.b)[k * D.ld + i], ak = gptr(D.h)[k * D.ld + i] - sk;
            v[0] += sk * ak;
            v[1] += ak * ak;
        }
    }
}
__global__ void gk_blend(ClusterDev D, double alpha)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double v[2] = {0.0, 0.0};
    gk_blend_at(D, alpha, i, v);
    gk_block_reduce_store<2>(v, D.partial + blockIdx.x * 4);
}
// trial poses Xn = X (+) (p b + q h); partial "changed" count
__device__ __forceinline__ void gk_update_at(const ClusterDev& D, double p, double q, int i, double (&v)[1])
{
    if (i == 0) { gptr(D.Xn.x)[0] = gptr(D.X.x)[0]; gptr(D.Xn.y)[0] = gptr(D.X.y)[0]; gptr(D.Xn.th)[0] = gptr(D.X.th)[0]; gptr(D.Xn.c)[0] = gptr(D.X.c)[0]; gptr(D.Xn.s)[0] = gptr(D.X.s)[0]; }
    if (i >= 1 && i <= D.L) {
        const double x0 = gptr(D.X.x)[i], y0 = gptr(D.X.y)[i], th0 = gptr(D.X.th)[i];
        const double b0 = gptr(D.b)[i], b1 = gptr(D.b)[D.ld + i], b2 = gptr(D.b)[2 * D.ld + i];
        const double h0 = gptr(D.h)[i], h1 = gptr(D.h)[D.ld + i], h2 = gptr(D.h)[2 * D.ld + i];
        __builtin_amdgcn_sched_barrier(0);              // GK_OPERANDS_FIRST
        const double x = x0 + fma(p, b0, q * h0);
        const double y = y0 + fma(p, b1, q * h1);
        const double th = wrap_pi(th0 + fma(p, b2, q * h2));
        double s, c;
        sincos_pi(th, s, c);
        gptr(D.Xn.x)[i] = x; gptr(D.Xn.y)[i] = y; gptr(D.Xn.th)[i] = th; gptr(D.Xn.c)[i] = c; gptr(D.Xn.s)[i] = s;
        v[0] = (x != x0 || y != y0 || th != th0) ? 1.0 : 0.0;
    }
}
__global__ void gk_update(ClusterDev D, double p, double q)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double v[1] = {0.0};
    gk_update_at(D, p, q, i, v);
    gk_block_reduce_store<1>(v, D.partial + blockIdx.x * 4);
}

// ---- literal normal equations (Levenberg retry, cluster_common.hpp::cluster_dogleg) ---------------------------
// g2o's own system for the sub-graph: H = sum_e J_e^T Om_e J_e over the chain edges 1..L and the loops, poses 1..L
// free (pose 0 is the gauge), + lambda on the diagonal; Hd is the (n+1) x n column-major array dense_chol.hpp factors
// (n = 3 L, lower triangle, right-hand side b in row n).  One thread per pose builds its diagonal block and the
// off-diagonal blocks it is the LATER end of, in a fixed order (no atomics: same bits every run).
__device__ __forceinline__ void gk_edge_J(const Pose2& a, const Pose2& b, double cz, double sz, double (&Ja)[3][3], double (&Jb)[3][3])
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double u0 = k == 0 ? 1.0 : 0.0, u1 = k == 1 ? 1.0 : 0.0, u2 = k == 2 ? 1.0 : 0.0;
        se2_apply_J(a, b, cz, sz, u0, u1, u2, 0.0, 0.0, 0.0, Ja[0][k], Ja[1][k], Ja[2][k]);
        se2_apply_J(a, b, cz, sz, 0.0, 0.0, 0.0, u0, u1, u2, Jb[0][k], Jb[1][k], Jb[2][k]);
    }
}
// out += A^T Om B
__device__ __forceinline__ void gk_atob(const double (&A)[3][3], const Sym3& om, const double (&B)[3][3], double (&out)[3][3])
{
    double OB[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) om.mul(B[0][c], B[1][c], B[2][c], OB[0][c], OB[1][c], OB[2][c]);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) out[r][c] += A[0][r] * OB[0][c] + A[1][r] * OB[1][c] + A[2][r] * OB[2][c];
}
// St: DenseStore or BandStore (cluster_common.hpp) -- where entry (i, j), i >= j, of the lower triangle lives
template <class St>
__global__ void gk_literal_H(ClusterDev D, St S, double lambda)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < 1 || p > D.L) return;
    double dg[3][3] = {{lambda, 0, 0}, {0, lambda, 0}, {0, 0, lambda}};
    auto put = [&](int prow, int pcol, const double (&B)[3][3], bool add) {      // block (prow, pcol), prow > pcol >= 1
        const int ui0 = S.unk(prow), uj0 = S.unk(pcol);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                // (a banded store orders the border poses last: the block may sit above the diagonal there, and the lower
                // triangle holds its transpose)
                const int ui = ui0 + r, uj = uj0 + c;
                double* q = ui >= uj ? S.lower(ui, uj) : S.lower(uj, ui);
                *q = add ? *q + B[r][c] : B[r][c];
            }
    };
    {   // chain edge p (poses p-1 -> p): this pose is its second end
        const int k = D.lo + p - 1;
        double Ja[3][3], Jb[3][3];
        gk_edge_J(gk_pose(D.X, p - 1), gk_pose(D.X, p), D.chain[(size_t)F_CZ * D.estride + k], D.chain[(size_t)F_SZ * D.estride + k], Ja, Jb);
        const Sym3 om = gk_sym(D.chain, D.estride, F_OM, k);
        gk_atob(Jb, om, Jb, dg);
        if (p >= 2) { double off[3][3] = {}; gk_atob(Jb, om, Ja, off); put(p, p - 1, off, false); }
    }
    if (p < D.L) {   // chain edge p+1 (poses p -> p+1): first end
        const int k = D.lo + p;
        double Ja[3][3], Jb[3][3];
        gk_edge_J(gk_pose(D.X, p), gk_pose(D.X, p + 1), D.chain[(size_t)F_CZ * D.estride + k], D.chain[(size_t)F_SZ * D.estride + k], Ja, Jb);
        gk_atob(Ja, gk_sym(D.chain, D.estride, F_OM, k), Ja, dg);
    }
    for (int q = D.adj_ptr[p]; q < D.adj_ptr[p + 1]; ++q) {
        const int it = D.adj_item[q], l = it >> 1, c = D.lcand[l];
        const int f = D.lfrom[l], t = D.lto[l];
        double Ja[3][3], Jb[3][3];
        gk_edge_J(gk_pose(D.X, f), gk_pose(D.X, t), D.cand[(size_t)F_CZ * D.cstride + c], D.cand[(size_t)F_SZ * D.cstride + c], Ja, Jb);
        const Sym3 om = gk_sym(D.cand, D.cstride, F_OM, c);
        if (it & 1) gk_atob(Jb, om, Jb, dg); else gk_atob(Ja, om, Ja, dg);
        const int other = (it & 1) ? f : t;
        if (other >= 1 && other < p) {                // this pose is the later end: the off-diagonal block is its job
            double off[3][3] = {};
            if (it & 1) gk_atob(Jb, om, Ja, off); else gk_atob(Ja, om, Jb, off);
            put(p, other, off, true);                  // (several loops may join the same pair: accumulated in list order)
        }
    }
    const int u0 = S.unk(p);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c <= r; ++c) *S.lower(u0 + r, u0 + c) = dg[r][c];
        *S.lower(S.n(), u0 + r) = D.b[r * D.ld + p];
    }
}
// h <- the dense solution; partial |h|^2, b.h
__global__ void gk_h_from_dense(ClusterDev D, const double* x)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double v[2] = {0.0, 0.0};
    if (i >= 1 && i <= D.L) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double hk = x[3 * (i - 1) + k];
            D.h[k * D.ld + i] = hk;
            v[0] += hk * hk;
            v[1] += D.b[k * D.ld + i] * hk;
        }
    }
    gk_block_reduce_store<2>(v, D.partial + blockIdx.x * 4);
}
// partial b^T H h and h^T H h over the edges (H without lambda)
__global__ void gk_quad_bh(ClusterDev D)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double v[2] = {0.0, 0.0};
    auto vec = [&](const double* a, int p, double& x0, double& x1, double& x2) {
        x0 = p > 0 ? a[p] : 0.0; x1 = p > 0 ? a[D.ld + p] : 0.0; x2 = p > 0 ? a[2 * D.ld + p] : 0.0;
    };
    int f = -1, t = -1;
    double cz = 0, sz = 0;
    Sym3 om{};
    if (i >= 1 && i <= D.L) {
        const int k = D.lo + i - 1;
        f = i - 1; t = i;
        cz = D.chain[(size_t)F_CZ * D.estride + k]; sz = D.chain[(size_t)F_SZ * D.estride + k];
        om = gk_sym(D.chain, D.estride, F_OM, k);
    } else if (i > D.L && i <= D.L + D.nl) {
        const int l = i - D.L - 1, c = D.lcand[l];
        f = D.lfrom[l]; t = D.lto[l];
        cz = D.cand[(size_t)F_CZ * D.cstride + c]; sz = D.cand[(size_t)F_SZ * D.cstride + c];
        om = gk_sym(D.cand, D.cstride, F_OM, c);
    }
    if (f >= 0) {
        const Pose2 a = gk_pose(D.X, f), b = gk_pose(D.X, t);
        double a0, a1, a2, b0, b1, b2, wb0, wb1, wb2, wh0, wh1, wh2;
        vec(D.b, f, a0, a1, a2); vec(D.b, t, b0, b1, b2);
        se2_apply_J(a, b, cz, sz, a0, a1, a2, b0, b1, b2, wb0, wb1, wb2);
        vec(D.h, f, a0, a1, a2); vec(D.h, t, b0, b1, b2);
        se2_apply_J(a, b, cz, sz, a0, a1, a2, b0, b1, b2, wh0, wh1, wh2);
        double o0, o1, o2;
        om.mul(wh0, wh1, wh2, o0, o1, o2);
        v[0] = wb0 * o0 + wb1 * o1 + wb2 * o2;
        v[1] = wh0 * o0 + wh1 * o1 + wh2 * o2;
    }
    gk_block_reduce_store<2>(v, D.partial + blockIdx.x * 4);
}

// ------------------------------------------------------------------------------------------
// host driver
// ------------------------------------------------------------------------------------------
class ClusterSolver2 {
public:
    ~ClusterSolver2() { release(); }
    double term_eps = 0.0;             // convergence shortcut of the trial loop (Se2View::term_eps)

    // Solve chain lo..hi (records `chain`) + the loops `members` (indices into the candidate
    // records; ids in from/to are global vertex ids), starting from the poses `src` (global
    // indexing).  The optimised poses stay in result() (local indexing 0..hi-lo).
    // chi_host (optional): per-edge chi2, odometry lo..hi-1 first, then the loops in `members` order.
    hipError_t solve(hipStream_t st, const double* chain, int estride, const double* cand, int cstride,
                     const PoseArr& src, int lo, int hi, const std::vector<int>& members, const int* from,
                     const int* to, int iterations, ClusterOut& out, std::vector<double>* chi_host);
    const PoseArr& result() const { return dev_.X; }

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
    static constexpr int kD = 3;
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
        hipLaunchKernelGGL(gk_literal_H<St>, dim3((D.L + 1 + kGB - 1) / kGB), dim3(kGB), 0, st_, D, S, lambda);
    }

private:
    double* d_H_ = nullptr; size_t capH_ = 0;       // dense system + factor of damped_solve
    ClusterDev dev_{};
    hipStream_t st_ = nullptr;
    int nblk_ = 1;
    std::vector<double>* chi_host_ = nullptr;
    int capL_ = 0, capNl_ = 0;
    double *d_edge_ = nullptr, *d_loop_ = nullptr, *d_S_ = nullptr, *d_partial_ = nullptr, *d_scal_ = nullptr;
    int *d_int_ = nullptr, *d_info_ = nullptr;
    double* h_scal_ = nullptr;          // pinned [12] + info
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
    hipError_t evaluate(const PoseArr& Y, double* eo, double* leo, double& chi)
    {
        hipLaunchKernelGGL(gk_eval, dim3(nblk_), dim3(kGB), 0, st_, dev_, Y, eo, leo);
        sum_partials(1, 7);
        IPC_CL_CHK(fetch(8));
        chi = h_scal_[7];
        return hipSuccess;
    }
};

inline hipError_t ClusterSolver2::ensure(int L, int nl)
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
        IPC_CL_CHK(hipMalloc(&d_edge_, sizeof(double) * (43 * ld + ld + nN)));
        IPC_CL_CHK(hipMalloc(&d_loop_, sizeof(double) * (27 * (size_t)nN + 8)));
        IPC_CL_CHK(hipMalloc(&d_S_, sizeof(double) * 2 * (3 * (size_t)nN + 1) * (3 * (size_t)nN)));   // system + factor
        IPC_CL_CHK(hipMalloc(&d_partial_, sizeof(double) * 4 * ((nL + nN + 1 + kGB) / kGB + 1)));
        IPC_CL_CHK(hipMalloc(&d_int_, sizeof(int) * LoopTables::capacity(nL, nN)));
        capL_ = nL; capNl_ = nN;
    }
    return hipSuccess;
}

inline hipError_t ClusterSolver2::linearize(double& bb, double& bHb, double& hh, double& bh, int& info)
{
    ClusterDev& D = dev_;
    const dim3 grid(nblk_), block(kGB);
    const int L = D.L, nl = D.nl, ld = D.ld;
    hipLaunchKernelGGL(gk_force, grid, block, 0, st_, D);
    hipLaunchKernelGGL(gk_b, grid, block, 0, st_, D);
    sum_partials(1, 0);
    hipLaunchKernelGGL(gk_bHb_psi, grid, block, 0, st_, D);
    sum_partials(1, 1);
    if (!want_plain_) {                              // cluster_dogleg: only b, b^T b and b^T H b are needed, a damped solve follows
        IPC_CL_CHK(hipGetLastError());
        IPC_CL_CHK(fetch(4));
        bb = h_scal_[0]; bHb = h_scal_[1]; hh = 0.0; bh = 0.0; info = 1;
        return hipSuccess;
    }
    hipLaunchKernelGGL(gk_scan, dim3(9), dim3(1024), 0, st_, D.ps, 9, L, ld);
    hipLaunchKernelGGL(gk_assemble, dim3((nl + 63) / 64, nl), dim3(64), 0, st_, D);
    IPC_CL_CHK(chol_solve_device(D.S, D.S + (size_t)(3 * nl + 1) * (3 * nl), 3 * nl, D.rhs, d_info_, st_));
    hipLaunchKernelGGL(gk_nu, dim3((nl + 63) / 64), dim3(64), 0, st_, D);
    hipLaunchKernelGGL(gk_events, dim3((L + 2 + kGB - 1) / kGB), block, 0, st_, D);
    hipLaunchKernelGGL(gk_scan, dim3(3), dim3(1024), 0, st_, D.nd, 3, L, ld);
    hipLaunchKernelGGL(gk_rho, grid, block, 0, st_, D);
    hipLaunchKernelGGL(gk_scan, dim3(1), dim3(1024), 0, st_, D.sc, 1, L, ld);
    hipLaunchKernelGGL(gk_term, grid, block, 0, st_, D);
    hipLaunchKernelGGL(gk_scan, dim3(2), dim3(1024), 0, st_, D.sc + ld, 2, L, ld);
    hipLaunchKernelGGL(gk_h, grid, block, 0, st_, D);
    sum_partials(2, 2);
    IPC_CL_CHK(hipGetLastError());
    IPC_CL_CHK(fetch(4));
    std::memcpy(&info, h_scal_ + 12, sizeof(int));
    bb = h_scal_[0]; bHb = h_scal_[1]; hh = h_scal_[2]; bh = h_scal_[3];
    return hipSuccess;
}

inline hipError_t ClusterSolver2::damped_solve(double lambda, bool& ok, double& hh, double& bh, double& bHh, double& hHh)
{
    ClusterDev& D = dev_;
    const int n = 3 * D.L;
    ok = false;
    IPC_CL_CHK(hipMemsetAsync(d_info_, 0, sizeof(int), st_));
    bool banded = false;
    if (literal_band_min_n >= 0 && n >= literal_band_min_n)
        IPC_CL_CHK(literal_band_damped(*this, lambda, banded, D.sc, d_info_));   // (solution in the scan workspace: 3 ld doubles)
    if (!banded) {
        if (n > kMaxDenseN) return hipSuccess;                   // (no band and too large for the dense store: the solve reports Fail)
        const size_t m = (size_t)(n + 1) * n;
        if (2 * m > capH_) {
            hipFree(d_H_); d_H_ = nullptr; capH_ = 0;
            IPC_CL_CHK(hipMalloc(&d_H_, sizeof(double) * 2 * m));
            capH_ = 2 * m;
        }
        IPC_CL_CHK(hipMemsetAsync(d_H_, 0, sizeof(double) * m, st_));
        hipLaunchKernelGGL(gk_literal_H<DenseStore>, dim3((D.L + 1 + kGB - 1) / kGB), dim3(kGB), 0, st_, D, DenseStore{d_H_, n, 3}, lambda);
        IPC_CL_CHK(chol_solve_device(d_H_, d_H_ + m, n, D.sc, d_info_, st_));
    }
    hipLaunchKernelGGL(gk_h_from_dense, dim3(nblk_), dim3(kGB), 0, st_, D, (const double*)D.sc);
    sum_partials(2, 2);
    hipLaunchKernelGGL(gk_quad_bh, dim3(nblk_), dim3(kGB), 0, st_, D);
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

inline hipError_t ClusterSolver2::blend(double alpha, double& c, double& bma)
{
    hipLaunchKernelGGL(gk_blend, dim3(nblk_), dim3(kGB), 0, st_, dev_, alpha);
    sum_partials(2, 4);
    IPC_CL_CHK(fetch(6));
    c = h_scal_[4]; bma = h_scal_[5];
    return hipSuccess;
}

inline hipError_t ClusterSolver2::trial(double p, double q, double& newChi, bool& anyChanged)
{
    hipLaunchKernelGGL(gk_update, dim3(nblk_), dim3(kGB), 0, st_, dev_, p, q);
    sum_partials(1, 6);
    IPC_CL_CHK(evaluate(dev_.Xn, dev_.en, dev_.len, newChi));
    anyChanged = h_scal_[6] != 0.0;
    return hipSuccess;
}

inline hipError_t ClusterSolver2::max_edge_chi2(double& mx_out)
{
    ClusterDev& D = dev_;
    hipLaunchKernelGGL(gk_chi_edges, dim3(nblk_), dim3(kGB), 0, st_, D);
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

inline hipError_t ClusterSolver2::solve(hipStream_t st, const double* chain, int estride, const double* cand,
                                        int cstride, const PoseArr& src, int lo, int hi,
                                        const std::vector<int>& members, const int* from, const int* to,
                                        int iterations, ClusterOut& out, std::vector<double>* chi_host)
{
    const int L = hi - lo, nl = (int)members.size(), NS = 3 * nl;
    IPC_CL_CHK(ensure(L, nl));
    const int ld = L + 2;
    st_ = st; chi_host_ = chi_host;
    ClusterDev& D = dev_;
    D.chain = chain; D.estride = estride; D.lo = lo; D.L = L; D.nl = nl; D.ld = ld;
    D.cand = cand; D.cstride = cstride;
    {   // carve the workspaces
        double* p = d_edge_;
        auto take = [&](size_t n) { double* q = p; p += n; return q; };
        D.X = PoseArr{take(ld), take(ld), take(ld), take(ld), take(ld)};
        D.Xn = PoseArr{take(ld), take(ld), take(ld), take(ld), take(ld)};
        D.e = take(3 * ld); D.en = take(3 * ld); D.g = take(3 * ld); D.m = take(3 * ld);
        D.b = take(3 * ld); D.h = take(3 * ld); D.ps = take(9 * ld); D.nd = take(3 * ld); D.sc = take(3 * ld);
        D.chi_edges = take(ld + nl);
        double* q = d_loop_;
        auto takel = [&](size_t n) { double* r = q; q += n; return r; };
        D.le = takel(3 * nl); D.len = takel(3 * nl); D.lg = takel(3 * nl); D.lm = takel(3 * nl);
        D.gam = takel(9 * nl); D.nu = takel(3 * nl); D.rhs = takel(3 * nl);
        D.S = d_S_; D.ldS = NS + 1;
        D.partial = d_partial_; D.scal = d_scal_;
    }
    tab_.build(lo, hi, members, from, to);
    lband_stale = true;
    IPC_CL_CHK(hipMemcpyAsync(d_int_, tab_.host.data(), sizeof(int) * tab_.size(), hipMemcpyHostToDevice, st));
    IPC_CL_CHK(hipStreamSynchronize(st));           // the host table is pageable and reused
    D.lfrom = tab_.lfrom(d_int_); D.lto = tab_.lto(d_int_); D.lcand = tab_.lcand(d_int_);
    D.adj_ptr = tab_.adj_ptr(d_int_); D.adj_item = tab_.adj_item(d_int_);
    D.ev_ptr = tab_.ev_ptr(d_int_); D.ev_item = tab_.ev_item(d_int_);
    const double* sp[5] = {src.x, src.y, src.th, src.c, src.s};
    double* xp[5] = {D.X.x, D.X.y, D.X.th, D.X.c, D.X.s};
    for (int k = 0; k < 5; ++k)
        IPC_CL_CHK(hipMemcpyAsync(xp[k], sp[k] + lo, sizeof(double) * (L + 1), hipMemcpyDeviceToDevice, st));
    nblk_ = (L + nl + 1 + kGB - 1) / kGB;            // indices 0 .. L+nl
    IPC_CL_CHK(hipMemsetAsync(d_info_, 0, sizeof(int), st));
    return cluster_dogleg(*this, iterations, out, term_eps, tab_.L + tab_.nl, allow_damping);
}

}  // namespace ipc
