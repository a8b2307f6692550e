// SE(3) consistency cell solver -- one workgroup per cell, hand-written for gfx950 (wave64).
//
// Same contract and structure as se2_cell.hpp (reference src/consensus_utils.cpp:7-22 on a
// gauge-fixed odometry chain + one or two loop closures, g2o dog-leg), for g2o's EdgeSE3 /
// VertexSE3 conventions:
//   pose X = (R, t); error e = toVectorMQT(Z^-1 Xa^-1 Xb) = (t_E, q_E.xyz) with the quaternion
//   normalised and w >= 0; update X <- X * fromVectorMQT(delta) (right-multiplicative, body frame).
// Analytic Jacobians (the derivative isometry3d_gradients.h evaluates, in closed form):
//   B = de/dXb = D(E) = blockdiag(R_E, w_E I + [v_E]x)
//   A = de/dXa = -D(E) * Ad,   Ad = [[Rab^T, -2 Rab^T [tab]x], [0, Rab^T]],  Xab = Xa^-1 Xb.
// Chain closed form used for the Gauss-Newton solve (u = Jc h, see se2_cell.hpp):
//   B_j h_j + A_j h_{j-1} = u_j  <=>  h_j = rho_j + Ad_j h_{j-1},  rho_j = D(E_j)^-1 u_j,
// which in world-frame coordinates (omega = R h_q, tau = R h_t) is a pair of prefix sums
//   omega_j = omega_{j-1} + R_j rho_q,   tau_j = tau_{j-1} + R_j rho_t + 2 omega_{j-1} x (t_j - t_{j-1}),
// and the loop rows of G = Jl Jc^-1 are element-wise:
//   G_{l,j} = sigma_l D(E_l) T(X_j^-1 X_to(l)) D(E_j)^-1  for lo_l < j <= hi_l,
//   T(X) = [[R^T, -2 R^T [t]x], [0, R^T]].
#pragma once
#include <type_traits>

#include "block_prims.hpp"

namespace ipc {

// chain / candidate record fields (field-major): Rz (9, row-major), tz (3), Omega upper
// triangle (21, row order, already scaled by s for odometry), Sigma = Omega^-1 upper (21)
enum Se3Field { G_RZ = 0, G_TZ = 9, G_OM = 12, G_SG = 33, G_NFIELDS = 54 };

// blocked chain records: pairs 0..5 = Rz (9) + tz (3), 6..16 = information (21 + pad), 17..27 = covariance (21 + pad)
constexpr int kSe3BlkPairs = 28;

struct Se3View {
    const double* chain;      // [G_NFIELDS][estride]
    int estride;
    const double* chain_rec;  // the same values record-major: [edge][G_NFIELDS]
    const double2* chain_blk; // the same values in blocks of 64 edges: [block][kSe3BlkPairs][64] double2 (se3_lds_cell.hpp)
    const double* pose0;      // [12][V] open-loop poses: R row-major (9), t (3)
    int V;
    const double* cand;       // [G_NFIELDS][cstride]
    int cstride;
    const int* cand_from;
    const int* cand_to;
    double term_eps;          // convergence shortcut of the trial loop, see Se2View::term_eps
    double* dbg;              // debug side channel (NULL in production)
};

struct Pose3 { double R[9]; double t[3]; };

// ---------------- small dense helpers (fully unrolled, registers only) ----------------
__device__ __forceinline__ void m3_mul(const double* A, const double* B, double* C)      // C = A B
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
__device__ __forceinline__ void m3_tmul(const double* A, const double* B, double* C)     // C = A^T B
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}
__device__ __forceinline__ void m3_mult(const double* A, const double* B, double* C)     // C = A B^T
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + A[3 * i + 2] * B[3 * j + 2];
}
__device__ __forceinline__ void m3_vec(const double* A, const double* x, double* y)      // y = A x
{
#pragma unroll
    for (int i = 0; i < 3; ++i) y[i] = A[3 * i] * x[0] + A[3 * i + 1] * x[1] + A[3 * i + 2] * x[2];
}
__device__ __forceinline__ void m3_tvec(const double* A, const double* x, double* y)     // y = A^T x
{
#pragma unroll
    for (int i = 0; i < 3; ++i) y[i] = A[i] * x[0] + A[3 + i] * x[1] + A[6 + i] * x[2];
}
__device__ __forceinline__ void cross3(const double* a, const double* b, double* c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
// symmetric 6x6 stored as the 21 upper-triangle values in row order
__device__ __forceinline__ int sym6_idx(int i, int j)
{
    const int a = i < j ? i : j, b = i < j ? j : i;
    return a * 6 - a * (a - 1) / 2 + (b - a);
}
__device__ __forceinline__ void sym6_mul(const double* S, const double* x, double* y)
{
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) acc += S[sym6_idx(i, j)] * x[j];
        y[i] = acc;
    }
}
__device__ __forceinline__ double sym6_quad(const double* S, const double* x)
{
    double y[6];
    sym6_mul(S, x, y);
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) acc += x[i] * y[i];
    return acc;
}

// Eigen Quaternion(Matrix3) + g2o normalize (unit norm, w >= 0): q = (x, y, z, w)
__device__ __forceinline__ void quat_from_R(const double* R, double* q)
{
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t;
        q[1] = (R[2] - R[6]) * t;
        q[2] = (R[3] - R[1]) * t;
    } else if (R[0] >= R[4] && R[0] >= R[8]) {            // i = 0
        t = sqrt(R[0] - R[4] - R[8] + 1.0);
        q[0] = 0.5 * t; t = 0.5 / t;
        q[3] = (R[7] - R[5]) * t; q[1] = (R[3] + R[1]) * t; q[2] = (R[6] + R[2]) * t;
    } else if (R[4] > R[0] && R[4] >= R[8]) {             // i = 1
        t = sqrt(R[4] - R[8] - R[0] + 1.0);
        q[1] = 0.5 * t; t = 0.5 / t;
        q[3] = (R[2] - R[6]) * t; q[2] = (R[7] + R[5]) * t; q[0] = (R[1] + R[3]) * t;
    } else {                                              // i = 2
        t = sqrt(R[8] - R[0] - R[4] + 1.0);
        q[2] = 0.5 * t; t = 0.5 / t;
        q[3] = (R[3] - R[1]) * t; q[0] = (R[2] + R[6]) * t; q[1] = (R[5] + R[7]) * t;
    }
    const double n = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double sgn = q[3] < 0 ? -n : n;
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] *= sgn;
}
// Eigen Quaternion::toRotationMatrix for (w, x, y, z)
__device__ __forceinline__ void R_from_quat(double w, double x, double y, double z, double* R)
{
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

// Everything an SE3 edge a -> b contributes at the current poses.
struct Edge3 {
    double Rab[9], tab[3];    // Xab = Xa^-1 Xb
    double RE[9];             // rotation of E = Z^-1 Xab
    double qw, qv[3];         // normalised quaternion of RE (w >= 0)
    double e[6];              // error (t_E, qv)
};
__device__ __forceinline__ void se3_edge(const Pose3& a, const Pose3& b, const double* Rz, const double* tz, Edge3& E)
{
    m3_tmul(a.R, b.R, E.Rab);
    double d[3] = {b.t[0] - a.t[0], b.t[1] - a.t[1], b.t[2] - a.t[2]};
    m3_tvec(a.R, d, E.tab);
    m3_tmul(Rz, E.Rab, E.RE);
    double l[3] = {E.tab[0] - tz[0], E.tab[1] - tz[1], E.tab[2] - tz[2]};
    m3_tvec(Rz, l, E.e);
    double q[4];
    quat_from_R(E.RE, q);
    E.qv[0] = q[0]; E.qv[1] = q[1]; E.qv[2] = q[2]; E.qw = q[3];
    E.e[3] = q[0]; E.e[4] = q[1]; E.e[5] = q[2];
}
// y = D(E)^T x = (RE^T x_t, (w I - [v]x) x_q)
__device__ __forceinline__ void se3_Dt(const Edge3& E, const double* x, double* y)
{
    m3_tvec(E.RE, x, y);
    double c[3];
    cross3(E.qv, x + 3, c);
#pragma unroll
    for (int i = 0; i < 3; ++i) y[3 + i] = E.qw * x[3 + i] - c[i];
}
// y = D(E) x = (RE x_t, (w I + [v]x) x_q)
__device__ __forceinline__ void se3_D(const Edge3& E, const double* x, double* y)
{
    m3_vec(E.RE, x, y);
    double c[3];
    cross3(E.qv, x + 3, c);
#pragma unroll
    for (int i = 0; i < 3; ++i) y[3 + i] = E.qw * x[3 + i] + c[i];
}
// y = D(E)^-1 x = (RE^T x_t, (w I - [v]x + v v^T / w) x_q)
__device__ __forceinline__ void se3_Dinv(const Edge3& E, const double* x, double* y)
{
    m3_tvec(E.RE, x, y);
    double c[3];
    cross3(E.qv, x + 3, c);
    const double vx = (E.qv[0] * x[3] + E.qv[1] * x[4] + E.qv[2] * x[5]) / E.qw;
#pragma unroll
    for (int i = 0; i < 3; ++i) y[3 + i] = E.qw * x[3 + i] - c[i] + E.qv[i] * vx;
}
// y = Ad x,  Ad = [[Rab^T, -2 Rab^T [tab]x],[0, Rab^T]]
__device__ __forceinline__ void se3_Ad(const Edge3& E, const double* x, double* y)
{
    double c[3];
    cross3(E.tab, x + 3, c);
    double u[3] = {x[0] - 2 * c[0], x[1] - 2 * c[1], x[2] - 2 * c[2]};
    m3_tvec(E.Rab, u, y);
    m3_tvec(E.Rab, x + 3, y + 3);
}
// y = Ad^T x = (Rab x_t, Rab x_q + 2 tab x (Rab x_t))
__device__ __forceinline__ void se3_Adt(const Edge3& E, const double* x, double* y)
{
    m3_vec(E.Rab, x, y);
    double r[3], c[3];
    m3_vec(E.Rab, x + 3, r);
    cross3(E.tab, y, c);
#pragma unroll
    for (int i = 0; i < 3; ++i) y[3 + i] = r[i] + 2 * c[i];
}
// w = J_edge (va, vb) = D(E) (vb - Ad va)
__device__ __forceinline__ void se3_apply_J(const Edge3& E, const double* va, const double* vb, double* w)
{
    double a[6], d[6];
    se3_Ad(E, va, a);
#pragma unroll
    for (int i = 0; i < 6; ++i) d[i] = vb[i] - a[i];
    se3_D(E, d, w);
}

struct LoopConst3 {
    int f, t, lo, hi;
    double sigma;
    double Rz[9], tz[3];
    double om[21], sg[21];
};
struct LoopState3 {
    Pose3 pf, pt;             // end-point poses
    double Rab[9], tab[3], RE[9], qw, qv[3];   // the loop edge at these poses (Edge3 without e)
    double e[6];
    double g[6];              // D^T Om e   (added with a minus sign to b at the "to" pose)
    double m[6];              // Ad^T g     (added to b at the "from" pose)
    double chi;
};

template <int W, int NL>
struct Se3Scratch {
    double red[32 * 16 * 3];  // wide reductions: [wave][96]; narrow ones: [value][16]
    double hi_pose[W][12];
    double hi_vec[W][6];
    double lo_vec[W][6];
    double lvec[NL][2][6];
    double scan[W][9];
    double sol[NL * 6 + 3];
};

template <int W, int M, int NL>
struct Se3Shared {
    Se3Scratch<W, NL> scr[2];
    LoopConst3 lc[NL];
    LoopState3 ls[2][NL];
    double w0tot[80];         // wave-0 temporaries of the capacitance solve: totals,
    double w0gam[NL][36];     //   Gamma_l (6x6 each),
    double w0aug[NL * 6][NL * 6 + 1];   // the augmented system,
    double w0mu[NL * 6];      //   its solution
};

// SPD solve by Cholesky, N = 6 or 12 (wave 0 only)
template <int N>
__device__ __forceinline__ bool chol_solve_n(double (&A)[N][N], double (&b)[N])
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

struct CellResult3 {
    double max_chi2, chi2_total;
    int iterations, tries, flags, evals;
};

template <int W, int M, int NL>
__device__ void se3_solve_cell(const Se3View& P, int lo_abs, int L, const int (&cand)[2], int iterations,
                               Se3Shared<W, M, NL>& sh, CellResult3& res)
{
    constexpr int NS = NL * 6;
    const double term_scale = P.term_eps / (double)(L + NL);   // 0: the test is off
    bool lastGN = false;
    constexpr int NSS = NS * (NS + 1) / 2;
    constexpr int KR = 2 + NS + NSS;                 // b^T b, b^T H b, d, S upper: 29 (diag) / 92 (pair)
    constexpr int NG = (KR + 15) / 16;               // packed groups of 16
    static_assert(NG * 16 <= 96 && W <= 16, "reduction scratch layout");
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int jbase = 1 + wave * 64 * M + lane;

    // ---------------- loop constants -> LDS ----------------
    if (tid < NL) {
        const int l = tid, c = cand[l];
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

    // ---------------- per-lane state ----------------
    Pose3 X[M], Xn[M];
    double e[M][6], b[M][6], h[M][6];
    bool valid[M];
    unsigned eoff[M];
    Pose3 gauge;
#pragma unroll
    for (int k = 0; k < 9; ++k) gauge.R[k] = P.pose0[(size_t)k * P.V + lo_abs];
#pragma unroll
    for (int k = 0; k < 3; ++k) gauge.t[k] = P.pose0[(size_t)(9 + k) * P.V + lo_abs];
#pragma unroll
    for (int s = 0; s < M; ++s) {
        const int j = jbase + s * 64;
        valid[s] = j <= L;
        eoff[s] = (unsigned)(valid[s] ? lo_abs + j - 1 : lo_abs) * (unsigned)(G_NFIELDS * 8);   // byte offset of the edge record
        const int ja = valid[s] ? lo_abs + j : lo_abs;
#pragma unroll
        for (int k = 0; k < 9; ++k) X[s].R[k] = P.pose0[(size_t)k * P.V + ja];
#pragma unroll
        for (int k = 0; k < 3; ++k) X[s].t[k] = P.pose0[(size_t)(9 + k) * P.V + ja];
        Xn[s] = X[s];
#pragma unroll
        for (int k = 0; k < 6; ++k) { e[s][k] = b[s][k] = h[s][k] = 0.0; }
    }
    __syncthreads();
    int lf[NL], lt[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) { lf[l] = sh.lc[l].f; lt[l] = sh.lc[l].t; }
    if (tid == 0) {
#pragma unroll
        for (int l = 0; l < NL; ++l)
#pragma unroll
            for (int bsel = 0; bsel < 2; ++bsel) {
                if (lf[l] == 0) sh.ls[bsel][l].pf = gauge;
                if (lt[l] == 0) sh.ls[bsel][l].pt = gauge;
            }
    }

    auto opaque = [&]() {
#pragma unroll
        for (int s = 0; s < M; ++s) asm volatile("" : "+v"(eoff[s]));
    };
    auto ldc = [&](int field, int s) -> double {
        // record-major copy: one per-slot offset, the field is a compile-time displacement
        const char* fb = reinterpret_cast<const char*>(P.chain_rec) + field * 8;
        return *reinterpret_cast<const double*>(fb + eoff[s]);
    };
    auto ld_rz = [&](int s, double* Rz, double* tz) {
#pragma unroll
        for (int k = 0; k < 9; ++k) Rz[k] = ldc(G_RZ + k, s);
#pragma unroll
        for (int k = 0; k < 3; ++k) tz[k] = ldc(G_TZ + k, s);
    };
    auto ld_sym = [&](int field0, int s, double* S) {
#pragma unroll
        for (int k = 0; k < 21; ++k) S[k] = ldc(field0 + k, s);
    };

    int phase = 0, cur = 0;

    auto prev_pose = [&](const Pose3 (&Y)[M], const Pose3& edge, Pose3 (&A)[M]) {
        Pose3 carry = edge;
#pragma unroll
        for (int s = 0; s < M; ++s) {
#pragma unroll
            for (int k = 0; k < 9; ++k) A[s].R[k] = lane_prev(Y[s].R[k], carry.R[k]);
#pragma unroll
            for (int k = 0; k < 3; ++k) A[s].t[k] = lane_prev(Y[s].t[k], carry.t[k]);
            if (s + 1 < M) {
#pragma unroll
                for (int k = 0; k < 9; ++k) carry.R[k] = read_lane(Y[s].R[k], 63);
#pragma unroll
                for (int k = 0; k < 3; ++k) carry.t[k] = read_lane(Y[s].t[k], 63);
            }
        }
    };
    auto publish_poses = [&](const Pose3 (&Y)[M], Se3Scratch<W, NL>& S, int lsbuf) {
        if (lane == 63) {
#pragma unroll
            for (int k = 0; k < 9; ++k) S.hi_pose[wave][k] = Y[M - 1].R[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) S.hi_pose[wave][9 + k] = Y[M - 1].t[k];
        }
#pragma unroll
        for (int l = 0; l < NL; ++l)
#pragma unroll
            for (int s = 0; s < M; ++s) {
                const int j = jbase + s * 64;
                if (j == lf[l]) sh.ls[lsbuf][l].pf = Y[s];
                if (j == lt[l]) sh.ls[lsbuf][l].pt = Y[s];
            }
    };
    auto publish_endpoint_vec = [&](const double (&v)[M][6], Se3Scratch<W, NL>& S) {
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            if (tid == 0) {
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    if (lf[l] == 0) S.lvec[l][0][k] = 0.0;
                    if (lt[l] == 0) S.lvec[l][1][k] = 0.0;
                }
            }
#pragma unroll
            for (int s = 0; s < M; ++s) {
                const int j = jbase + s * 64;
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    if (j == lf[l]) S.lvec[l][0][k] = v[s][k];
                    if (j == lt[l]) S.lvec[l][1][k] = v[s][k];
                }
            }
        }
    };
    auto edge_pose_of = [&](const Se3Scratch<W, NL>& S) -> Pose3 {
        Pose3 p = gauge;
        if (wave > 0) {
#pragma unroll
            for (int k = 0; k < 9; ++k) p.R[k] = S.hi_pose[wave - 1][k];
#pragma unroll
            for (int k = 0; k < 3; ++k) p.t[k] = S.hi_pose[wave - 1][9 + k];
        }
        return p;
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
    // loop l: evaluate the loop edge at the end-point poses of buffer bsel (one lane)
    auto loop_eval = [&](int l, int bsel) -> double {
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
        return chi;
    };
    auto loop_quad = [&](int l, const Se3Scratch<W, NL>& S) -> double {
        Edge3 E;
        loop_edge(l, cur, E);
        double va[6], vb[6], w[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) { va[k] = S.lvec[l][0][k]; vb[k] = S.lvec[l][1][k]; }
        se3_apply_J(E, va, vb, w);
        return sym6_quad(sh.lc[l].om, w);
    };

    Pose3 edge = gauge, edgeN = gauge;
    // keep: store the odometry errors (committed state); trial passes only need chi2 -- an accepted
    // trial recomputes its errors once, which is cheaper than a second error array in registers
    auto evaluate = [&](const Pose3 (&Y)[M], auto keep_c, int bsel, bool changed, bool& anyChanged) -> double {
        constexpr bool KEEP = decltype(keep_c)::value != 0;
        Se3Scratch<W, NL>& S = sh.scr[phase & 1];
        publish_poses(Y, S, bsel);
        __syncthreads();
        edgeN = edge_pose_of(S);
        ++phase;
        Pose3 An[M];
        prev_pose(Y, edgeN, An);
        double part = 0.0;
#pragma unroll
        for (int s = 0; s < M; ++s) {
            if (!valid[s]) continue;
            double Rz[9], tz[3], om[21];
            ld_rz(s, Rz, tz);
            Edge3 E;
            se3_edge(An[s], Y[s], Rz, tz, E);
            if (KEEP) {
#pragma unroll
                for (int k = 0; k < 6; ++k) e[s][k] = E.e[k];
            }
            ld_sym(G_OM, s, om);
            part += sym6_quad(om, E.e);
        }
        if (tid < NL) part += loop_eval(tid, bsel);
        part = wave_sum(part);
        const bool wchg = __ballot(changed) != 0ull;
        Se3Scratch<W, NL>& S2 = sh.scr[phase & 1];
        if (lane == 0) { S2.red[wave] = part; S2.red[16 + wave] = wchg ? 1.0 : 0.0; }
        __syncthreads();
        double tot[2];
        gather_totals<2>(S2.red, W, tot);
        ++phase;
        anyChanged = tot[1] != 0.0;
        return tot[0];
    };

    int evals = 0;
    double currentChi;
    {
        bool dummy;
        currentChi = evaluate(X, std::integral_constant<int, 1>{}, cur, false, dummy);
        edge = edgeN;
        ++evals;
    }

    double delta = 1e4;
    const int maxTrials = 100;
    int it_done = 0, tries_total = 0, flags = 0;

    for (int it = 0; it < iterations; ++it) {
        opaque();
        Pose3 A[M];
        prev_pose(X, edge, A);
        // ---- phase A: g = D^T Om e, m = Ad^T g; b_j = m_{j+1} - g_j + loops ----
        double pb[6];                                 // b of this wave's predecessor pose
        {
            double g[M][6], m[M][6];
#pragma unroll
            for (int s = 0; s < M; ++s) {
#pragma unroll
                for (int k = 0; k < 6; ++k) { g[s][k] = 0.0; m[s][k] = 0.0; }
                if (!valid[s]) continue;
                double Rz[9], tz[3], om[21], qo[6];
                ld_rz(s, Rz, tz);
                Edge3 E;
                se3_edge(A[s], X[s], Rz, tz, E);
                ld_sym(G_OM, s, om);
                sym6_mul(om, e[s], qo);
                se3_Dt(E, qo, g[s]);
                se3_Adt(E, g[s], m[s]);
            }
            Se3Scratch<W, NL>& S = sh.scr[phase & 1];
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < 6; ++k) S.lo_vec[wave][k] = m[0][k];
            }
            if (lane == 63) {
#pragma unroll
                for (int k = 0; k < 6; ++k) S.hi_vec[wave][k] = g[M - 1][k];
            }
            __syncthreads();
            double nx[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) nx[k] = (wave + 1 < W) ? S.lo_vec[wave + 1][k] : 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) pb[k] = 0.0;
            if (wave > 0) {
#pragma unroll
                for (int k = 0; k < 6; ++k) pb[k] = read_lane(m[0][k], 0) - S.hi_vec[wave - 1][k];
                const int jp = wave * 64 * M;
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    const LoopState3& st = sh.ls[cur][l];
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        if (jp == lt[l]) pb[k] -= st.g[k];
                        if (jp == lf[l]) pb[k] += st.m[k];
                    }
                }
            }
            ++phase;
#pragma unroll
            for (int s = M - 1; s >= 0; --s) {
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const double u = lane_next(m[s][k], nx[k]);
                    if (s > 0) nx[k] = read_lane(m[s][k], 0);
                    b[s][k] = valid[s] ? u - g[s][k] : 0.0;
                }
            }
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const LoopState3& st = sh.ls[cur][l];
#pragma unroll
                for (int s = 0; s < M; ++s) {
                    const int j = jbase + s * 64;
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        if (j == lt[l]) b[s][k] -= st.g[k];
                        if (j == lf[l]) b[s][k] += st.m[k];
                    }
                }
            }
        }
        // ---- phase B: b^T b, b^T H b and the loop-free capacitance partials; solve on wave 0 ----
        // G_{l,j} = Gamma_l Phi_j with (poses shifted by the gauge position o, t~ = t - o)
        //   Phi_j   = T(X~_j^-1) D(E_j)^-1 = [[U, K],[0, Vq]],  U = R_j RE_j^T, Vq = R_j Q_j^-1, K = 2 [t~_j]x Vq
        //   Gamma_l = sigma_l D(E_l) T(X~_to) = sigma_l [[RE_l R_to^T, -2 RE_l R_to^T [t~_to]x],[0, Q_l R_to^T]]
        // so per edge only Psi_j = Phi_j Cov_j Phi_j^T (21 values) and w_j = Phi_j e_j (6) are
        // accumulated per loop range; S_ll' = Gamma_l M_ll' Gamma_l'^T, d_l = e_l - Gamma_l W_l.
        double bb, bHb, alpha, hsdNorm;
        double nu[NL][6];                             // Gamma_l^T mu_l
        int rlo[NL], rhi[NL];
#pragma unroll
        for (int l = 0; l < NL; ++l) { rlo[l] = sh.lc[l].lo; rhi[l] = sh.lc[l].hi; }
        {
            Se3Scratch<W, NL>& S = sh.scr[phase & 1];
            publish_endpoint_vec(b, S);
            // value layout: [0] b^T b, [1] b^T H b, [2..8) W_1, [8..29) M_11; pair cells: [29..35) W_2,
            // [35..56) M_22, [56..77) M_12
            constexpr int NV = NL == 1 ? 29 : 77;
            constexpr int NGR = (NV + 15) / 16;
            double part[NGR * 16];
#pragma unroll
            for (int k = 0; k < NGR * 16; ++k) part[k] = 0.0;
            double cb[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) cb[k] = pb[k];
#pragma unroll
            for (int s = 0; s < M; ++s) {
                double qb[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    qb[k] = lane_prev(b[s][k], cb[k]);
                    if (s + 1 < M) cb[k] = read_lane(b[s][k], 63);
                }
                if (!valid[s]) continue;
                double Rz[9], tz[3], om[21], sg[21];
                ld_rz(s, Rz, tz);
                Edge3 E;
                se3_edge(A[s], X[s], Rz, tz, E);
                ld_sym(G_OM, s, om);
                ld_sym(G_SG, s, sg);
#pragma unroll
                for (int k = 0; k < 6; ++k) part[0] += b[s][k] * b[s][k];
                double w[6];
                se3_apply_J(E, qb, b[s], w);
                part[1] += sym6_quad(om, w);
                // Phi = [[U, K],[0, Vq]]
                double U[9], Qi[9], Vq[9], K[9];
                m3_mult(X[s].R, E.RE, U);
                {
                    const double iw = 1.0 / E.qw;
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int k = 0; k < 3; ++k) Qi[3 * i + k] = E.qv[i] * E.qv[k] * iw + (i == k ? E.qw : 0.0);
                    Qi[1] += E.qv[2]; Qi[2] -= E.qv[1];      // - [v]x
                    Qi[3] -= E.qv[2]; Qi[5] += E.qv[0];
                    Qi[6] += E.qv[1]; Qi[7] -= E.qv[0];
                }
                m3_mul(X[s].R, Qi, Vq);
                const double tt[3] = {X[s].t[0] - gauge.t[0], X[s].t[1] - gauge.t[1], X[s].t[2] - gauge.t[2]};
#pragma unroll
                for (int c = 0; c < 3; ++c) {             // K[:,c] = 2 tt x Vq[:,c]
                    K[0 + c] = 2 * (tt[1] * Vq[6 + c] - tt[2] * Vq[3 + c]);
                    K[3 + c] = 2 * (tt[2] * Vq[0 + c] - tt[0] * Vq[6 + c]);
                    K[6 + c] = 2 * (tt[0] * Vq[3 + c] - tt[1] * Vq[0 + c]);
                }
                auto phi = [&](int r, int k) -> double {  // Phi[r][k]
                    if (r < 3) return k < 3 ? U[3 * r + k] : K[3 * r + (k - 3)];
                    return k < 3 ? 0.0 : Vq[3 * (r - 3) + (k - 3)];
                };
                // PS = Phi Cov (rows r, columns c), Psi = PS Phi^T (upper triangle, sym6 order)
                double psi[21], wj[6];
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
                        for (int k = (c < 3 ? 0 : 3); k < 6; ++k) acc += ps[k] * phi(c, k);
                        psi[sym6_idx(r, c)] = acc;
                    }
                    double acc = 0.0;
#pragma unroll
                    for (int a = (r < 3 ? 0 : 3); a < 6; ++a) acc += phi(r, a) * e[s][a];
                    wj[r] = acc;
                }
                const int j = jbase + s * 64;
                const double m1 = (j > rlo[0] && j <= rhi[0]) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < 6; ++k) part[2 + k] += m1 * wj[k];
#pragma unroll
                for (int k = 0; k < 21; ++k) part[8 + k] += m1 * psi[k];
                if constexpr (NL == 2) {
                    const double m2 = (j > rlo[1] && j <= rhi[1]) ? 1.0 : 0.0, m12 = m1 * m2;
#pragma unroll
                    for (int k = 0; k < 6; ++k) part[29 + k] += m2 * wj[k];
#pragma unroll
                    for (int k = 0; k < 21; ++k) { part[35 + k] += m2 * psi[k]; part[56 + k] += m12 * psi[k]; }
                }
            }
#pragma unroll
            for (int gq = 0; gq < NGR; ++gq) {
                double v16[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) v16[k] = part[16 * gq + k];
                wave_sum16_store(v16, &S.red[wave * 96 + 16 * gq]);
            }
            __syncthreads();
            ++phase;
            Se3Scratch<W, NL>& S2 = sh.scr[phase & 1];
            if (wave == 0) {
                constexpr int RS = NS + 1;
                // totals
#pragma unroll
                for (int k0 = 0; k0 < NV; k0 += 64) {
                    if (k0 + lane < NV) {
                        double acc = 0.0;
#pragma unroll
                        for (int w = 0; w < W; ++w) acc += S.red[w * 96 + k0 + lane];
                        sh.w0tot[k0 + lane] = acc;
                    }
                }
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
                            // P = RE Rto^T, row r
                            double Pr[3];
#pragma unroll
                            for (int k = 0; k < 3; ++k) Pr[k] = st.RE[3 * r] * st.pt.R[3 * k] + st.RE[3 * r + 1] * st.pt.R[3 * k + 1] + st.RE[3 * r + 2] * st.pt.R[3 * k + 2];
                            if (c < 3) g = Pr[c];
                            else {
                                // -2 (P [tt]x)[r][c-3];  [tt]x = [[0,-t2,t1],[t2,0,-t0],[-t1,t0,0]]
                                const int cc = c - 3;
                                const double col[3] = {cc == 0 ? 0.0 : (cc == 1 ? -tt[2] : tt[1]),
                                                       cc == 0 ? tt[2] : (cc == 1 ? 0.0 : -tt[0]),
                                                       cc == 0 ? -tt[1] : (cc == 1 ? tt[0] : 0.0)};
                                g = -2 * (Pr[0] * col[0] + Pr[1] * col[1] + Pr[2] * col[2]);
                            }
                        } else if (c >= 3) {
                            // (Q Rto^T)[r-3][c-3],  Q = w I + [v]x
                            const int rr = r - 3, cc = c - 3;
                            double Qr[3] = {rr == 0 ? st.qw : (rr == 1 ? st.qv[2] : -st.qv[1]),
                                            rr == 0 ? -st.qv[2] : (rr == 1 ? st.qw : st.qv[0]),
                                            rr == 0 ? st.qv[1] : (rr == 1 ? -st.qv[0] : st.qw)};
                            g = Qr[0] * st.pt.R[3 * cc] + Qr[1] * st.pt.R[3 * cc + 1] + Qr[2] * st.pt.R[3 * cc + 2];
                        }
                        sh.w0gam[l][lane] = q.sigma * g;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // augmented system: one entry per (lane, round)
#pragma unroll
                for (int q0 = 0; q0 < NS * RS; q0 += 64) {
                    const int idx = q0 + lane;
                    if (idx < NS * RS) {
                        const int r = idx / RS, c = idx % RS;
                        const int l1 = r / 6, i = r % 6;
                        const double* g1 = &sh.w0gam[l1][6 * i];
                        double val;
                        if (c < NS) {
                            const int l2 = c / 6, k = c % 6;
                            const double* g2 = &sh.w0gam[l2][6 * k];
                            const int mb = l1 == l2 ? (l1 == 0 ? 8 : 35) : 56;
                            val = 0.0;
#pragma unroll
                            for (int bq = 0; bq < 6; ++bq) {
                                double t = 0.0;
#pragma unroll
                                for (int a = 0; a < 6; ++a) t += g1[a] * sh.w0tot[mb + sym6_idx(a, bq)];
                                val += t * g2[bq];
                            }
                            if (l1 == l2) val += sh.lc[l1].sg[sym6_idx(i, k)];
                        } else {
                            const int wb = l1 == 0 ? 2 : 29;
                            double t = 0.0;
#pragma unroll
                            for (int a = 0; a < 6; ++a) t += g1[a] * sh.w0tot[wb + a];
                            val = sh.ls[cur][l1].e[i] - t;
                        }
                        sh.w0aug[r][c] = val;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
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
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                double lq = 0.0;
                if (lane < NL) lq = loop_quad(lane, S);
                double bHbTot = sh.w0tot[1] + read_lane(lq, 0);
                if (NL == 2) bHbTot += read_lane(lq, 1);
                if (lane < NS) {
                    const int l = lane / 6, c = lane % 6;
                    double t = 0.0;
#pragma unroll
                    for (int r = 0; r < 6; ++r) t += sh.w0gam[l][6 * r + c] * sh.w0mu[6 * l + r];
                    S2.sol[lane] = t;
                }
                if (lane == 0) {
                    S2.sol[NS] = sh.w0tot[0];
                    S2.sol[NS + 1] = bHbTot;
                    S2.sol[NS + 2] = okS ? 1.0 : 0.0;
                }
            }
            __syncthreads();
            ++phase;
#pragma unroll
            for (int l = 0; l < NL; ++l)
#pragma unroll
                for (int k = 0; k < 6; ++k) nu[l][k] = S2.sol[6 * l + k];
            bb = S2.sol[NS];
            bHb = S2.sol[NS + 1];
            if (S2.sol[NS + 2] == 0.0) { flags |= 2; break; }
            alpha = bb / bHb;
            hsdNorm = sqrt(alpha * alpha * bb);
        }

        // ---- phase C: u = -Sg G^T mu - e, rho = D^-1 u, world-frame prefix sums -> h ----
        double hgnNorm, bh, hHh;
        {
            double rq[M][3], rt[M][3];               // world-frame increments R_j rho_q, R_j rho_t
#pragma unroll
            for (int s = 0; s < M; ++s) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { rq[s][k] = 0.0; rt[s][k] = 0.0; }
                if (!valid[s]) continue;
                double Rz[9], tz[3], sg[21];
                ld_rz(s, Rz, tz);
                Edge3 E;
                se3_edge(A[s], X[s], Rz, tz, E);
                ld_sym(G_SG, s, sg);
                // sum_l G_l^T mu_l = Phi_j^T n,  n = sum_l mask_l nu_l;
                // Phi^T n = (U^T n_t, Vq^T (n_q - 2 t~ x n_t))
                const int j = jbase + s * 64;
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
                    m3_mult(X[s].R, E.RE, U);
                    const double iw = 1.0 / E.qw;
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int k = 0; k < 3; ++k) Qi[3 * i + k] = E.qv[i] * E.qv[k] * iw + (i == k ? E.qw : 0.0);
                    Qi[1] += E.qv[2]; Qi[2] -= E.qv[1];
                    Qi[3] -= E.qv[2]; Qi[5] += E.qv[0];
                    Qi[6] += E.qv[1]; Qi[7] -= E.qv[0];
                    m3_mul(X[s].R, Qi, Vq);
                    const double tt[3] = {X[s].t[0] - gauge.t[0], X[s].t[1] - gauge.t[1], X[s].t[2] - gauge.t[2]};
                    double cr[3], y3[3];
                    cross3(tt, nn, cr);
#pragma unroll
                    for (int k = 0; k < 3; ++k) y3[k] = nn[3 + k] - 2 * cr[k];
                    m3_tvec(U, nn, wv);
                    m3_tvec(Vq, y3, wv + 3);
                }
                double v[6], u[6], rho[6];
                sym6_mul(sg, wv, v);
#pragma unroll
                for (int k = 0; k < 6; ++k) u[k] = -v[k] - e[s][k];
                se3_Dinv(E, u, rho);
                m3_vec(X[s].R, rho + 3, rq[s]);
                m3_vec(X[s].R, rho, rt[s]);
            }
            // omega prefix (wave-local)
            double lo_[M][3], co[3] = {0, 0, 0};
#pragma unroll
            for (int s = 0; s < M; ++s)
#pragma unroll
                for (int k = 0; k < 3; ++k) { lo_[s][k] = wave_inclusive_scan(rq[s][k]) + co[k]; co[k] = read_lane(lo_[s][k], 63); }
            // tau prefix (wave-local) with the wave-local omega of the previous pose
            double lt_[M][3], ct[3] = {0, 0, 0}, cp[3] = {0, 0, 0};
#pragma unroll
            for (int s = 0; s < M; ++s) {
                double op[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) { op[k] = lane_prev(lo_[s][k], cp[k]); cp[k] = read_lane(lo_[s][k], 63); }
                double term[3] = {0, 0, 0};
                if (valid[s]) {
                    double d3[3] = {X[s].t[0] - A[s].t[0], X[s].t[1] - A[s].t[1], X[s].t[2] - A[s].t[2]}, c[3];
                    cross3(op, d3, c);
#pragma unroll
                    for (int k = 0; k < 3; ++k) term[k] = rt[s][k] + 2 * c[k];
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) { lt_[s][k] = wave_inclusive_scan(term[k]) + ct[k]; ct[k] = read_lane(lt_[s][k], 63); }
            }
            Se3Scratch<W, NL>& S = sh.scr[phase & 1];
            double last[3] = {edge.t[0], edge.t[1], edge.t[2]};
#pragma unroll
            for (int s = 0; s < M; ++s) {
                const unsigned long long vm = __ballot(valid[s]);
                if (vm) {
                    const int ll = 63 - __clzll((long long)vm);
#pragma unroll
                    for (int k = 0; k < 3; ++k) last[k] = read_lane(X[s].t[k], ll);
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { S.scan[wave][k] = co[k]; S.scan[wave][3 + k] = ct[k]; S.scan[wave][6 + k] = last[k] - edge.t[k]; }
            }
            __syncthreads();
            double bo[3] = {0, 0, 0}, bt[3] = {0, 0, 0};   // bases: omega, tau of the predecessor pose
#pragma unroll
            for (int w = 0; w < W; ++w) {
                if (w < wave) {
                    double dT[3] = {S.scan[w][6], S.scan[w][7], S.scan[w][8]}, c[3];
                    cross3(bo, dT, c);
#pragma unroll
                    for (int k = 0; k < 3; ++k) bt[k] += S.scan[w][3 + k] + 2 * c[k];
#pragma unroll
                    for (int k = 0; k < 3; ++k) bo[k] += S.scan[w][k];
                }
            }
            ++phase;
            // h (body frame) and the per-iteration scalars |h|^2, b.h, h^T H h
            // |h|^2 and b.h; h^T H h = b^T h since h solves H h = b
            double p0 = 0.0, p1 = 0.0;
#pragma unroll
            for (int s = 0; s < M; ++s) {
                if (valid[s]) {
                    double om3[3], ta3[3], d3[3] = {X[s].t[0] - edge.t[0], X[s].t[1] - edge.t[1], X[s].t[2] - edge.t[2]}, c[3];
                    cross3(bo, d3, c);
#pragma unroll
                    for (int k = 0; k < 3; ++k) { om3[k] = lo_[s][k] + bo[k]; ta3[k] = lt_[s][k] + bt[k] + 2 * c[k]; }
                    m3_tvec(X[s].R, ta3, &h[s][0]);
                    m3_tvec(X[s].R, om3, &h[s][3]);
#pragma unroll
                    for (int k = 0; k < 6; ++k) { p0 += h[s][k] * h[s][k]; p1 += b[s][k] * h[s][k]; }
                } else {
#pragma unroll
                    for (int k = 0; k < 6; ++k) h[s][k] = 0.0;
                }
            }
            Se3Scratch<W, NL>& S2 = sh.scr[phase & 1];
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

        // converged (Se2View::term_eps): in the Newton regime (the last iteration took the full Gauss-Newton
        // step at its first trial) and one more such step cannot move any edge's chi2 by more than
        // 2 sqrt(term_eps) relative; g2o would still run its trial loop to Terminate
        if (lastGN && hgnNorm < delta && fabs(bh) < term_scale * currentChi) { it_done = it + 1; tries_total += maxTrials; flags |= 1; break; }
        // ---- trial loop ----
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
                double p0 = 0.0, p1 = 0.0;
#pragma unroll
                for (int s = 0; s < M; ++s) {
                    if (!valid[s]) continue;
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        const double sk = alpha * b[s][k], ak = h[s][k] - sk;
                        p0 += sk * ak;
                        p1 += ak * ak;
                    }
                }
                p0 = wave_sum(p0); p1 = wave_sum(p1);
                Se3Scratch<W, NL>& S = sh.scr[phase & 1];
                if (lane == 0) { S.red[wave] = p0; S.red[16 + wave] = p1; }
                __syncthreads();
                double tot[2];
                gather_totals<2>(S.red, W, tot);
                ++phase;
                const double c = tot[0], bma = tot[1], hsdSq = alpha * alpha * bb;
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
            bool changed = false;
#pragma unroll
            for (int s = 0; s < M; ++s) {
                if (!valid[s]) { Xn[s] = X[s]; continue; }
                double dl[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    if (stepType == 0) dl[k] = h[s][k];
                    else if (stepType == 1) dl[k] = sdScale * (alpha * b[s][k]);
                    else { const double sk = alpha * b[s][k]; dl[k] = sk + beta * (h[s][k] - sk); }
                }
                // VertexSE3::oplusImpl: X <- X * fromVectorMQT(dl)
                double wq = 1.0 - (dl[3] * dl[3] + dl[4] * dl[4] + dl[5] * dl[5]);
                double dR[9];
                if (wq < 0) { R_from_quat(1.0, 0.0, 0.0, 0.0, dR); }
                else { wq = sqrt(wq); R_from_quat(wq, dl[3], dl[4], dl[5], dR); }
                m3_mul(X[s].R, dR, Xn[s].R);
                double rt3[3];
                m3_vec(X[s].R, dl, rt3);
#pragma unroll
                for (int k = 0; k < 3; ++k) Xn[s].t[k] = X[s].t[k] + rt3[k];
#pragma unroll
                for (int k = 0; k < 9; ++k) changed |= Xn[s].R[k] != X[s].R[k];
#pragma unroll
                for (int k = 0; k < 3; ++k) changed |= Xn[s].t[k] != X[s].t[k];
            }
            const int trial = cur ^ 1;
            bool anyChanged;
            const double newChi = evaluate(Xn, std::integral_constant<int, 0>{}, trial, changed, anyChanged);
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
                edge = edgeN;
#pragma unroll
                for (int s = 0; s < M; ++s) {
                    X[s] = Xn[s];
#pragma unroll
                    for (int k = 0; k < 3; ++k) asm volatile("" : "+v"(X[s].t[k]));   // (hidden from CSE with the trial pass)
                }
                {   // errors of the new committed state (same arithmetic as the trial pass, no barrier)
                    Pose3 An[M];
                    prev_pose(X, edge, An);
#pragma unroll
                    for (int s = 0; s < M; ++s) {
                        if (!valid[s]) continue;
                        double Rz[9], tz[3];
                        ld_rz(s, Rz, tz);
                        Edge3 E;
                        se3_edge(An[s], X[s], Rz, tz, E);
#pragma unroll
                        for (int k = 0; k < 6; ++k) e[s][k] = E.e[k];
                    }
                }
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
        if (numTries == maxTrials || !goodStep) { flags |= 1; break; }
    }

    // ---- per-edge chi2 ----
    double mx = 0.0;
    bool nan = false;
#pragma unroll
    for (int s = 0; s < M; ++s) {
        if (!valid[s]) continue;
        double om[21];
        ld_sym(G_OM, s, om);
        const double c = sym6_quad(om, e[s]);
        if (c != c) nan = true;
        else mx = fmax(mx, c);
    }
    mx = wave_max(mx);
    {
        Se3Scratch<W, NL>& S = sh.scr[phase & 1];
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
