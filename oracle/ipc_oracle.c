/*
 * oracle/ipc_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * Plain-C restatement of the reference's consistency-check hot path, used only as the
 * parity checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * Nothing under ipc_amd/ may include, link or call this file.
 *
 * PARITY UNPINNED at the g2o boundary: all floating-point arithmetic of the reference's
 * path lives in g2o tag 20201223_git (README.md:7 of the reference), which is neither
 * vendored under /root/reference nor installed in this image, and the reference has no
 * tests / golden vectors.  The g2o parts below are restated
 * from the published algorithm (upstream file names given per function) and are pinned by
 * closed-form known-answer tests instead (tests/test_oracle_kat.py).
 *
 * What is restated, with the reference call sites each part follows:
 *   - normalize_theta, SE2 compose/inverse, EdgeSE2 error + analytic Jacobians, VertexSE2
 *     oplus                    [g2o stuff/misc.h, types/slam2d/{se2.h,edge_se2.cpp,vertex_se2.h}]
 *   - Isometry3 <-> (t, quaternion-vector) maps, EdgeSE3 error + analytic Jacobians,
 *     VertexSE3 oplus          [g2o types/slam3d/{isometry3d_mappings.cpp,edge_se3.cpp,
 *                               isometry3d_gradients.h,vertex_se3.h}]
 *   - BaseBinaryEdge::constructQuadraticForm, BlockSolver::buildSystem, sparse Cholesky,
 *     multiplyHessian          [g2o core/{base_binary_edge.hpp,block_solver.hpp},
 *                               solvers/eigen/linear_solver_eigen.h]
 *   - OptimizationAlgorithmDogleg::solve and SparseOptimizer::optimize
 *                              [g2o core/{optimization_algorithm_dogleg.cpp,sparse_optimizer.cpp}]
 *     selected by the reference at src/utils.cpp:105 ("dl_var")
 *   - isAgreeingWithCurrentState            reference src/consensus_utils.cpp:7-22
 *   - fixComplementary (gauge = pose lo)    reference src/consensus_utils.cpp:29-43
 *   - propagateGuess / propagateCurrentGuess reference src/consensus_utils.cpp:61-71,99-116
 *   - robustifyVoters (info *= s)           reference src/consensus_utils.cpp:124-130
 *   - computeIndependentSubgraph            reference src/consensus.cpp:124-171
 *   - agreementCheck                        reference src/consensus.cpp:43-75
 *   - cmpTime candidate ordering            reference src/utils.cpp:379-390, src/simulation.cpp:26
 *   - pair cell P1 / set-max P2             SURVEY.md section 8a rows P1, P2 (the re-formulation
 *                                           the north star asks for; no reference counterpart)
 *
 * Build: gcc -O3 -fPIC -shared (no -ffast-math, no -march: plain IEEE double like the
 * reference's "-std=c++14 -O3", CMakeLists.txt:4).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#define ORACLE_PI 3.14159265358979323846

/* ------------------------------------------------------------------------------------ */
/* g2o stuff/misc.h : normalize_theta  (range [-pi, pi))                                  */
/* ------------------------------------------------------------------------------------ */
double oracle_normalize_theta(double theta)
{
    if (theta >= -ORACLE_PI && theta < ORACLE_PI) return theta;
    double multiplier = floor(theta / (2 * ORACLE_PI));
    theta = theta - multiplier * 2 * ORACLE_PI;
    if (theta >= ORACLE_PI) theta -= 2 * ORACLE_PI;
    if (theta < -ORACLE_PI) theta += 2 * ORACLE_PI;
    return theta;
}

/* ------------------------------------------------------------------------------------ */
/* pose storage: dim 2 -> 3 doubles (x, y, theta); dim 3 -> 12 doubles (R row-major, t)   */
/* error/tangent dimension d = 3 (SE2) or 6 (SE3)                                        */
/* ------------------------------------------------------------------------------------ */
static int pose_size(int dim) { return dim == 2 ? 3 : 12; }
static int tan_dim(int dim) { return dim == 2 ? 3 : 6; }
static int meas_size(int dim) { return dim == 2 ? 3 : 7; }
static int info_size(int dim) { return dim == 2 ? 6 : 21; }

/* ---- SE2 (g2o types/slam2d/se2.h) ---- */
static void se2_mul(const double *a, const double *b, double *r)
{
    double c = cos(a[2]), s = sin(a[2]);
    double x = a[0] + (c * b[0] - s * b[1]);
    double y = a[1] + (s * b[0] + c * b[1]);
    double th = oracle_normalize_theta(a[2] + b[2]);
    r[0] = x; r[1] = y; r[2] = th;
}
static void se2_inv(const double *a, double *r)
{
    double th = oracle_normalize_theta(-a[2]);
    double c = cos(th), s = sin(th);
    double x = c * (-a[0]) - s * (-a[1]);
    double y = s * (-a[0]) + c * (-a[1]);
    r[0] = x; r[1] = y; r[2] = th;
}

/* ---- SE3 as (R,t) (Eigen::Isometry3d semantics) ---- */
static void se3_mul(const double *a, const double *b, double *r)
{
    double R[9], t[3];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j)
            R[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
        t[i] = a[3 * i] * b[9] + a[3 * i + 1] * b[10] + a[3 * i + 2] * b[11] + a[9 + i];
    }
    memcpy(r, R, sizeof R); memcpy(r + 9, t, sizeof t);
}
static void se3_inv(const double *a, double *r)
{
    double R[9], t[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[3 * i + j] = a[3 * j + i];
    for (int i = 0; i < 3; ++i)
        t[i] = -(R[3 * i] * a[9] + R[3 * i + 1] * a[10] + R[3 * i + 2] * a[11]);
    memcpy(r, R, sizeof R); memcpy(r + 9, t, sizeof t);
}
/* Eigen Quaternion::toRotationMatrix for q = (w, x, y, z) */
static void quat_to_R(double w, double x, double y, double z, double *R)
{
    double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    double twx = tx * w, twy = ty * w, twz = tz * w;
    double txx = tx * x, txy = ty * x, txz = tz * x;
    double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
/* Eigen Quaternion(Matrix3) followed by g2o internal::normalize (unit norm, w >= 0);
 * q = (x, y, z, w) */
static void R_to_quat_normalized(const double *R, double *q)
{
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t;
        q[1] = (R[2] - R[6]) * t;
        q[2] = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (R[3 * k + j] - R[3 * j + k]) * t;
        q[j] = (R[3 * j + i] + R[3 * i + j]) * t;
        q[k] = (R[3 * k + i] + R[3 * i + k]) * t;
    }
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) q[i] /= n;
    if (q[3] < 0) for (int i = 0; i < 4; ++i) q[i] = -q[i];
}
/* g2o internal::fromVectorMQT: (t, qx, qy, qz) -> Isometry3 */
static void se3_from_mqt(const double *v, double *X)
{
    double w = 1 - (v[3] * v[3] + v[4] * v[4] + v[5] * v[5]);
    if (w < 0) {
        for (int i = 0; i < 9; ++i) X[i] = (i % 4 == 0) ? 1.0 : 0.0;
    } else {
        w = sqrt(w);
        quat_to_R(w, v[3], v[4], v[5], X);
    }
    X[9] = v[0]; X[10] = v[1]; X[11] = v[2];
}
/* g2o internal::toVectorMQT */
static void se3_to_mqt(const double *X, double *v)
{
    double q[4];
    R_to_quat_normalized(X, q);
    v[0] = X[9]; v[1] = X[10]; v[2] = X[11];
    v[3] = q[0]; v[4] = q[1]; v[5] = q[2];
}

/* measurement as read from a g2o file -> pose storage.
 * EdgeSE2::read: SE2(x, y, theta) as-is.  EdgeSE3::read: quaternion (qx qy qz qw)
 * re-normalised, then fromVectorQT. */
void oracle_meas_to_pose(int dim, const double *m, double *X)
{
    if (dim == 2) { X[0] = m[0]; X[1] = m[1]; X[2] = m[2]; return; }
    double n = sqrt(m[3] * m[3] + m[4] * m[4] + m[5] * m[5] + m[6] * m[6]);
    quat_to_R(m[6] / n, m[3] / n, m[4] / n, m[5] / n, X);
    X[9] = m[0]; X[10] = m[1]; X[11] = m[2];
}
/* information: upper-triangular row order as in the file -> full d x d */
void oracle_info_to_full(int dim, const double *u, double *F)
{
    int d = tan_dim(dim), k = 0;
    for (int i = 0; i < d; ++i)
        for (int j = i; j < d; ++j) { F[i * d + j] = u[k]; F[j * d + i] = u[k]; ++k; }
}

void oracle_pose_mul(int dim, const double *a, const double *b, double *r)
{ if (dim == 2) se2_mul(a, b, r); else se3_mul(a, b, r); }
void oracle_pose_inv(int dim, const double *a, double *r)
{ if (dim == 2) se2_inv(a, r); else se3_inv(a, r); }
void oracle_pose_identity(int dim, double *X)
{
    if (dim == 2) { X[0] = X[1] = X[2] = 0; return; }
    for (int i = 0; i < 12; ++i) X[i] = 0;
    X[0] = X[4] = X[8] = 1;
}

/* EdgeSE2::computeError / EdgeSE3::computeError:
 *   e = toVector(Zinv * (Xi^-1 * Xj))                                                   */
void oracle_edge_error(int dim, const double *Zinv, const double *Xi, const double *Xj, double *e)
{
    double a[12], b[12], c[12];
    oracle_pose_inv(dim, Xi, a);
    oracle_pose_mul(dim, a, Xj, b);
    oracle_pose_mul(dim, Zinv, b, c);
    if (dim == 2) { e[0] = c[0]; e[1] = c[1]; e[2] = c[2]; }
    else se3_to_mqt(c, e);
}

/* VertexSE2::oplusImpl (t += dt, theta = normalize(theta + dtheta)) /
 * VertexSE3::oplusImpl (X = X * fromVectorMQT(delta)).  The every-1000-calls
 * re-orthogonalisation of VertexSE3 is a rounding-level repair and is not restated. */
void oracle_pose_oplus(int dim, double *X, const double *delta)
{
    if (dim == 2) {
        X[0] += delta[0]; X[1] += delta[1];
        X[2] = oracle_normalize_theta(X[2] + delta[2]);
    } else {
        double inc[12], r[12];
        se3_from_mqt(delta, inc);
        se3_mul(X, inc, r);
        memcpy(X, r, sizeof r);
    }
}

/* EdgeSE2::linearizeOplus (analytic, g2o types/slam2d/edge_se2.cpp) and the analytic
 * derivative EdgeSE3::linearizeOplus evaluates (isometry3d_gradients.h computes the exact
 * derivative of toVectorMQT(Z^-1 (Xi*Di)^-1 (Xj*Dj)); here in closed form:
 *   B = blockdiag(R_E, w_E I + [v_E]x),  A = -B * [[Rab^T, -2 Rab^T [tab]x], [0, Rab^T]]
 * with E = Zinv*Xi^-1*Xj, (w_E, v_E) its normalised quaternion, Xab = Xi^-1 Xj).
 * A, B are d x d row-major: de/dXi, de/dXj. */
void oracle_edge_jacobians(int dim, const double *Z, const double *Zinv, const double *Xi,
                           const double *Xj, double *A, double *B)
{
    if (dim == 2) {
        double thetai = Xi[2];
        double dx = Xj[0] - Xi[0], dy = Xj[1] - Xi[1];
        double si = sin(thetai), ci = cos(thetai);
        double Ji[9] = { -ci, -si, -si * dx + ci * dy,
                          si, -ci, -ci * dx - si * dy,
                          0, 0, -1 };
        double Jj[9] = { ci, si, 0, -si, ci, 0, 0, 0, 1 };
        double cz = cos(Zinv[2]), sz = sin(Zinv[2]);
        double z[9] = { cz, -sz, 0, sz, cz, 0, 0, 0, 1 };
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double sa = 0, sb = 0;
                for (int k = 0; k < 3; ++k) { sa += z[3 * i + k] * Ji[3 * k + j]; sb += z[3 * i + k] * Jj[3 * k + j]; }
                A[3 * i + j] = sa; B[3 * i + j] = sb;
            }
        (void)Z;
        return;
    }
    double Xiinv[12], Xab[12], E[12], q[4];
    se3_inv(Xi, Xiinv);
    se3_mul(Xiinv, Xj, Xab);
    se3_mul(Zinv, Xab, E);
    R_to_quat_normalized(E, q);
    double D[36];
    memset(D, 0, sizeof D);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) D[6 * i + j] = E[3 * i + j];
    double w = q[3], vx = q[0], vy = q[1], vz = q[2];
    double Q[9] = { w, -vz, vy, vz, w, -vx, -vy, vx, w };
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) D[6 * (3 + i) + 3 + j] = Q[3 * i + j];
    /* T = [[Rab^T, -2 Rab^T [tab]x],[0, Rab^T]] */
    double T[36];
    memset(T, 0, sizeof T);
    double tx = Xab[9], ty = Xab[10], tz = Xab[11];
    double S[9] = { 0, -tz, ty, tz, 0, -tx, -ty, tx, 0 };
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double rt = Xab[3 * j + i];
            T[6 * i + j] = rt;
            T[6 * (3 + i) + 3 + j] = rt;
            double s = 0;
            for (int k = 0; k < 3; ++k) s += Xab[3 * k + i] * S[3 * k + j];
            T[6 * i + 3 + j] = -2 * s;
        }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0;
            for (int k = 0; k < 6; ++k) s += D[6 * i + k] * T[6 * k + j];
            A[6 * i + j] = -s;
            B[6 * i + j] = D[6 * i + j];
        }
    (void)Z;
}

/* ------------------------------------------------------------------------------------ */
/* propagateGuess (consensus_utils.cpp:99-116): vertex 0 at origin, v[i] = v[i-1]*z[i-1]  */
/* ------------------------------------------------------------------------------------ */
void oracle_propagate(int dim, int V, const double *odom_meas, double *poses)
{
    int ps = pose_size(dim), ms = meas_size(dim);
    double Z[12];
    oracle_pose_identity(dim, poses);
    for (int i = 1; i < V; ++i) {
        oracle_meas_to_pose(dim, odom_meas + (size_t)(i - 1) * ms, Z);
        oracle_pose_mul(dim, poses + (size_t)(i - 1) * ps, Z, poses + (size_t)i * ps);
    }
}

/* ------------------------------------------------------------------------------------ */
/* The sub-problem: poses 0..L (pose 0 = gauge, fixed: fixComplementary) and ne edges.    */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    int dim, d, ps;
    int L, ne;
    int *from, *to;      /* local pose indices 0..L                                   */
    double *Z, *Zinv;    /* ne * ps                                                    */
    double *info;        /* ne * d * d (already scaled by s for odometry)              */
    double *X;           /* (L+1) * ps current estimate                                */
    double *Xbak;        /* push()/pop() copy                                          */
    double *err;         /* ne * d                                                     */
    /* linear system (skyline, lower triangle, scalar granularity, natural order) */
    int n;
    int *first;          /* first stored column of row r                               */
    size_t *rowp;        /* offset of row r in val                                     */
    size_t nnz;
    double *H, *Lf, *b, *x;
    double *hsd, *hdl, *aux;
} sub_t;

typedef struct {
    int iterations;      /* outer iterations executed                                  */
    int tries_total;     /* sum of numTries                                            */
    int terminated;      /* 1 if dog-leg returned Terminate, 2 if Fail                 */
    double chi2_initial, chi2_final;
} oracle_stats_t;

static double *sk(sub_t *s, double *M, int r, int c) { return M + s->rowp[r] + (c - s->first[r]); }

static void sub_free(sub_t *s)
{
    free(s->from); free(s->to); free(s->Z); free(s->Zinv); free(s->info); free(s->X);
    free(s->Xbak); free(s->err); free(s->first); free(s->rowp); free(s->H); free(s->Lf);
    free(s->b); free(s->x); free(s->hsd); free(s->hdl); free(s->aux);
}

/* BlockSolver::buildStructure: index mapping in vertex-id order; the block pattern of H is
 * the chain plus one block per edge with two free ends.  Storage here is a skyline
 * (fill stays inside it), ordering is natural; g2o's AMD ordering changes fill only. */
static void sub_build_structure(sub_t *s)
{
    int d = s->d, L = s->L;
    int *fb = (int *)malloc(sizeof(int) * (size_t)(L > 0 ? L : 1));
    for (int p = 0; p < L; ++p) fb[p] = p;
    for (int e = 0; e < s->ne; ++e) {
        int f = s->from[e], t = s->to[e];
        if (f == 0 || t == 0) continue;
        int lo = (f < t ? f : t) - 1, hi = (f < t ? t : f) - 1;
        if (lo < fb[hi]) fb[hi] = lo;
    }
    s->n = d * L;
    s->first = (int *)malloc(sizeof(int) * (size_t)(s->n + 1));
    s->rowp = (size_t *)malloc(sizeof(size_t) * (size_t)(s->n + 1));
    size_t off = 0;
    for (int r = 0; r < s->n; ++r) {
        s->first[r] = d * fb[r / d];
        s->rowp[r] = off;
        off += (size_t)(r - s->first[r] + 1);
    }
    s->rowp[s->n] = off;
    s->nnz = off;
    free(fb);
    s->H = (double *)malloc(sizeof(double) * (off + 1));
    s->Lf = (double *)malloc(sizeof(double) * (off + 1));
    size_t nv = (size_t)(s->n + 1);
    s->b = (double *)calloc(nv, sizeof(double));
    s->x = (double *)calloc(nv, sizeof(double));
    s->hsd = (double *)calloc(nv, sizeof(double));
    s->hdl = (double *)calloc(nv, sizeof(double));
    s->aux = (double *)calloc(nv, sizeof(double));
}

/* SparseOptimizer::computeActiveErrors + activeRobustChi2 (no robust kernel) */
static double sub_compute_errors(sub_t *s)
{
    int d = s->d, ps = s->ps;
    double chi = 0;
    for (int e = 0; e < s->ne; ++e) {
        double *er = s->err + (size_t)e * d;
        oracle_edge_error(s->dim, s->Zinv + (size_t)e * ps, s->X + (size_t)s->from[e] * ps,
                          s->X + (size_t)s->to[e] * ps, er);
    }
    for (int e = 0; e < s->ne; ++e) {
        const double *er = s->err + (size_t)e * d, *om = s->info + (size_t)e * d * d;
        double c = 0;
        for (int i = 0; i < d; ++i) {
            double t = 0;
            for (int j = 0; j < d; ++j) t += om[i * d + j] * er[j];
            c += er[i] * t;
        }
        chi += c;
    }
    return chi;
}
static double edge_chi2(const sub_t *s, int e)
{
    int d = s->d;
    const double *er = s->err + (size_t)e * d, *om = s->info + (size_t)e * d * d;
    double c = 0;
    for (int i = 0; i < d; ++i) {
        double t = 0;
        for (int j = 0; j < d; ++j) t += om[i * d + j] * er[j];
        c += er[i] * t;
    }
    return c;
}

/* BlockSolver::buildSystem: zero H and b; per active edge linearizeOplus +
 * BaseBinaryEdge::constructQuadraticForm (b -= J^T Omega e, H += J^T Omega J; fixed
 * vertices skipped). */
static void sub_build_system(sub_t *s)
{
    int d = s->d, ps = s->ps;
    memset(s->H, 0, sizeof(double) * s->nnz);
    memset(s->b, 0, sizeof(double) * (size_t)s->n);
    double A[36], B[36], AtO[36], BtO[36], omr[6];
    for (int e = 0; e < s->ne; ++e) {
        int f = s->from[e], t = s->to[e];
        const double *om = s->info + (size_t)e * d * d, *er = s->err + (size_t)e * d;
        oracle_edge_jacobians(s->dim, s->Z + (size_t)e * ps, s->Zinv + (size_t)e * ps,
                              s->X + (size_t)f * ps, s->X + (size_t)t * ps, A, B);
        for (int i = 0; i < d; ++i) {
            double v = 0;
            for (int j = 0; j < d; ++j) v += om[i * d + j] * er[j];
            omr[i] = -v;
        }
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) {
                double sa = 0, sb = 0;
                for (int k = 0; k < d; ++k) { sa += A[k * d + i] * om[k * d + j]; sb += B[k * d + i] * om[k * d + j]; }
                AtO[i * d + j] = sa; BtO[i * d + j] = sb;
            }
        if (f != 0) {
            int r0 = (f - 1) * d;
            for (int i = 0; i < d; ++i) {
                double v = 0;
                for (int k = 0; k < d; ++k) v += A[k * d + i] * omr[k];
                s->b[r0 + i] += v;
                for (int j = 0; j <= i; ++j) {
                    double h = 0;
                    for (int k = 0; k < d; ++k) h += AtO[i * d + k] * A[k * d + j];
                    *sk(s, s->H, r0 + i, r0 + j) += h;
                }
            }
        }
        if (t != 0) {
            int r0 = (t - 1) * d;
            for (int i = 0; i < d; ++i) {
                double v = 0;
                for (int k = 0; k < d; ++k) v += B[k * d + i] * omr[k];
                s->b[r0 + i] += v;
                for (int j = 0; j <= i; ++j) {
                    double h = 0;
                    for (int k = 0; k < d; ++k) h += BtO[i * d + k] * B[k * d + j];
                    *sk(s, s->H, r0 + i, r0 + j) += h;
                }
            }
        }
        if (f != 0 && t != 0) {
            /* block (f,t) = A^T Omega B ; stored in the lower triangle */
            int rf = (f - 1) * d, rt = (t - 1) * d;
            for (int i = 0; i < d; ++i)
                for (int j = 0; j < d; ++j) {
                    double h = 0;
                    for (int k = 0; k < d; ++k) h += AtO[i * d + k] * B[k * d + j];
                    if (rt > rf) *sk(s, s->H, rt + j, rf + i) += h;
                    else         *sk(s, s->H, rf + i, rt + j) += h;
                }
        }
    }
}

/* SparseBlockMatrix::multiplySymmetricUpperTriangle: dest += H * src */
static void sub_multiply_hessian(sub_t *s, double *dest, const double *src)
{
    for (int r = 0; r < s->n; ++r) {
        const double *row = s->H + s->rowp[r];
        int f = s->first[r];
        double acc = 0;
        for (int c = f; c < r; ++c) {
            double v = row[c - f];
            acc += v * src[c];
            dest[c] += v * src[r];
        }
        acc += row[r - f] * src[r];
        dest[r] += acc;
    }
}

static int g_wide_dots = 0;
void oracle_set_wide_dots(int on) { g_wide_dots = on; }

/* LinearSolverEigen::solve: LL^T of H (+lambda on the diagonal when asked), fails on a
 * non-positive pivot like SimplicialLLT; then two triangular solves. */
static int sub_solve(sub_t *s, double lambda, int add_lambda)
{
    int n = s->n;
    memcpy(s->Lf, s->H, sizeof(double) * s->nnz);
    if (add_lambda)
        for (int r = 0; r < n; ++r) *sk(s, s->Lf, r, r) += lambda;
    for (int i = 0; i < n; ++i) {
        double *ri = s->Lf + s->rowp[i];
        int fi = s->first[i];
        for (int j = fi; j <= i; ++j) {
            const double *rj = s->Lf + s->rowp[j];
            int fj = s->first[j];
            int k0 = fi > fj ? fi : fj;
            double sum = ri[j - fi];
            if (g_wide_dots && j - k0 >= 16) {
                /* (oracle_set_wide_dots: the same envelope factorisation, the dot product over the shared columns
                 * accumulated in eight independent partial sums instead of one serial chain -- rounding-level
                 * different, several times faster; for the late states of C4 / C5, whose systems have 15 000 to
                 * 180 000 unknowns, tests/golden/make_late_state_golden.py) */
                const double *a = ri + (k0 - fi), *b = rj + (k0 - fj);
                int len = j - k0, k = 0;
                double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
                for (; k + 8 <= len; k += 8) {
                    s0 += a[k] * b[k];         s1 += a[k + 1] * b[k + 1];
                    s2 += a[k + 2] * b[k + 2]; s3 += a[k + 3] * b[k + 3];
                    s4 += a[k + 4] * b[k + 4]; s5 += a[k + 5] * b[k + 5];
                    s6 += a[k + 6] * b[k + 6]; s7 += a[k + 7] * b[k + 7];
                }
                for (; k < len; ++k) s0 += a[k] * b[k];
                sum -= ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
            } else
            for (int k = k0; k < j; ++k) sum -= ri[k - fi] * rj[k - fj];
            if (j < i) ri[j - fi] = sum / rj[j - fj];
            else {
                if (!(sum > 0)) return 0;
                ri[i - fi] = sqrt(sum);
            }
        }
    }
    for (int i = 0; i < n; ++i) {
        const double *ri = s->Lf + s->rowp[i];
        int fi = s->first[i];
        double sum = s->b[i];
        for (int k = fi; k < i; ++k) sum -= ri[k - fi] * s->x[k];
        s->x[i] = sum / ri[i - fi];
    }
    for (int i = n - 1; i >= 0; --i) {
        const double *ri = s->Lf + s->rowp[i];
        int fi = s->first[i];
        double xi = s->x[i] / ri[i - fi];
        s->x[i] = xi;
        for (int k = fi; k < i; ++k) s->x[k] -= ri[k - fi] * xi;
    }
    return 1;
}

static double vdot(const double *a, const double *b, int n)
{ double s = 0; for (int i = 0; i < n; ++i) s += a[i] * b[i]; return s; }

/* SparseOptimizer::update: oplus on every non-fixed active vertex */
static void sub_update(sub_t *s, const double *h)
{
    for (int p = 1; p <= s->L; ++p)
        oracle_pose_oplus(s->dim, s->X + (size_t)p * s->ps, h + (size_t)(p - 1) * s->d);
}

/* SparseOptimizer::optimize(iterations) driving OptimizationAlgorithmDogleg::solve
 * (initialDelta 1e4, maxTrialsAfterFailure 100, initialLambda 1e-7, lambdaFactor 10). */
static void sub_optimize(sub_t *s, int iterations, oracle_stats_t *st)
{
    int n = s->n;
    double delta = 1e4, currentLambda = 1e-7;
    const double lambdaFactor = 10.0;
    const int maxTrials = 100;
    int wasPD = 1;
    st->iterations = 0; st->tries_total = 0; st->terminated = 0;
    if (n == 0) return;
    for (int it = 0; it < iterations; ++it) {
        double currentChi = sub_compute_errors(s);
        if (it == 0) st->chi2_initial = currentChi;
        sub_build_system(s);
        memset(s->aux, 0, sizeof(double) * (size_t)n);
        sub_multiply_hessian(s, s->aux, s->b);
        double bNormSquared = vdot(s->b, s->b, n);
        double alpha = bNormSquared / vdot(s->aux, s->b, n);
        for (int i = 0; i < n; ++i) s->hsd[i] = alpha * s->b[i];
        double hsdNorm = sqrt(vdot(s->hsd, s->hsd, n));
        double hgnNorm = -1.0;
        int solvedGN = 0, goodStep = 0, numTries = 0, failed = 0;
        do {
            ++numTries;
            if (!solvedGN) {
                const double minLambda = 1e-12, maxLambda = 1e3;
                solvedGN = 1;
                int solverOk = 0;
                while (!solverOk) {
                    solverOk = sub_solve(s, currentLambda, !wasPD);
                    wasPD = wasPD && solverOk;
                    if (!wasPD) {
                        if (solverOk) {
                            double c = currentLambda / (0.5 * lambdaFactor);
                            currentLambda = c > minLambda ? c : minLambda;
                        } else {
                            currentLambda *= lambdaFactor;
                            if (currentLambda > maxLambda) { currentLambda = maxLambda; failed = 1; break; }
                        }
                    }
                }
                if (failed) break;
                hgnNorm = sqrt(vdot(s->x, s->x, n));
            }
            const double *hgn = s->x;
            if (hgnNorm < delta) {
                memcpy(s->hdl, hgn, sizeof(double) * (size_t)n);
            } else if (hsdNorm > delta) {
                double f = delta / hsdNorm;
                for (int i = 0; i < n; ++i) s->hdl[i] = f * s->hsd[i];
            } else {
                for (int i = 0; i < n; ++i) s->aux[i] = hgn[i] - s->hsd[i];
                double c = vdot(s->hsd, s->aux, n);
                double bmaSquaredNorm = vdot(s->aux, s->aux, n);
                double beta;
                if (c <= 0.)
                    beta = (-c + sqrt(c * c + bmaSquaredNorm * (delta * delta - vdot(s->hsd, s->hsd, n)))) / bmaSquaredNorm;
                else {
                    double hsdSqrNorm = vdot(s->hsd, s->hsd, n);
                    beta = (delta * delta - hsdSqrNorm) / (c + sqrt(c * c + bmaSquaredNorm * (delta * delta - hsdSqrNorm)));
                }
                for (int i = 0; i < n; ++i) s->hdl[i] = s->hsd[i] + beta * (hgn[i] - s->hsd[i]);
            }
            memset(s->aux, 0, sizeof(double) * (size_t)n);
            sub_multiply_hessian(s, s->aux, s->hdl);
            double linearGain = -1 * vdot(s->aux, s->hdl, n) + 2 * vdot(s->b, s->hdl, n);
            memcpy(s->Xbak, s->X, sizeof(double) * (size_t)(s->L + 1) * s->ps);   /* push   */
            sub_update(s, s->hdl);
            double newChi = sub_compute_errors(s);
            double nonLinearGain = currentChi - newChi;
            if (fabs(linearGain) < 1e-12) linearGain = 1e-12;
            double rho = nonLinearGain / linearGain;
            if (rho > 0) {                                                          /* discardTop */
                goodStep = 1;
            } else {
                memcpy(s->X, s->Xbak, sizeof(double) * (size_t)(s->L + 1) * s->ps); /* pop */
            }
            double hdlNorm = sqrt(vdot(s->hdl, s->hdl, n));
            if (rho > 0.75) { double c3 = 3 * hdlNorm; delta = delta > c3 ? delta : c3; }
            else if (rho < 0.25) delta *= 0.5;
        } while (!goodStep && numTries < maxTrials);
        st->iterations = it + 1;
        st->tries_total += numTries;
        if (failed) { st->terminated = 2; break; }
        if (numTries == maxTrials || !goodStep) { st->terminated = 1; break; }
    }
}

/* ------------------------------------------------------------------------------------ */
/* Sub-problem construction: chain [lo,hi] of the global graph + nl loop edges.           */
/* odom_info is the UN-scaled file information; s_factor is applied here                  */
/* (robustifyVoters, consensus_utils.cpp:124-130).                                        */
/* ------------------------------------------------------------------------------------ */
static void sub_init(sub_t *s, int dim, const double *odom_meas, const double *odom_info,
                     double s_factor, const double *poses, int lo, int hi, int nl,
                     const int *loop_ids, const double *loop_meas, const double *loop_info)
{
    memset(s, 0, sizeof *s);
    s->dim = dim; s->d = tan_dim(dim); s->ps = pose_size(dim);
    s->L = hi - lo; s->ne = s->L + nl;
    int d = s->d, ps = s->ps, ms = meas_size(dim), is = info_size(dim);
    size_t ne = (size_t)(s->ne > 0 ? s->ne : 1);
    s->from = (int *)malloc(sizeof(int) * ne);
    s->to = (int *)malloc(sizeof(int) * ne);
    s->Z = (double *)malloc(sizeof(double) * ne * ps);
    s->Zinv = (double *)malloc(sizeof(double) * ne * ps);
    s->info = (double *)malloc(sizeof(double) * ne * d * d);
    s->err = (double *)malloc(sizeof(double) * ne * d);
    s->X = (double *)malloc(sizeof(double) * (size_t)(s->L + 1) * ps);
    s->Xbak = (double *)malloc(sizeof(double) * (size_t)(s->L + 1) * ps);
    memcpy(s->X, poses + (size_t)lo * ps, sizeof(double) * (size_t)(s->L + 1) * ps);
    for (int j = 0; j < s->L; ++j) {
        s->from[j] = j; s->to[j] = j + 1;
        oracle_meas_to_pose(dim, odom_meas + (size_t)(lo + j) * ms, s->Z + (size_t)j * ps);
        oracle_pose_inv(dim, s->Z + (size_t)j * ps, s->Zinv + (size_t)j * ps);
        oracle_info_to_full(dim, odom_info + (size_t)(lo + j) * is, s->info + (size_t)j * d * d);
        for (int k = 0; k < d * d; ++k) s->info[(size_t)j * d * d + k] *= s_factor;
    }
    for (int l = 0; l < nl; ++l) {
        int e = s->L + l;
        s->from[e] = loop_ids[2 * l] - lo; s->to[e] = loop_ids[2 * l + 1] - lo;
        oracle_meas_to_pose(dim, loop_meas + (size_t)l * ms, s->Z + (size_t)e * ps);
        oracle_pose_inv(dim, s->Z + (size_t)e * ps, s->Zinv + (size_t)e * ps);
        oracle_info_to_full(dim, loop_info + (size_t)l * is, s->info + (size_t)e * d * d);
    }
    sub_build_structure(s);
}

/* isAgreeingWithCurrentState (consensus_utils.cpp:7-22) on an explicit sub-problem:
 * optimize(iter_base * (5 if #edges > 100)), computeActiveErrors, per-edge chi2.
 * chi2_out (size L + nl, odometry first, then loops) and poses_out ((L+1)*ps) may be NULL.
 * Returns max edge chi2. */
double oracle_solve_cell(int dim, const double *odom_meas, const double *odom_info,
                         double s_factor, const double *poses, int lo, int hi, int nl,
                         const int *loop_ids, const double *loop_meas, const double *loop_info,
                         int iter_base, double *chi2_out, double *poses_out, oracle_stats_t *stats)
{
    sub_t s;
    oracle_stats_t st;
    memset(&st, 0, sizeof st);
    sub_init(&s, dim, odom_meas, odom_info, s_factor, poses, lo, hi, nl, loop_ids, loop_meas, loop_info);
    st.chi2_initial = sub_compute_errors(&s);                    /* consensus_utils.cpp:11 */
    int iter = iter_base;
    iter = s.ne > 100 ? iter * 5 : iter;                          /* consensus_utils.cpp:12-13 */
    sub_optimize(&s, iter, &st);                                  /* consensus_utils.cpp:14 */
    st.chi2_final = sub_compute_errors(&s);                       /* consensus_utils.cpp:15 */
    /* consensus_utils.cpp:17-19 returns false as soon as ONE edge has chi2 > th, and a NaN chi2 is not "> th": the  */
    /* decision is (max over the edges that have a number) > th.  NaN is reported only when no edge has a positive   */
    /* number (the check agrees either way).  Rounds 1-3 let one NaN edge mask the others.                           */
    double mx = 0;
    int any_nan = 0;
    for (int e = 0; e < s.ne; ++e) {
        double c = edge_chi2(&s, e);
        if (chi2_out) chi2_out[e] = c;
        if (c != c) any_nan = 1;
        else if (c > mx) mx = c;
    }
    if (any_nan && !(mx > 0)) mx = NAN;
    if (poses_out) memcpy(poses_out, s.X, sizeof(double) * (size_t)(s.L + 1) * s.ps);
    if (stats) *stats = st;
    sub_free(&s);
    return mx;
}

/* ------------------------------------------------------------------------------------ */
/* Candidate ordering: cmpTime (utils.cpp:379-390) = ascending max(id0,id1); std::sort    */
/* leaves ties unspecified, the build fixes (max id, file index).                         */
/* ------------------------------------------------------------------------------------ */
typedef struct { int key, idx; } okey_t;
static int okey_cmp(const void *a, const void *b)
{
    const okey_t *x = (const okey_t *)a, *y = (const okey_t *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}
void oracle_candidate_order(int N, const int *ids, int *order)
{
    okey_t *k = (okey_t *)malloc(sizeof(okey_t) * (size_t)(N > 0 ? N : 1));
    for (int i = 0; i < N; ++i) {
        k[i].key = ids[2 * i] > ids[2 * i + 1] ? ids[2 * i] : ids[2 * i + 1];
        k[i].idx = i;
    }
    qsort(k, (size_t)N, sizeof(okey_t), okey_cmp);
    for (int i = 0; i < N; ++i) order[i] = k[i].idx;
    free(k);
}

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* ------------------------------------------------------------------------------------ */
/* Pair cell P1: returns max chi2 of the cell (i,j); i == j = diagonal.    */
/* *solved = 0 when the intervals do not overlap with positive length                     */
/* (consensus.cpp:157-159 rule) -- then the cell is the AND of the two diagonals and the  */
/* return value is max(diag_i, diag_j) supplied by the caller through diag (may be NULL). */
/* ------------------------------------------------------------------------------------ */
double oracle_pair_cell(int dim, const double *odom_meas, const double *odom_info, double s_factor,
                        const double *poses, const int *ids, const double *meas, const double *info,
                        int i, int j, int fast_iter, int slow_iter, int *solved, oracle_stats_t *stats)
{
    int ms = meas_size(dim), is = info_size(dim);
    int loi = imin(ids[2 * i], ids[2 * i + 1]), hii = imax(ids[2 * i], ids[2 * i + 1]);
    if (i == j) {
        *solved = 1;
        return oracle_solve_cell(dim, odom_meas, odom_info, s_factor, poses, loi, hii, 1, ids + 2 * i,
                                 meas + (size_t)i * ms, info + (size_t)i * is, fast_iter, NULL, NULL, stats);
    }
    int loj = imin(ids[2 * j], ids[2 * j + 1]), hij = imax(ids[2 * j], ids[2 * j + 1]);
    if (imin(hii, hij) - imax(loi, loj) <= 0) { *solved = 0; return 0; }
    *solved = 1;
    int lid[4] = { ids[2 * i], ids[2 * i + 1], ids[2 * j], ids[2 * j + 1] };
    double lm[14], li[42];
    memcpy(lm, meas + (size_t)i * ms, sizeof(double) * ms);
    memcpy(lm + ms, meas + (size_t)j * ms, sizeof(double) * ms);
    memcpy(li, info + (size_t)i * is, sizeof(double) * is);
    memcpy(li + is, info + (size_t)j * is, sizeof(double) * is);
    return oracle_solve_cell(dim, odom_meas, odom_info, s_factor, poses, imin(loi, loj), imax(hii, hij), 2,
                             lid, lm, li, slow_iter, NULL, NULL, stats);
}

/* Batch of pair cells over POSIX threads that draw cells from one shared queue (an atomic counter over the
 * list, in list order: callers put the expensive cells first): the all-core CPU baseline SURVEY.md 8(d) asks
 * for (the reference itself is single-threaded; its only concurrency is independent processes,
 * bash/ipc_experiments_2D.sh:34).  A static partition left the run waiting for whichever thread drew a cell
 * that runs to the iteration cap (round 3: 10x scaling on 256 threads).
 * max_chi2_out / iterations_out have n entries.  Returns the number of threads actually started. */
#include <pthread.h>
typedef struct {
    int dim; const double *odom_meas, *odom_info; double s_factor; const double *poses; const int *ids;
    const double *meas, *info; int fast_iter, slow_iter; int n; const int *ci, *cj; double *mx; int *its;
    int t, T; int *next;
} mt_job_t;
static void *mt_worker(void *arg)
{
    mt_job_t *q = (mt_job_t *)arg;
    if (q->T == 0) return NULL;
    for (;;) {
        const int k = __atomic_fetch_add(q->next, 1, __ATOMIC_RELAXED);
        if (k >= q->n) break;
        int solved;
        oracle_stats_t st;
        memset(&st, 0, sizeof st);
        double c = oracle_pair_cell(q->dim, q->odom_meas, q->odom_info, q->s_factor, q->poses, q->ids, q->meas,
                                    q->info, q->ci[k], q->cj[k], q->fast_iter, q->slow_iter, &solved, &st);
        q->mx[k] = solved ? c : NAN;
        if (q->its) q->its[k] = st.iterations;
    }
    return NULL;
}
int oracle_pair_cells_mt(int dim, const double *odom_meas, const double *odom_info, double s_factor,
                         const double *poses, const int *ids, const double *meas, const double *info,
                         int fast_iter, int slow_iter, int n, const int *ci, const int *cj, int nthreads,
                         double *max_chi2_out, int *iterations_out)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 1024) nthreads = 1024;
    mt_job_t *jobs = (mt_job_t *)malloc(sizeof(mt_job_t) * (size_t)nthreads);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    int started = 0, next = 0;
    for (int t = 0; t < nthreads; ++t) {
        mt_job_t j = { dim, odom_meas, odom_info, s_factor, poses, ids, meas, info, fast_iter, slow_iter, n, ci, cj,
                       max_chi2_out, iterations_out, t, nthreads, &next };
        jobs[t] = j;
    }
    for (int t = 1; t < nthreads; ++t)
        if (pthread_create(&th[t], NULL, mt_worker, &jobs[t]) == 0) ++started;
        else { jobs[t].T = 0; }
    /* a thread that could not be created: its share is done here after ours */
    mt_worker(&jobs[0]);
    for (int t = 1; t < nthreads; ++t) {
        if (jobs[t].T) pthread_join(th[t], NULL);
        else { jobs[t].T = nthreads; mt_worker(&jobs[t]); }
    }
    free(jobs); free(th);
    return started + 1;
}

/* Full matrix over candidates [0,N): maxchi2 (N*N doubles, symmetric, NaN where the cell is
 * not solved), okmat (N*N bytes).  rows_begin/rows_end/row_stride restrict the owner rows
 * (cells (i,j), i<=j, with i in the row set) -- used for shard tests; other cells untouched. */
void oracle_consistency_matrix(int dim, int V, const double *odom_meas, const double *odom_info,
                               double s_factor, int N, const int *ids, const double *meas,
                               const double *info, double fast_th, int fast_iter, double slow_th,
                               int slow_iter, double *maxchi2, unsigned char *okmat)
{
    int ps = pose_size(dim);
    double *poses = (double *)malloc(sizeof(double) * (size_t)V * ps);
    oracle_propagate(dim, V, odom_meas, poses);
    int solved;
    for (int i = 0; i < N; ++i) {
        double c = oracle_pair_cell(dim, odom_meas, odom_info, s_factor, poses, ids, meas, info, i, i,
                                    fast_iter, slow_iter, &solved, NULL);
        if (maxchi2) maxchi2[(size_t)i * N + i] = c;
        okmat[(size_t)i * N + i] = !(c > fast_th);               /* chi2 > th => reject */
    }
    for (int i = 0; i < N; ++i)
        for (int j = i + 1; j < N; ++j) {
            double c = oracle_pair_cell(dim, odom_meas, odom_info, s_factor, poses, ids, meas, info, i, j,
                                        fast_iter, slow_iter, &solved, NULL);
            unsigned char ok;
            if (solved) ok = !(c > slow_th);
            else { ok = okmat[(size_t)i * N + i] && okmat[(size_t)j * N + j]; c = NAN; }
            if (maxchi2) { maxchi2[(size_t)i * N + j] = c; maxchi2[(size_t)j * N + i] = c; }
            okmat[(size_t)i * N + j] = ok; okmat[(size_t)j * N + i] = ok;
        }
    free(poses);
}

/* Set-max P2: candidates in cmpTime order; k joins iff C[k][k] and C[k][j] for every
 * already accepted j. */
void oracle_set_max(int N, const unsigned char *okmat, const int *order, unsigned char *accepted)
{
    memset(accepted, 0, (size_t)N);
    for (int a = 0; a < N; ++a) {
        int k = order[a];
        if (!okmat[(size_t)k * N + k]) continue;
        int ok = 1;
        for (int j = 0; j < N && ok; ++j)
            if (accepted[j] && !okmat[(size_t)k * N + j]) ok = 0;
        if (ok) accepted[k] = 1;
    }
}

/* ------------------------------------------------------------------------------------ */
/* Faithful incremental IPC (reference src/consensus.cpp:9-33,43-75,124-171).             */
/* State: global pose estimates (mutated on accept), consensus set.                       */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    int dim, V, N;
    double *odom_meas, *odom_info;
    double s_factor, fast_th, slow_th;
    int fast_iter, slow_iter;
    int *ids; double *meas, *info;
    double *poses;
    int *cns; int ncns;             /* consensus set: candidate indices, in acceptance order */
} oracle_ipc_t;

oracle_ipc_t *oracle_ipc_create(int dim, int V, const double *odom_meas, const double *odom_info,
                                double s_factor, double fast_th, int fast_iter, double slow_th,
                                int slow_iter, int N, const int *ids, const double *meas,
                                const double *info)
{
    oracle_ipc_t *h = (oracle_ipc_t *)calloc(1, sizeof *h);
    int ms = meas_size(dim), is = info_size(dim), ps = pose_size(dim);
    h->dim = dim; h->V = V; h->N = N;
    h->s_factor = s_factor; h->fast_th = fast_th; h->slow_th = slow_th;
    h->fast_iter = fast_iter; h->slow_iter = slow_iter;
    size_t ne = (size_t)(V > 1 ? V - 1 : 1), nn = (size_t)(N > 0 ? N : 1);
    h->odom_meas = (double *)malloc(sizeof(double) * ne * ms);
    h->odom_info = (double *)malloc(sizeof(double) * ne * is);
    memcpy(h->odom_meas, odom_meas, sizeof(double) * (size_t)(V - 1) * ms);
    memcpy(h->odom_info, odom_info, sizeof(double) * (size_t)(V - 1) * is);
    h->ids = (int *)malloc(sizeof(int) * nn * 2);
    h->meas = (double *)malloc(sizeof(double) * nn * ms);
    h->info = (double *)malloc(sizeof(double) * nn * is);
    memcpy(h->ids, ids, sizeof(int) * (size_t)N * 2);
    memcpy(h->meas, meas, sizeof(double) * (size_t)N * ms);
    memcpy(h->info, info, sizeof(double) * (size_t)N * is);
    h->poses = (double *)malloc(sizeof(double) * (size_t)V * ps);
    oracle_propagate(dim, V, odom_meas, h->poses);               /* consensus.cpp:23 */
    h->cns = (int *)malloc(sizeof(int) * nn);
    h->ncns = 0;
    return h;
}
void oracle_ipc_destroy(oracle_ipc_t *h)
{
    if (!h) return;
    free(h->odom_meas); free(h->odom_info); free(h->ids); free(h->meas); free(h->info);
    free(h->poses); free(h->cns); free(h);
}
int oracle_ipc_consensus_size(const oracle_ipc_t *h) { return h->ncns; }
void oracle_ipc_consensus(const oracle_ipc_t *h, int *out) { memcpy(out, h->cns, sizeof(int) * (size_t)h->ncns); }
void oracle_ipc_poses(const oracle_ipc_t *h, double *out)
{ memcpy(out, h->poses, sizeof(double) * (size_t)h->V * pose_size(h->dim)); }

/* The whole state of the reference's IPC object (include/ipc/consensus.hpp:23-32: the vertex estimates of the borrowed
 * optimizer and _max_consensus_set) set from outside: a run continued from a saved state, e.g. one the GPU engine dumped
 * late in a run the oracle cannot reach on its own in days (tests/golden/make_late_state_golden.py). */
void oracle_ipc_set_state(oracle_ipc_t *h, const double *poses, const int *cns, int ncns)
{
    memcpy(h->poses, poses, sizeof(double) * (size_t)h->V * pose_size(h->dim));
    h->cns = (int *)realloc(h->cns, sizeof(int) * ((size_t)ncns + (size_t)h->N + 1));   /* (room for every later accept) */
    memcpy(h->cns, cns, sizeof(int) * (size_t)ncns);
    h->ncns = ncns;
}

/* agreementCheck (consensus.cpp:43-75) for candidate k; returns 1 accepted / 0 rejected.
 * info_out (may be NULL): [lo, hi, n_cluster_loops, iterations]; maxchi2_out may be NULL. */
int oracle_ipc_agreement_check(oracle_ipc_t *h, int k, int *info_out, double *maxchi2_out)
{
    int dim = h->dim, ms = meas_size(dim), is = info_size(dim), ps = pose_size(dim);
    int lo = imin(h->ids[2 * k], h->ids[2 * k + 1]), hi = imax(h->ids[2 * k], h->ids[2 * k + 1]);
    /* computeIndependentSubgraph, consensus.cpp:124-171 */
    unsigned char *inc = (unsigned char *)calloc((size_t)(h->ncns > 0 ? h->ncns : 1), 1);
    int *members = (int *)malloc(sizeof(int) * (size_t)(h->ncns + 1));
    int nm = 0, found = 1;
    while (found) {
        found = 0;
        for (int c = 0; c < h->ncns; ++c) {
            if (inc[c]) continue;
            int e = h->cns[c];
            int t0 = imin(h->ids[2 * e], h->ids[2 * e + 1]), t1 = imax(h->ids[2 * e], h->ids[2 * e + 1]);
            int inter = imin(t1, hi) - imax(t0, lo);
            if (inter <= 0) continue;
            lo = imin(lo, t0); hi = imax(hi, t1);
            inc[c] = 1; found = 1;
            /* eset_independent is a std::set of edge pointers (consensus.cpp:47, :165): an edge that sits in
             * _max_consensus_set twice (a re-checked candidate is pushed again on accept, :70) enters once */
            int dup = 0;
            for (int m = 0; m < nm; ++m) dup |= members[m] == e;
            if (!dup) members[nm++] = e;
        }
    }
    int intersection = nm > 0;
    double th = intersection ? h->slow_th : h->fast_th;           /* consensus.cpp:50-52 */
    int iter_base = intersection ? h->slow_iter : h->fast_iter;
    int ncluster = nm;
    {
        int dup = 0;                                               /* consensus.cpp:56: insert into the same std::set */
        for (int m = 0; m < nm; ++m) dup |= members[m] == k;
        if (!dup) members[nm++] = k;
    }
    int *lid = (int *)malloc(sizeof(int) * 2 * (size_t)nm);
    double *lm = (double *)malloc(sizeof(double) * (size_t)nm * ms);
    double *li = (double *)malloc(sizeof(double) * (size_t)nm * is);
    for (int m = 0; m < nm; ++m) {
        int e = members[m];
        lid[2 * m] = h->ids[2 * e]; lid[2 * m + 1] = h->ids[2 * e + 1];
        memcpy(lm + (size_t)m * ms, h->meas + (size_t)e * ms, sizeof(double) * ms);
        memcpy(li + (size_t)m * is, h->info + (size_t)e * is, sizeof(double) * is);
    }
    double *newposes = (double *)malloc(sizeof(double) * (size_t)(hi - lo + 1) * ps);
    oracle_stats_t st;
    /* store / fixComplementary / isAgreeingWithCurrentState, consensus.cpp:59-62 */
    double mx = oracle_solve_cell(dim, h->odom_meas, h->odom_info, h->s_factor, h->poses, lo, hi, nm,
                                  lid, lm, li, iter_base, NULL, newposes, &st);
    int agree = !(mx > th);
    if (agree) {                                                   /* consensus.cpp:69-71 */
        memcpy(h->poses + (size_t)lo * ps, newposes, sizeof(double) * (size_t)(hi - lo + 1) * ps);
        h->cns[h->ncns++] = k;
        double Z[12];
        for (int i = hi + 1; i < h->V; ++i) {                      /* propagateCurrentGuess */
            oracle_meas_to_pose(dim, h->odom_meas + (size_t)(i - 1) * ms, Z);
            oracle_pose_mul(dim, h->poses + (size_t)(i - 1) * ps, Z, h->poses + (size_t)i * ps);
        }
    }                                                              /* else restore: poses untouched */
    if (info_out) { info_out[0] = lo; info_out[1] = hi; info_out[2] = ncluster; info_out[3] = st.iterations; }
    if (maxchi2_out) *maxchi2_out = mx;
    free(inc); free(members); free(lid); free(lm); free(li); free(newposes);
    return agree;
}

/* sizes for the ctypes wrapper */
int oracle_stats_size(void) { return (int)sizeof(oracle_stats_t); }
