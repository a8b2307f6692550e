"""A second, independent restatement of the g2o edge / vertex conventions the oracle rests on, written
against scipy's Rotation class instead of the oracle's hand-rolled 3x3 / quaternion code, and an
independent solver (scipy.optimize.least_squares on whitened residuals, parameters applied through the
independent oplus) for whole cells.  The reference's arithmetic lives in g2o 20201223 (absent here, SURVEY
8c), so this does not pin the oracle to g2o -- it removes the oracle's own algebra as a single point of
failure: a sign / ordering / frame slip in oracle/ipc_oracle.c shows up as a disagreement here.

Conventions restated (g2o upstream paths):
  EdgeSE2::computeError   e = (Z^-1 * (Xi^-1 * Xj)).toVector(), theta normalised          (types/slam2d/edge_se2.h)
  VertexSE2::oplusImpl    t += d_t (world frame), theta = normalize(theta + d_theta)       (types/slam2d/vertex_se2.h)
  EdgeSE3::computeError   e = toVectorMQT(Z^-1 * Xi^-1 * Xj) = (t, q_xyz) with q_w >= 0    (types/slam3d/edge_se3.cpp)
  VertexSE3::oplusImpl    X <- X * fromVectorMQT(d), q_w = sqrt(1 - |q_xyz|^2)             (types/slam3d/vertex_se3.h)
"""
import math
import os

import numpy as np
import pytest
from scipy.optimize import least_squares
from scipy.spatial.transform import Rotation

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---------------------------------------------------------------------------------------------
# independent pose algebra: SE2 as (x, y, theta), SE3 as (scipy Rotation, t)
# ---------------------------------------------------------------------------------------------
def wrap(a):
    return math.atan2(math.sin(a), math.cos(a))


def se2_err(z, xi, xj):
    """z, xi, xj = (x, y, theta).  Homogeneous 3x3 matrices, so nothing is shared with the oracle's formulas."""
    def T(p):
        c, s = math.cos(p[2]), math.sin(p[2])
        return np.array([[c, -s, p[0]], [s, c, p[1]], [0, 0, 1.0]])
    E = np.linalg.inv(T(z)) @ np.linalg.inv(T(xi)) @ T(xj)
    return np.array([E[0, 2], E[1, 2], math.atan2(E[1, 0], E[0, 0])])


def se2_oplus(x, d):
    return np.array([x[0] + d[0], x[1] + d[1], wrap(x[2] + d[2])])


class P3:
    def __init__(self, R, t):
        self.R, self.t = R, np.asarray(t, dtype=np.float64)

    @staticmethod
    def from_meas(m):                      # x y z qx qy qz qw (quaternion re-normalised, as EdgeSE3::read does)
        q = np.asarray(m[3:7], dtype=np.float64)
        return P3(Rotation.from_quat(q / np.linalg.norm(q)), m[:3])

    def __mul__(self, o):
        return P3(self.R * o.R, self.t + self.R.apply(o.t))

    def inv(self):
        Ri = self.R.inv()
        return P3(Ri, -Ri.apply(self.t))

    def flat12(self):                      # the oracle's storage: R row-major, then t
        return np.concatenate([self.R.as_matrix().reshape(9), self.t])


def se3_err(Z, Xi, Xj):
    E = Z.inv() * (Xi.inv() * Xj)
    q = E.R.as_quat()                      # x y z w
    if q[3] < 0:
        q = -q
    return np.concatenate([E.t, q[:3]])


def se3_oplus(X, d):
    n2 = float(d[3] ** 2 + d[4] ** 2 + d[5] ** 2)
    if n2 > 1.0:
        dR = Rotation.identity()
    else:
        dR = Rotation.from_quat([d[3], d[4], d[5], math.sqrt(1.0 - n2)])
    return X * P3(dR, d[:3])


def rand_meas3(rng, scale=1.0):
    rv = rng.normal(0, 0.8, 3)
    return np.concatenate([rng.normal(0, scale, 3), Rotation.from_rotvec(rv).as_quat()])


# ---------------------------------------------------------------------------------------------
# edge errors / oplus against the oracle
# ---------------------------------------------------------------------------------------------
def test_se2_error_and_oplus_against_homogeneous_matrices(oracle):
    O = oracle
    rng = np.random.default_rng(11)
    for _ in range(200):
        z = np.array([rng.normal(0, 2), rng.normal(0, 2), rng.uniform(-math.pi, math.pi)])
        xi = np.array([rng.normal(0, 5), rng.normal(0, 5), rng.uniform(-math.pi, math.pi)])
        xj = np.array([rng.normal(0, 5), rng.normal(0, 5), rng.uniform(-math.pi, math.pi)])
        e = O.edge_error(2, z, xi, xj)
        ei = se2_err(z, xi, xj)
        assert np.allclose(e[:2], ei[:2], atol=1e-12)
        assert abs(wrap(e[2] - ei[2])) < 1e-12
        d = rng.normal(0, 0.5, 3)
        y = O.pose_oplus(2, xi, d)
        yi = se2_oplus(xi, d)
        assert np.allclose(y[:2], yi[:2], atol=1e-13) and abs(wrap(y[2] - yi[2])) < 1e-12


def test_se3_error_and_oplus_against_scipy_rotations(oracle):
    O = oracle
    rng = np.random.default_rng(12)
    for trial in range(200):
        mz, mi, mj = rand_meas3(rng), rand_meas3(rng, 5.0), rand_meas3(rng, 5.0)
        Z, Xi, Xj = P3.from_meas(mz), P3.from_meas(mi), P3.from_meas(mj)
        oZ, oXi, oXj = O.meas_to_pose(3, mz), O.meas_to_pose(3, mi), O.meas_to_pose(3, mj)
        assert np.allclose(oXi, Xi.flat12(), atol=1e-14)           # storage convention: R row-major, t
        e = O.edge_error(3, oZ, oXi, oXj)
        assert np.allclose(e, se3_err(Z, Xi, Xj), atol=1e-12), trial
        d = np.concatenate([rng.normal(0, 0.5, 3), rng.normal(0, 0.2, 3)])
        assert np.allclose(O.pose_oplus(3, oXi, d), se3_oplus(Xi, d).flat12(), atol=1e-13)
        # composition / inverse
        assert np.allclose(O.pose_mul(3, oXi, oXj), (Xi * Xj).flat12(), atol=1e-12)
        assert np.allclose(O.pose_inv(3, oXi), Xi.inv().flat12(), atol=1e-13)


# ---------------------------------------------------------------------------------------------
# whole cells: independent residual function + scipy's trust-region least squares
# ---------------------------------------------------------------------------------------------
def _sym(up, d):
    M = np.zeros((d, d))
    k = 0
    for i in range(d):
        for j in range(i, d):
            M[i, j] = M[j, i] = up[k]
            k += 1
    return M


def _independent_problem(g, cfg_s, lo, hi, loops, x0):
    """sum_e e^T Omega e over the poses lo+1..hi (pose lo fixed), parametrised by increments v around the poses x0
    through the independent oplus; returns (resid(v) -> whitened residuals [n_edges * d], n_edges, d, poses_of(v))."""
    dim = g.dim
    d = 3 if dim == 2 else 6
    L = hi - lo
    if dim == 2:
        base = [np.array(p) for p in x0]
        mk = lambda m: np.asarray(m, dtype=np.float64)
        err, oplus = se2_err, se2_oplus
    else:
        base = [P3(Rotation.from_matrix(p[:9].reshape(3, 3)), p[9:]) for p in x0]
        mk = P3.from_meas
        err, oplus = se3_err, se3_oplus
    edges = []
    for k in range(lo, hi):
        edges.append((k - lo, k + 1 - lo, mk(g.odom_meas[k]), _sym(g.odom_info[k], d) * cfg_s))
    for l in loops:
        a, b = g.loop_ids[l]
        edges.append((a - lo, b - lo, mk(g.loop_meas[l]), _sym(g.loop_info[l], d)))
    chol = [np.linalg.cholesky(om).T for (_, _, _, om) in edges]          # e^T Om e = |U e|^2

    def poses_of(v):
        return [base[0]] + [oplus(base[j], v[(j - 1) * d:j * d]) for j in range(1, L + 1)]

    def resid(v):
        X = poses_of(v)
        return np.concatenate([U @ err(z, X[a], X[b]) for (a, b, z, _), U in zip(edges, chol)])

    return resid, len(edges), d, L


@pytest.mark.parametrize("fixture,dim", [("small_se2_spoiled_n6_seed3.g2o", 2), ("small_se3_spoiled_n5_seed4.g2o", 3)])
def test_converged_cells_are_stationary_points_of_an_independent_objective(oracle, fixture, dim):
    """Diagonal and pair cells of the golden fixtures.  At the poses the oracle's dog-leg (g2o's algorithm
    restated) ends on, the independently written residual function gives the same per-edge chi2, and its gradient
    (central differences through the independent oplus) vanishes: the oracle converged to a minimum of the problem
    as restated here, not of a mis-assembled one.  Where the minimum is unambiguous (consistent closures, small
    chi2) scipy's trust-region solver started from the open-loop poses reaches the same chi2 as well."""
    from ipc_amd import graphio
    O = oracle
    g = graphio.read_g2o(os.path.join(GOLD, fixture))
    s = 10.0 if dim == 2 else 50.0
    poses = O.propagate(dim, g.odom_meas)
    lo_all, hi_all = g.loop_ids.min(1), g.loop_ids.max(1)
    cells = [(k, k) for k in range(g.N)]
    for i in range(g.N):
        for j in range(i + 1, g.N):
            if min(hi_all[i], hi_all[j]) - max(lo_all[i], lo_all[j]) > 0:
                cells.append((i, j))
    checked = minima = 0
    for (i, j) in cells[:24]:
        loops = [i] if i == j else [i, j]
        lo, hi = int(min(lo_all[loops])), int(max(hi_all[loops]))
        if hi - lo > 45:
            continue
        r = O.solve_cell(dim, g.odom_meas, g.odom_info, s, poses, lo, hi, g.loop_ids[loops], g.loop_meas[loops],
                         g.loop_info[loops], 500, want_poses=True)
        if not r["terminated"]:
            continue
        resid, ne, d, L = _independent_problem(g, s, lo, hi, loops, r["poses"])
        r0 = resid(np.zeros(L * d))
        chi = (r0.reshape(ne, d) ** 2).sum(1)
        assert np.allclose(chi, r["chi2"], rtol=1e-8, atol=1e-10), (i, j)
        h = 1e-6
        grad = np.zeros(L * d)
        for k in range(L * d):
            v = np.zeros(L * d)
            v[k] = h
            fp = (resid(v) ** 2).sum()
            v[k] = -h
            fm = (resid(v) ** 2).sum()
            grad[k] = (fp - fm) / (2 * h)
        # scale: the gradient of one edge term alone is ~ 2 |U^T U e| ~ 2 sqrt(chi2_e) * sqrt(|Omega|)
        scale = 2.0 * math.sqrt(max(chi.max(), 1e-12)) * 40.0
        assert np.abs(grad).max() <= 1e-4 * scale + 1e-6, (i, j, float(np.abs(grad).max()), scale)
        checked += 1
        if r["max_chi2"] < 30.0:
            res0, _, _, _ = _independent_problem(g, s, lo, hi, loops, poses[lo:hi + 1])
            sol = least_squares(res0, np.zeros(L * d), method="trf", xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=600)
            assert 2.0 * sol.cost == pytest.approx(chi.sum(), rel=1e-6, abs=1e-8), (i, j)
            minima += 1
    assert checked >= 6 and minima >= 1
