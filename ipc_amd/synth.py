"""Synthetic stand-ins for the benchmark datasets and the outlier injector.

The datasets the reference's configs name (INTEL, MIT, sphere2500, ...) are not shipped with
the reference and there is no network, so every BASELINE config runs on a synthetic graph of
the canonical size:

  C1/C2  intel_like()   SE2, V=1228, 256 true loops   (cfg/2D/INTEL_params.yaml:6)
  C3     mit_like()     SE2, V=808,   20 true loops   (cfg/2D/MIT_params.yaml:6)
  C4     sphere_like()  SE3, V=2500, 2450 true loops  (cfg/3D/SPHERE_params.yaml:6)
  C5     chain3d()      SE3, V=50000, 5000 true loops with bounded span

inject_outliers() restates the sampling distribution of the reference's injector
(scripts/generateDataset.py:188-246, the Vertigo script): same RNG (Python's Mersenne twister),
same draw order, so that for a given seed it produces the same false loop closures as the
reference script does on the same clean file (pinned by tests/golden/, generated with the
reference script itself).  The 3-D quirk is preserved: the script writes the quaternion it
builds as "w x y z" into slots g2o reads as "qx qy qz qw" (generateDataset.py:101,225,239).
"""
import math
import random

import numpy as np

from .graphio import PoseGraph, info_size


# ----------------------------------------------------------------------------------------
# small pose helpers (data generation only; the engine's arithmetic lives in csrc/)
# ----------------------------------------------------------------------------------------
def _wrap(a):
    return (a + np.pi) % (2 * np.pi) - np.pi


def _se2_between(a, b):
    c, s = math.cos(a[2]), math.sin(a[2])
    dx, dy = b[0] - a[0], b[1] - a[1]
    return np.array([c * dx + s * dy, -s * dx + c * dy, _wrap(b[2] - a[2])])


def _quat_to_R(q):  # q = (x, y, z, w)
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _R_to_quat(R):  # returns (x, y, z, w), w >= 0
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = math.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[3] = (R[k, j] - R[j, k]) / s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
    q /= np.linalg.norm(q)
    return q if q[3] >= 0 else -q


def _rotvec_to_R(v):
    th = np.linalg.norm(v)
    if th < 1e-12:
        return np.eye(3)
    k = v / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * K @ K


def _upper(M):
    d = M.shape[0]
    return np.array([M[i, j] for i in range(d) for j in range(i, d)])


def _se3_rel_meas(Ra, ta, Rb, tb, rng, sig_t, sig_r):
    R = Ra.T @ Rb
    t = Ra.T @ (tb - ta)
    R = R @ _rotvec_to_R(rng.normal(0, sig_r, 3))
    t = t + rng.normal(0, sig_t, 3)
    return np.concatenate([t, _R_to_quat(R)])


# ----------------------------------------------------------------------------------------
# SE2 graphs
# ----------------------------------------------------------------------------------------
def _se2_info(rng, sig_xy, sig_th, skew=True):
    """Full (non-diagonal) 3x3 information, as public INTEL/MIT files carry."""
    sx, sy = sig_xy * rng.uniform(0.8, 1.25), sig_xy * rng.uniform(0.8, 1.25)
    phi = rng.uniform(-0.3, 0.3) if skew else 0.0
    c, s = math.cos(phi), math.sin(phi)
    Rm = np.array([[c, -s], [s, c]])
    cov = np.zeros((3, 3))
    cov[:2, :2] = Rm @ np.diag([sx * sx, sy * sy]) @ Rm.T
    cov[2, 2] = sig_th ** 2
    if skew:
        r = rng.uniform(-0.2, 0.2)
        cov[0, 2] = cov[2, 0] = r * sx * sig_th
    return cov, np.linalg.inv(cov)


def _se2_graph(V, n_loops, seed, laps, sig_o=(0.03, 0.012), sig_l=(0.05, 0.02), radius=1.0,
               min_gap=20, name="se2", odom_trust=10.0):
    """odom_trust: the written odometry covariance over-states the drawn noise by this factor,
    i.e. the graph is statistically consistent once the engine scales the odometry
    information by s_factor = odom_trust (the reference's robustifyVoters)."""
    rng = np.random.default_rng(seed)
    # ground-truth trajectory: a slowly deforming Lissajous tour that re-visits places
    t = np.linspace(0.0, 2 * np.pi * laps, V)
    gx = 12.0 * np.sin(t * 1.0 + 0.3) + 3.0 * np.sin(t * 0.11)
    gy = 8.0 * np.sin(t * 2.0) + 2.0 * np.cos(t * 0.07)
    gx += np.cumsum(rng.normal(0, 0.01, V))
    gy += np.cumsum(rng.normal(0, 0.01, V))
    gth = np.arctan2(np.gradient(gy), np.gradient(gx))
    gt = np.stack([gx, gy, gth], axis=1)
    th0 = gt[0, 2]
    gt[:, :2] -= gt[0, :2]
    c0, s0 = math.cos(-th0), math.sin(-th0)
    xy = gt[:, :2].copy()
    gt[:, 0] = c0 * xy[:, 0] - s0 * xy[:, 1]
    gt[:, 1] = s0 * xy[:, 0] + c0 * xy[:, 1]
    gt[:, 2] = _wrap(gt[:, 2] - th0)

    odom_meas = np.zeros((V - 1, 3))
    odom_info = np.zeros((V - 1, 6))
    for j in range(V - 1):
        cov, inf = _se2_info(rng, sig_o[0], sig_o[1])
        z = _se2_between(gt[j], gt[j + 1]) + rng.multivariate_normal(np.zeros(3), cov / odom_trust)
        z[2] = _wrap(z[2])
        odom_meas[j] = z
        odom_info[j] = _upper(inf)

    # true loops: spatially close pose pairs, far apart in time
    d2 = ((gt[:, None, :2] - gt[None, :, :2]) ** 2).sum(-1)
    ii, jj = np.where(np.triu(d2 < radius * radius, k=min_gap))
    if len(ii) < n_loops:
        raise RuntimeError("trajectory has only %d re-visits, need %d" % (len(ii), n_loops))
    pick = rng.choice(len(ii), size=n_loops, replace=False)
    pick.sort()
    loop_ids = np.stack([ii[pick], jj[pick]], axis=1).astype(np.int32)
    order = np.lexsort((loop_ids[:, 0], loop_ids[:, 1]))     # file order: by closing time
    loop_ids = loop_ids[order]
    loop_meas = np.zeros((n_loops, 3))
    loop_info = np.zeros((n_loops, 6))
    for k, (a, b) in enumerate(loop_ids):
        cov, inf = _se2_info(rng, sig_l[0], sig_l[1])
        z = _se2_between(gt[a], gt[b]) + rng.multivariate_normal(np.zeros(3), cov)
        z[2] = _wrap(z[2])
        loop_meas[k] = z
        loop_info[k] = _upper(inf)

    # file vertex estimates: open-loop odometry (what odometryInitialization would give)
    verts = np.zeros((V, 3))
    for j in range(V - 1):
        a = verts[j]
        c, s = math.cos(a[2]), math.sin(a[2])
        z = odom_meas[j]
        verts[j + 1] = [a[0] + c * z[0] - s * z[1], a[1] + s * z[0] + c * z[1], _wrap(a[2] + z[2])]
    return PoseGraph(2, verts, odom_meas, odom_info, loop_ids, loop_meas, loop_info,
                     dict(name=name, seed=seed, canonic_inliers=n_loops, ground_truth=gt))


def ring_se2(seed=20261003, V=2000, per_ring=50, radius=8.0, sig_o=(0.03, 0.012), sig_l=(0.05, 0.02), odom_trust=10.0):
    """The SE2 counterpart of sphere_like(): a slowly widening spiral with `per_ring` poses per turn, true loops join pose i
    with the pose one turn later (i + per_ring) -- V - per_ring loops of equal span that chain into ONE cluster in the
    faithful mode (round 5: the banded large-cluster solver on 3 x 3 blocks)."""
    rng = np.random.default_rng(seed)
    k = np.arange(V)
    ang = 2 * np.pi * k / per_ring
    rad = radius * (1.0 + 0.15 * k / V)
    gx, gy = rad * np.cos(ang), rad * np.sin(ang)
    gth = _wrap(ang + np.pi / 2)
    gt = np.stack([gx, gy, gth], axis=1)
    th0 = gt[0, 2]
    gt[:, :2] -= gt[0, :2]
    c0, s0 = math.cos(-th0), math.sin(-th0)
    xy = gt[:, :2].copy()
    gt[:, 0] = c0 * xy[:, 0] - s0 * xy[:, 1]
    gt[:, 1] = s0 * xy[:, 0] + c0 * xy[:, 1]
    gt[:, 2] = _wrap(gt[:, 2] - th0)
    odom_meas = np.zeros((V - 1, 3))
    odom_info = np.zeros((V - 1, 6))
    for j in range(V - 1):
        cov, inf = _se2_info(rng, sig_o[0], sig_o[1])
        z = _se2_between(gt[j], gt[j + 1]) + rng.multivariate_normal(np.zeros(3), cov / odom_trust)
        z[2] = _wrap(z[2])
        odom_meas[j] = z
        odom_info[j] = _upper(inf)
    n_loops = V - per_ring
    loop_ids = np.stack([np.arange(n_loops), np.arange(n_loops) + per_ring], axis=1).astype(np.int32)
    loop_meas = np.zeros((n_loops, 3))
    loop_info = np.zeros((n_loops, 6))
    for q, (a, b) in enumerate(loop_ids):
        cov, inf = _se2_info(rng, sig_l[0], sig_l[1])
        z = _se2_between(gt[a], gt[b]) + rng.multivariate_normal(np.zeros(3), cov)
        z[2] = _wrap(z[2])
        loop_meas[q] = z
        loop_info[q] = _upper(inf)
    verts = np.zeros((V, 3))
    for j in range(V - 1):
        a = verts[j]
        c, s_ = math.cos(a[2]), math.sin(a[2])
        z = odom_meas[j]
        verts[j + 1] = [a[0] + c * z[0] - s_ * z[1], a[1] + s_ * z[0] + c * z[1], _wrap(a[2] + z[2])]
    return PoseGraph(2, verts, odom_meas, odom_info, loop_ids, loop_meas, loop_info,
                     dict(name="ring-se2", seed=seed, canonic_inliers=n_loops, ground_truth=gt))


def intel_like(seed=20260929, V=1228, n_loops=256):
    return _se2_graph(V, n_loops, seed, laps=6.0, name="INTEL-like")


def mit_like(seed=20260930, V=808, n_loops=20):
    return _se2_graph(V, n_loops, seed, laps=3.0, name="MIT-like")


def small_se2(seed=7, V=60, n_loops=8):
    return _se2_graph(V, n_loops, seed, laps=2.5, radius=2.5, min_gap=6, name="small-se2")


# ----------------------------------------------------------------------------------------
# SE3 graphs
# ----------------------------------------------------------------------------------------
def _se3_info(rng, sig_t, sig_r, full=True):
    """6x6 information in g2o's (x y z qx qy qz) order; the rotation block is expressed on
    the quaternion vector part (~ half the rotation vector)."""
    d = np.array([sig_t, sig_t, sig_t, 0.5 * sig_r, 0.5 * sig_r, 0.5 * sig_r]) * rng.uniform(0.8, 1.25, 6)
    cov = np.diag(d * d)
    if full:
        Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        B = np.eye(6)
        B[:3, :3] = _rotvec_to_R(rng.normal(0, 0.2, 3))
        B[3:, 3:] = _rotvec_to_R(rng.normal(0, 0.2, 3))
        cov = B @ cov @ B.T
        del Q
    return cov, np.linalg.inv(cov)


def _se3_graph(gtR, gtt, loop_pairs, seed, sig_o=(0.02, 0.01), sig_l=(0.04, 0.02), name="se3",
               odom_trust=50.0, loop_trust=3.0):
    """odom_trust / loop_trust: the written covariances over-state the drawn noise by these
    factors (odometry: consistent once scaled by s_factor = odom_trust; loops: conservative,
    so that true closures pass the reference's 3-dof chi2 threshold on 6-dof errors)."""
    rng = np.random.default_rng(seed)
    ot = math.sqrt(odom_trust)
    lt = math.sqrt(loop_trust)
    V = gtt.shape[0]
    odom_meas = np.zeros((V - 1, 7))
    odom_info = np.zeros((V - 1, 21))
    for j in range(V - 1):
        _, inf = _se3_info(rng, sig_o[0], sig_o[1])
        odom_meas[j] = _se3_rel_meas(gtR[j], gtt[j], gtR[j + 1], gtt[j + 1], rng, sig_o[0] / ot, sig_o[1] / ot)
        odom_info[j] = _upper(inf)
    n = len(loop_pairs)
    loop_ids = np.asarray(loop_pairs, dtype=np.int32).reshape(n, 2)
    loop_meas = np.zeros((n, 7))
    loop_info = np.zeros((n, 21))
    for k, (a, b) in enumerate(loop_ids):
        _, inf = _se3_info(rng, sig_l[0], sig_l[1])
        loop_meas[k] = _se3_rel_meas(gtR[a], gtt[a], gtR[b], gtt[b], rng, sig_l[0] / lt, sig_l[1] / lt)
        loop_info[k] = _upper(inf)
    verts = np.zeros((V, 7))
    R, t = np.eye(3), np.zeros(3)
    verts[0] = [0, 0, 0, 0, 0, 0, 1]
    for j in range(V - 1):
        z = odom_meas[j]
        t = t + R @ z[:3]
        R = R @ _quat_to_R(z[3:])
        verts[j + 1] = np.concatenate([t, _R_to_quat(R)])
    return PoseGraph(3, verts, odom_meas, odom_info, loop_ids, loop_meas, loop_info,
                     dict(name=name, seed=seed, canonic_inliers=n))


def _look_frames(pos):
    """Orientation: x axis along the direction of travel, z roughly 'up' (radial)."""
    V = pos.shape[0]
    Rs = np.zeros((V, 3, 3))
    for i in range(V):
        fwd = pos[min(i + 1, V - 1)] - pos[max(i - 1, 0)]
        fwd /= np.linalg.norm(fwd) + 1e-12
        up = pos[i] / (np.linalg.norm(pos[i]) + 1e-12)
        if abs(fwd @ up) > 0.95:
            up = np.array([0.0, 0.0, 1.0]) if abs(fwd[2]) < 0.9 else np.array([1.0, 0.0, 0.0])
        y = np.cross(up, fwd)
        y /= np.linalg.norm(y)
        z = np.cross(fwd, y)
        Rs[i] = np.stack([fwd, y, z], axis=1)
    return Rs


def sphere_like(seed=20261001, rings=50, per_ring=50, radius=50.0):
    """sphere2500 stand-in: a spiral of `rings` turns with `per_ring` poses each on a sphere;
    true loops join pose i with the pose one turn later (i + per_ring): V - per_ring loops
    (2450 for 50 x 50), like the public sphere2500 graph."""
    V = rings * per_ring
    u = (np.arange(V) + 0.5) / V
    lat = (u - 0.5) * np.pi * 0.96
    lon = 2 * np.pi * np.arange(V) / per_ring
    pos = radius * np.stack([np.cos(lat) * np.cos(lon), np.cos(lat) * np.sin(lon), np.sin(lat)], axis=1)
    Rs = _look_frames(pos)
    pairs = [(i, i + per_ring) for i in range(V - per_ring)]
    g = _se3_graph(Rs, pos, pairs, seed, name="sphere-like")
    return g


def chain3d(seed=20261002, V=50000, n_loops=5000, max_span=200):
    """C5 stand-in: a 3-D random-walk helix with loops of bounded span."""
    rng = np.random.default_rng(seed)
    s = np.arange(V) * 0.05
    pos = np.stack([20 * np.cos(s) + 0.002 * np.arange(V), 20 * np.sin(s), 0.01 * np.arange(V) % 7.0
                    + np.cumsum(rng.normal(0, 0.01, V))], axis=1)
    Rs = _look_frames(pos + np.array([0, 0, 100.0]))
    a = rng.integers(0, V - max_span - 1, size=n_loops)
    span = rng.integers(2, max_span + 1, size=n_loops)
    pairs = np.stack([a, a + span], axis=1)
    pairs = pairs[np.lexsort((pairs[:, 0], pairs[:, 1]))]
    return _se3_graph(Rs, pos, [tuple(p) for p in pairs], seed, name="chain3d")


def small_se3(seed=11, V=40, n_loops=6):
    rng = np.random.default_rng(seed)
    s = np.arange(V) * 0.45
    pos = np.stack([4 * np.cos(s), 4 * np.sin(s), 0.15 * s], axis=1)
    Rs = _look_frames(pos + np.array([0, 0, 30.0]))
    per = int(round(2 * np.pi / 0.45))
    pairs = [(i, i + per) for i in rng.choice(V - per, size=n_loops, replace=False)]
    pairs.sort(key=lambda p: (p[1], p[0]))
    return _se3_graph(Rs, pos, pairs, seed, name="small-se3")


# ----------------------------------------------------------------------------------------
# outlier injection (distribution of reference scripts/generateDataset.py:188-246)
# ----------------------------------------------------------------------------------------
def _euler_to_wxyz(yaw, pitch, roll):
    sy, cy = math.sin(yaw * 0.5), math.cos(yaw * 0.5)
    sp, cp = math.sin(pitch * 0.5), math.cos(pitch * 0.5)
    sr, cr = math.sin(roll * 0.5), math.cos(roll * 0.5)
    return (cr * cp * cy + sr * sp * sy, sr * cp * cy - cr * sp * sy,
            cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy)


def sample_outliers(dim, n_poses, n_outliers, seed, group_size=1, local=False, perfect=False):
    """Returns (ids [n*g, 2] int32, meas [n*g, 3 or 7]).  Draw order identical to the
    reference script for a given seed (random.seed(seed); per outlier: v1, v2 until distinct,
    then the measurement components)."""
    rnd = random.Random()
    rnd.seed(seed)
    ids, meas = [], []
    top = n_poses - 1 - group_size
    for _ in range(n_outliers):
        v1 = v2 = 0
        while v1 == v2:
            v1 = rnd.randint(0, top)
            v2 = rnd.randint(v1, min(top, v1 + 20)) if local else rnd.randint(0, top)
            if v1 > v2:
                v1, v2 = v2, v1
            if v2 == v1 + 1:
                v2 = v1 + 2
        if dim == 2:
            m = [rnd.gauss(0, 0.3), rnd.gauss(0, 0.3), rnd.gauss(0, 10 * math.pi / 180.0)]
        else:
            m = [rnd.gauss(0, 0.3), rnd.gauss(0, 0.3), rnd.gauss(0, 0.3)]
            sigma = 10.0 * math.pi / 180.0
            roll, pitch, yaw = rnd.gauss(0, sigma), rnd.gauss(0, sigma), rnd.gauss(0, sigma)
            # the script emits (w x y z) into the slots g2o reads as (qx qy qz qw)
            m += list(_euler_to_wxyz(yaw, pitch, roll))
        if perfect:
            m = [0.0, 0.0, 0.0] if dim == 2 else [0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0]
        for g in range(group_size):
            ids.append((v1 + g, v2 + g))
            meas.append(list(m))
    return (np.asarray(ids, dtype=np.int32).reshape(-1, 2),
            np.asarray(meas, dtype=np.float64).reshape(-1, 3 if dim == 2 else 7))


def inject_outliers(g: PoseGraph, n_outliers, seed, group_size=1, local=False, information=None):
    """Appends false loop closures after all original edges (generateDataset.py:167-250).
    information=None copies the first non-odometry edge's information (ibid. 176-182)."""
    ids, meas = sample_outliers(g.dim, g.V, n_outliers, seed, group_size, local)
    if information is None:
        if g.N == 0:
            raise ValueError("no loop edge to copy the information matrix from")
        info_row = g.loop_info[0]
    else:
        info_row = np.asarray(information, dtype=np.float64).reshape(info_size(g.dim))
    info = np.tile(info_row, (ids.shape[0], 1))
    meta = dict(g.meta)
    meta["outliers"] = int(ids.shape[0])
    meta["outlier_seed"] = seed
    return PoseGraph(g.dim, g.vertices, g.odom_meas, g.odom_info,
                     np.concatenate([g.loop_ids, ids]), np.concatenate([g.loop_meas, meas]),
                     np.concatenate([g.loop_info, info]), meta)
