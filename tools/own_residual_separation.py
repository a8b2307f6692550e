#!/usr/bin/env python
"""How well does a candidate's OWN chi2 at the state it is checked from separate the accepted from the rejected candidates?
(the prediction the speculative pipeline schedules by: engine.hip, k_cand_own_chi2).  One-at-a-time faithful run, the
current poses read back before every check.  usage: python tools/own_residual_separation.py [C2 C1 ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("IPC_SPEC_WINDOW", "1")
import numpy as np
import bench
from ipc_amd.consensus import IPC


def own_chi2_se2(poses, ids, meas, info):
    a, b = poses[ids[0]], poses[ids[1]]
    ca, sa = np.cos(a[2]), np.sin(a[2])
    d = b[:2] - a[:2]
    rel = np.array([ca * d[0] + sa * d[1], -sa * d[0] + ca * d[1]]) - meas[:2]
    cz, sz = np.cos(meas[2]), np.sin(meas[2])
    e = np.array([cz * rel[0] + sz * rel[1], -sz * rel[0] + cz * rel[1], b[2] - a[2] - meas[2]])
    e[2] -= 2 * np.pi * np.rint(e[2] / (2 * np.pi))
    om = np.array([[info[0], info[1], info[2]], [info[1], info[3], info[4]], [info[2], info[4], info[5]]])
    return float(e @ om @ e)


def own_chi2_se3(poses, ids, meas, info):
    """EdgeSE3::computeError: toVectorMQT(Z^-1 (Xi^-1 Xj)) = (translation, x y z of the unit quaternion with w >= 0)."""
    Ra, ta = poses[ids[0]][:9].reshape(3, 3), poses[ids[0]][9:]
    Rb, tb = poses[ids[1]][:9].reshape(3, 3), poses[ids[1]][9:]
    q = meas[3:7] / np.linalg.norm(meas[3:7])
    x, y, z, w = q
    Rz = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                   [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                   [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    Rd = Rz.T @ (Ra.T @ Rb)
    td = Rz.T @ (Ra.T @ (tb - ta) - meas[:3])
    qw = 0.5 * np.sqrt(max(1e-300, 1.0 + np.trace(Rd)))                # (fine away from rotations by pi: these are residuals)
    qv = np.array([Rd[2, 1] - Rd[1, 2], Rd[0, 2] - Rd[2, 0], Rd[1, 0] - Rd[0, 1]]) / (4.0 * qw)
    e = np.concatenate([td, qv / np.sqrt(qw * qw + qv @ qv)])
    om = np.zeros((6, 6))
    om[np.triu_indices(6)] = info
    om = om + om.T - np.diag(np.diag(om))
    return float(e @ om @ e)


for wl in sys.argv[1:] or ["C2", "C1"]:
    g, cfg, _ = bench.build_workload(wl)
    eng = IPC(g, cfg)
    order = eng.candidate_order()
    eng.reset()
    rows = []
    for k in order:
        own = (own_chi2_se2 if g.dim == 2 else own_chi2_se3)(eng.current_poses(), g.loop_ids[k], g.loop_meas[k], g.loop_info[k])
        ok, info = eng.agreementCheck(int(k), with_info=True)
        rows.append((ok, own, info.iterations))
    r = np.array(rows, dtype=float)
    acc = r[:, 0] == 1
    print(wl, "accepted %d: own chi2 max %.3g p99 %.3g median %.3g | rejected %d: own chi2 min %.3g p1 %.3g median %.3g" % (
        acc.sum(), r[acc, 1].max(), np.percentile(r[acc, 1], 99), np.median(r[acc, 1]),
        (~acc).sum(), r[~acc, 1].min(), np.percentile(r[~acc, 1], 1), np.median(r[~acc, 1])))
    print("   (the engine's cut: %.0f = 10 x the slow threshold)" % (10 * cfg.slow_reject_th))
    for T in (30, 100, 300, 1e3, 3e3, 1e4):
        print("   threshold %6.0f: accepted above %3d | rejected below %4d (%.1f %% of the reject iterations)" % (
            T, int((r[acc, 1] > T).sum()), int((r[~acc, 1] <= T).sum()), 100 * r[~acc & (r[:, 1] <= T), 2].sum() / r[~acc, 2].sum()))
    eng.close()
