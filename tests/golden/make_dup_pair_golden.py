#!/usr/bin/env python
"""Fixture with several candidates on ONE vertex pair (the reference's injector draws pairs at random,
scripts/generateDataset.py:188-246, so real inputs have them; IPC::agreementCheck checks the edge OBJECT it is given,
src/consensus.cpp:43-56).  Built from the committed small_se2_spoiled_n6_seed3.g2o by appending

  A  a second edge on the pair of an ACCEPTED true loop, carrying an outlier's measurement        -> must be rejected
  B  a second edge on the pair of a REJECTED outlier, carrying the relative pose of the final map  -> must be accepted
  C  a true loop once more the other way round (to -> from, inverse measurement)                  -> must be accepted

and the CPU oracle's decisions of the faithful run in (last vertex, file index) order.  Needs nothing of the reference.
"""
import math
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import numpy as np

from ipc_amd import graphio
from oracle import oracle as O

PRM = (10.0, 6.251, 50, 11.345, 100)


def rel_pose(pa, pb):
    c, s = math.cos(pa[2]), math.sin(pa[2])
    dx, dy = pb[0] - pa[0], pb[1] - pa[1]
    th = pb[2] - pa[2]
    return [c * dx + s * dy, -s * dx + c * dy, math.atan2(math.sin(th), math.cos(th))]


def inverse(m):
    c, s = math.cos(m[2]), math.sin(m[2])
    return [-(c * m[0] + s * m[1]), -(-s * m[0] + c * m[1]), -m[2]]


def run(g):
    inc = O.IncrementalIPC(g.dim, g.odom_meas, g.odom_info, PRM[0], PRM[1], PRM[2], PRM[3], PRM[4], g.loop_ids,
                           g.loop_meas, g.loop_info)
    order = O.candidate_order(g.loop_ids)
    dec = np.array([inc.agreement_check(int(k))[0] for k in order], dtype=np.uint8)
    return inc, order, dec


def rel_pose3(pa, pb):
    """poses as the oracle keeps them ([R row-major (9), t (3)]) -> x y z qx qy qz qw of Xa^-1 Xb"""
    from scipy.spatial.transform import Rotation
    Ra, ta, Rb, tb = pa[:9].reshape(3, 3), pa[9:], pb[:9].reshape(3, 3), pb[9:]
    R, t = Ra.T @ Rb, Ra.T @ (tb - ta)
    return list(t) + list(Rotation.from_matrix(R).as_quat())        # scipy: x y z w


def inverse3(m):
    from scipy.spatial.transform import Rotation
    R = Rotation.from_quat(np.asarray(m[3:7]) / np.linalg.norm(m[3:7])).as_matrix()
    return list(-(R.T @ np.asarray(m[:3]))) + list(Rotation.from_matrix(R.T).as_quat())


def main3():
    """The same three duplicates on the SE3 fixture (thresholds of bash/ipc_experiments_3D.sh: s = 50, 6.251 / 6.251)."""
    global PRM
    PRM = (50.0, 6.251, 50, 6.251, 100)
    g = graphio.read_g2o(os.path.join(HERE, "small_se3_spoiled_n5_seed4.g2o"))
    inc, order, dec = run(g)
    accepted = {int(k) for k, d in zip(order, dec) if d}
    true_loop, outlier, other = 1, 7, 2
    assert true_loop in accepted and other in accepted and outlier not in accepted
    P = inc.poses()
    a, b = g.loop_ids[outlier]
    ids = np.vstack([g.loop_ids, g.loop_ids[true_loop], g.loop_ids[outlier], g.loop_ids[other][::-1]]).astype(np.int32)
    meas = np.vstack([g.loop_meas, g.loop_meas[outlier], rel_pose3(P[a], P[b]), inverse3(g.loop_meas[other])])
    info = np.vstack([g.loop_info, g.loop_info[true_loop], g.loop_info[outlier], g.loop_info[other]])
    g2 = graphio.PoseGraph(g.dim, g.vertices, g.odom_meas, g.odom_info, ids, meas, info)
    graphio.write_g2o(os.path.join(HERE, "small_se3_dup_pairs.g2o"), g2)
    g2 = graphio.read_g2o(os.path.join(HERE, "small_se3_dup_pairs.g2o"))
    _, order2, dec2 = run(g2)
    by_index = np.zeros(g2.N, dtype=np.uint8)
    by_index[order2] = dec2
    n = g.N
    print("SE3 decisions by file index:", by_index, " A/B/C:", by_index[n], by_index[n + 1], by_index[n + 2])
    assert by_index[true_loop] == 1 and by_index[n] == 0
    assert by_index[outlier] == 0 and by_index[n + 1] == 1
    np.savez_compressed(os.path.join(HERE, "small_se3_dup_pairs_expected.npz"), order=order2, decision=dec2,
                        params=np.array(PRM), duplicates=np.array([[true_loop, n], [outlier, n + 1], [other, n + 2]]))


def main():
    g = graphio.read_g2o(os.path.join(HERE, "small_se2_spoiled_n6_seed3.g2o"))
    inc, order, dec = run(g)
    accepted = {int(k) for k, d in zip(order, dec) if d}
    true_loop, outlier, other = 2, 8, 3
    assert true_loop in accepted and other in accepted and outlier not in accepted
    P = inc.poses()
    a, b = g.loop_ids[outlier]
    ids = np.vstack([g.loop_ids, g.loop_ids[true_loop], g.loop_ids[outlier], g.loop_ids[other][::-1]]).astype(np.int32)
    meas = np.vstack([g.loop_meas, g.loop_meas[outlier], rel_pose(P[a], P[b]), inverse(g.loop_meas[other])])
    info = np.vstack([g.loop_info, g.loop_info[true_loop], g.loop_info[outlier], g.loop_info[other]])
    g2 = graphio.PoseGraph(g.dim, g.vertices, g.odom_meas, g.odom_info, ids, meas, info)
    graphio.write_g2o(os.path.join(HERE, "small_se2_dup_pairs.g2o"), g2)
    g2 = graphio.read_g2o(os.path.join(HERE, "small_se2_dup_pairs.g2o"))
    _, order2, dec2 = run(g2)
    by_index = np.zeros(g2.N, dtype=np.uint8)
    by_index[order2] = dec2
    n = g.N
    print("decisions by file index:", by_index, " A/B/C:", by_index[n], by_index[n + 1], by_index[n + 2])
    assert by_index[true_loop] == 1 and by_index[n] == 0            # same pair, different verdicts
    assert by_index[outlier] == 0 and by_index[n + 1] == 1
    assert by_index[n + 2] == 1
    np.savez_compressed(os.path.join(HERE, "small_se2_dup_pairs_expected.npz"), order=order2, decision=dec2,
                        params=np.array(PRM), duplicates=np.array([[true_loop, n], [outlier, n + 1], [other, n + 2]]))


if __name__ == "__main__":
    main()
    main3()
