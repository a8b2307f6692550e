#!/usr/bin/env python
"""Memory-system experiment for the SE3 cell kernels: a helix whose odometry records are all identical, so a
build that makes every slot read the SAME 64 records (-DIPC_DBG_SAMEREC: L1-resident) computes bit-identical
results to the normal build (records streamed from L2) -- the time difference is what the constant loads cost.

usage (GPU box): python tools/uniform_chain_experiment.py ipc_amd/libipc_amd.so ipc_amd/libipc_dbg_samerec.so
"""
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def helix(V=2500, n_true=245, n_out=200, seed=77):
    from ipc_amd import synth
    from ipc_amd.graphio import PoseGraph
    rng = np.random.default_rng(seed)
    Rz = synth._rotvec_to_R(np.array([0.0, 0.02, 0.1]))
    tz = np.array([1.0, 0.0, 0.02])
    z = np.concatenate([tz, synth._R_to_quat(Rz)])
    _, inf = synth._se3_info(rng, 0.02, 0.01)
    odom_meas = np.tile(z, (V - 1, 1))
    odom_info = np.tile(synth._upper(inf), (V - 1, 1))
    gtR = np.zeros((V, 3, 3)); gtt = np.zeros((V, 3))
    R, t = np.eye(3), np.zeros(3)
    verts = np.zeros((V, 7))
    for j in range(V):
        gtR[j], gtt[j] = R, t
        verts[j] = np.concatenate([t, synth._R_to_quat(R)])
        t = t + R @ tz
        R = R @ Rz
    pairs = [(int(i), int(i) + 63) for i in np.linspace(0, V - 64, n_true).astype(int)]
    ids = np.asarray(pairs, dtype=np.int32)
    lm = np.zeros((len(pairs), 7)); li = np.zeros((len(pairs), 21))
    for k, (a, b) in enumerate(pairs):
        _, inf2 = synth._se3_info(rng, 0.04, 0.02)
        lm[k] = synth._se3_rel_meas(gtR[a], gtt[a], gtR[b], gtt[b], rng, 0.02, 0.01)
        li[k] = synth._upper(inf2)
    g = PoseGraph(3, verts, odom_meas, odom_info, ids, lm, li, dict(name="helix", seed=seed, canonic_inliers=len(pairs)))
    return synth.inject_outliers(g, n_out, seed=seed)


def run(lib, out):
    import time
    from ipc_amd import capi
    capi.LIB_PATH = lib
    from ipc_amd.consensus import IPC, Config
    g = helix()
    eng = IPC(g, Config(6.251, 50, 6.251, 100, 50.0))
    eng.run()
    bits, acc = eng.run()
    sms, launches = eng.solver_time_ms()
    c = eng.cell_info()
    c = c[np.lexsort((c["j"], c["i"]))]
    L = (c["hi"] - c["lo"]).astype(np.float64)
    np.savez(out, bits=bits, acc=acc, chi=c["max_chi2"], it=c["iterations"], ev=c["evals"], ms=sms,
             pose_it=float((L * c["iterations"]).sum()), pose_ev=float((L * c["evals"]).sum()), n=len(c))


if __name__ == "__main__":
    if sys.argv[1] == "--run":
        run(sys.argv[2], sys.argv[3]); sys.exit(0)
    res = []
    for k, lib in enumerate(sys.argv[1:]):
        out = "/tmp/uce_%d.npz" % k
        subprocess.check_call([sys.executable, __file__, "--run", os.path.abspath(lib), out])
        res.append(np.load(out))
        r = res[-1]
        print("%-40s %9.1f ms  cells %d  pose-iterations %.4g  pose-evals %.4g  accepted %d" % (
            os.path.basename(lib), float(r["ms"]), int(r["n"]), float(r["pose_it"]), float(r["pose_ev"]), int(r["acc"].sum())))
    for r in res[1:]:
        same = np.array_equal(res[0]["bits"], r["bits"]) and np.array_equal(res[0]["chi"], r["chi"], equal_nan=True) \
            and np.array_equal(res[0]["it"], r["it"]) and np.array_equal(res[0]["ev"], r["ev"])
        print("identical to the first build:", bool(same))
