#!/usr/bin/env python
"""Oracle expectations for the harness's FINAL MAP at BASELINE sizes.

After the agreementCheck loop the reference divides the odometry information by s again, adds the accepted loops and runs
optimize(1000) over the whole graph with vertex 0 fixed (src/simulation.cpp:50-65).  The oracle restates that as one
oracle_solve_cell over the chain 0..V-1 with information (info * s) / s and the accepted loops, 1000 dog-leg iterations
(the cell solver applies consensus_utils.cpp's x5 rule, so it is given 200), from the open-loop poses.

Accepted sets = the faithful runs' consensus sets: C1 / C2 from the oracle's own full runs (c{1,2}_incremental_expected.npz),
C4 from the GPU run the late states were taken from (c4_late_states.npz: its decisions are oracle-checked on the prefix and
at the late positions; the set is an INPUT here, the oracle optimises whatever it is given).

usage: python tests/golden/make_final_map_golden.py [c1] [c2] [c4]
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import numpy as np

from oracle import oracle as O

WORKLOAD = {"c1": "C1", "c2": "C2", "c4": "C4", "c5": "C5"}


def accepted_set(tag, g):
    if tag in ("c1", "c2"):
        return np.load(os.path.join(HERE, "%s_incremental_expected.npz" % tag))["consensus"]
    return np.load(os.path.join(HERE, "%s_late_states.npz" % tag))["run_final_consensus"]


def run(tag):
    import bench
    g, cfg, desc = bench.build_workload(WORKLOAD[tag])
    cns = np.asarray(accepted_set(tag, g), dtype=np.int64)
    acc = np.zeros(g.N, dtype=np.uint8)
    acc[cns] = 1
    order = O.candidate_order(g.loop_ids)
    sel = [int(k) for k in order if acc[k]]                        # ipc_final_optimize: accepted candidates in processing order
    s = cfg.s_factor
    info = (np.asarray(g.odom_info) * s) / s                       # src/simulation.cpp:55-56 on top of robustifyVoters
    poses0 = O.propagate(g.dim, g.odom_meas)
    O.set_wide_dots(True)
    t0 = time.perf_counter()
    ref = O.solve_cell(g.dim, g.odom_meas, info, 1.0, poses0, 0, g.V - 1, np.asarray(g.loop_ids)[sel], np.asarray(g.loop_meas)[sel],
                       np.asarray(g.loop_info)[sel], 200, want_poses=True)
    dt = time.perf_counter() - t0
    O.set_wide_dots(False)
    np.savez_compressed(os.path.join(HERE, "%s_final_map_expected.npz" % tag), workload=WORKLOAD[tag], accepted=acc,
                        chi2_total=ref["chi2_final"], chi2_initial=ref["chi2_initial"], max_chi2=ref["max_chi2"],
                        iterations=ref["iterations"], terminated=ref["terminated"], poses=ref["poses"],
                        oracle_seconds_authoring_container=dt,
                        loop_ids_checksum=np.int64(np.asarray(g.loop_ids, dtype=np.int64).sum()),
                        meas_checksum=float(np.asarray(g.loop_meas).sum()))
    print("%s %s: %d accepted loops, chi2 %.6g -> %.9g, max edge chi2 %.6g, %d iterations (terminated %d), oracle %.1f s" %
          (tag, desc, len(sel), ref["chi2_initial"], ref["chi2_final"], ref["max_chi2"], ref["iterations"], ref["terminated"], dt))


if __name__ == "__main__":
    for tag in (sys.argv[1:] or ["c1", "c2"]):
        run(tag)
