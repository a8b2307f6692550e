#!/usr/bin/env python
"""Generates the committed expectations of the faithful incremental mode (IPC::agreementCheck loop,
reference src/consensus.cpp:43-75 driven by src/simulation.cpp:34-47) on the bench workloads.

The CPU oracle (oracle/ipc_oracle.c, test infrastructure) runs the whole candidate list of

  c1  bench.py workload C1 (INTEL-like SE2, V=1228, 256 true loops + 100 outliers, seed 100):
      clusters up to ~250 accepted loops, capacitance systems of ~760 unknowns
  c2  bench.py workload C2 (same graph + 1000 outliers, seed 1000)
  se3 small sphere SE3 (bench.py workload C4s: V=500, 60 true loops + 60 outliers): clusters of >= 40 loops
  c4m sphere2500-like SE3 (bench.py workload C4m: V=2500, every 10th true loop + 200 outliers): clusters up to 244
      loops = 1 464 unknowns (round 4; 181 s)
  c3  bench.py workload C3 (MIT-like SE2, V=808, 20 true loops + 5000 outliers): 5 020 candidates, 9 accepted (round 4; 101 s)
  c4  bench.py workload C4 (BASELINE configs[3]: sphere2500-like SE3, all 2 450 true loops + 2 000 outliers) -- a PREFIX:
      the candidates the oracle finishes within --seconds (round 5; clusters of > 1 000 loops, the banded capacitance solve)
  c5  bench.py workload C5 (BASELINE configs[4]: V=50 000 SE3 chain, 5 000 true loops + 20 000 local outliers) -- a PREFIX
  r2k bench.py workload R2k (SE2 spiral, V=2 000, 1 950 true loops of span 50 + 300 outliers) -- a PREFIX: the SE2
      instance of the banded large-cluster solver (one growing cluster, 3 x 3 blocks)

and records per candidate (in processing order): decision, cluster span lo/hi, cluster size,
max edge chi2.  The workloads themselves are regenerated from their seeds by ipc_amd.synth (the
fixtures hold only the expectations, a few KB each).  Run time here: c1 ~2 min, c2 ~10 min, se3 ~1 min.

usage: python tests/golden/make_incremental_golden.py [--seconds S] [c1] [c2] [se3] [c4m] [c3] [c4] [c5]
(--seconds: stop after the first candidate that ends beyond S seconds of oracle time; the fixture then holds that prefix)
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import numpy as np

from oracle import oracle as O

WORKLOAD = {"c1": "C1", "c2": "C2", "se3": "C4s", "c4m": "C4m", "c3": "C3", "c4": "C4", "c5": "C5", "r2k": "R2k"}


def run(tag, seconds=None):
    import bench
    g, cfg, desc = bench.build_workload(WORKLOAD[tag])
    inc = O.IncrementalIPC(g.dim, g.odom_meas, g.odom_info, cfg.s_factor, cfg.fast_reject_th,
                           cfg.fast_reject_iter_base, cfg.slow_reject_th, cfg.slow_reject_iter_base,
                           g.loop_ids, g.loop_meas, g.loop_info)
    order = O.candidate_order(g.loop_ids)
    n = len(order)
    dec = np.zeros(n, dtype=np.uint8)
    lo = np.zeros(n, dtype=np.int32)
    hi = np.zeros(n, dtype=np.int32)
    cl = np.zeros(n, dtype=np.int32)
    it = np.zeros(n, dtype=np.int32)
    mx = np.zeros(n)
    t0 = time.perf_counter()
    for q, k in enumerate(order):
        ok, info = inc.agreement_check(int(k))
        dec[q], lo[q], hi[q], cl[q], it[q], mx[q] = ok, info["lo"], info["hi"], info["cluster"], info["iterations"], info["max_chi2"]
        if seconds is not None and time.perf_counter() - t0 > seconds:
            n = q + 1
            break
    dt = time.perf_counter() - t0
    order, dec, lo, hi, cl, it, mx = order[:n], dec[:n], lo[:n], hi[:n], cl[:n], it[:n], mx[:n]
    poses = inc.poses()
    if seconds is not None:
        poses = poses[:int(hi.max()) + 1]       # (a prefix: the poses up to the last vertex any checked candidate touches)
    np.savez_compressed(os.path.join(HERE, "%s_incremental_expected.npz" % tag), order=order, decision=dec, lo=lo,
                        hi=hi, cluster=cl, iterations=it, max_chi2=mx, consensus=inc.consensus(),
                        poses=poses, oracle_seconds_authoring_container=dt,
                        loop_ids_checksum=np.int64(np.asarray(g.loop_ids, dtype=np.int64).sum()),
                        meas_checksum=float(np.asarray(g.loop_meas).sum()))
    print(tag, desc, "N=%d accepted=%d max cluster=%d oracle %.1f s (1 thread)" % (n, int(dec.sum()), int(cl.max()), dt))


if __name__ == "__main__":
    args = sys.argv[1:]
    seconds = None
    if "--seconds" in args:
        i = args.index("--seconds")
        seconds = float(args[i + 1])
        del args[i:i + 2]
    for tag in (args or ["c1", "se3", "c2"]):
        run(tag, seconds)
