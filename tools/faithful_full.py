#!/usr/bin/env python
"""Full faithful run (IPC::agreementCheck over every candidate, reference src/simulation.cpp:34-47) of a bench workload
with progress lines, so that a run that is cut short still says how far it got and at what rate.
usage: python tools/faithful_full.py <workload> [max_candidates] [progress_every]
Prints one JSON line at the end: seconds, candidates/s, accepted, largest cluster, digest of (decision, iterations, tries,
max chi2 bits) per candidate, and -- where tests/golden/<workload>_incremental_expected.npz exists -- how the oracle's
prefix compares (decisions differing, worst relative chi2 difference)."""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "C4"
    limit = int(sys.argv[2]) if len(sys.argv) > 2 else -1
    every = int(sys.argv[3]) if len(sys.argv) > 3 else 250
    import bench
    from ipc_amd.consensus import IPC
    g, cfg, desc = bench.build_workload(which)
    eng = IPC(g, cfg)
    order = eng.candidate_order()
    if limit > 0:
        order = order[:limit]
    eng.reset()
    eng.agreementCheck(order[0])
    eng.reset()
    fx = os.path.join(ROOT, "tests", "golden", "%s_incremental_expected.npz" % which.lower())
    exp = np.load(fx) if os.path.exists(fx) else None
    h = hashlib.sha256()
    it = tr = acc = big = 0
    differ, worst, worst_at = 0, 0.0, None
    t0 = time.perf_counter()
    tl = t0
    for q, k in enumerate(order):
        ok, info = eng.agreementCheck(int(k), with_info=True)
        h.update(np.array([ok, info.iterations, info.tries], dtype=np.int64).tobytes())
        h.update(np.float64(info.max_chi2).tobytes())
        it += info.iterations; tr += info.tries; acc += ok
        big = max(big, info.n_cluster_loops)
        if exp is not None and q < len(exp["order"]):
            differ += int(ok != bool(exp["decision"][q]))
            ref = float(exp["max_chi2"][q])
            err = abs(info.max_chi2 - ref) / max(abs(ref), 1e-12)
            if err > worst:
                worst, worst_at = err, dict(position=q, candidate=int(k), decision=bool(ok), iterations=info.iterations,
                                            oracle_iterations=int(exp["iterations"][q]), max_chi2=info.max_chi2, oracle_max_chi2=ref,
                                            cluster=info.n_cluster_loops, flags=info.flags)
        if (q + 1) % every == 0:
            now = time.perf_counter()
            print("  %6d / %d  %.1f s  (%.1f /s over the last %d)  accepted %d  cluster %d loops, chain %d..%d, %d iterations"
                  % (q + 1, len(order), now - t0, every / (now - tl), every, acc, info.n_cluster_loops, info.lo, info.hi,
                     info.iterations), flush=True)
            tl = now
    dt = time.perf_counter() - t0
    out = dict(workload=which, desc=desc, candidates=len(order), seconds=round(dt, 3), candidates_per_s=round(len(order) / dt, 2),
               accepted=int(acc), largest_cluster_loops=int(big), iterations=int(it), tries=int(tr), digest=h.hexdigest()[:16])
    if exp is not None:
        out["oracle_prefix"] = dict(candidates=int(min(len(exp["order"]), len(order))), decisions_differing=differ,
                                    worst_rel_chi2_diff=worst, worst_at=worst_at)
    print(json.dumps(out), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
