#!/usr/bin/env python
"""Faithful incremental run of a bench workload through a given build of the library: time, iteration / trial counts and
a digest of every candidate's (decision, iterations, tries, max chi2 bits) -- two builds that are meant to be
bit-identical must print the same digest.  usage: python tools/lib_incremental.py <lib.so> [C1|C2|C4s] [reps]"""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    lib, which = os.path.abspath(sys.argv[1]), sys.argv[2] if len(sys.argv) > 2 else "C1"
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    from ipc_amd import capi
    capi.LIB_PATH = lib
    import bench
    from ipc_amd.consensus import IPC
    g, cfg, _ = bench.build_workload(which)
    eng = IPC(g, cfg)
    order = eng.candidate_order()
    best = 1e30
    for _ in range(reps):
        eng.reset()
        eng.agreementCheck(order[0])
        eng.reset()
        h = hashlib.sha256()
        it = tr = acc = 0
        rows = []
        t0 = time.perf_counter()
        for k in order:
            ok, info = eng.agreementCheck(k, with_info=True)
            rows.append((k, ok, info.iterations, info.tries, info.max_chi2, info.n_cluster_loops, info.lo, info.hi, info.chi2_initial))
            h.update(np.array([ok, info.iterations, info.tries], dtype=np.int64).tobytes())
            h.update(np.float64(info.max_chi2).tobytes())
            it += info.iterations; tr += info.tries; acc += ok
        best = min(best, time.perf_counter() - t0)
        if os.environ.get("IPC_DUMP_RUN"):             # per-candidate records of every repetition (to find where two runs part)
            np.save(os.environ["IPC_DUMP_RUN"] + ".rep%d.npy" % _, np.array(rows, dtype=np.float64))
            print("   rep", _, h.hexdigest()[:16], flush=True)
    print("%-28s %-4s %8.3f s  %7.1f candidates/s  accepted %d iterations %d tries %d digest %s" % (
        os.path.basename(lib), which, best, g.N / best, acc, it, tr, h.hexdigest()[:16]), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
