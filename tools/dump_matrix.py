#!/usr/bin/env python
"""Solve one workload on the GPU and save the consistency matrix, the accepted set and the per-cell
records (sorted by (i, j)) as a compressed .npz -- the reference a re-written kernel is compared
with (tools/compare_matrix.py).

usage: python tools/dump_matrix.py C4 gpurun_out/c4_ref.npz
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from bench import build_workload
from ipc_amd.consensus import IPC


def main(workload, path):
    g, cfg, desc = build_workload(workload)
    eng = IPC(g, cfg, device=0)
    t0 = time.perf_counter()
    bits, acc = eng.run()
    dt = time.perf_counter() - t0
    c = eng.cell_info()
    order = np.lexsort((c["j"], c["i"]))
    c = c[order]
    sms, launches = eng.solver_time_ms()
    np.savez_compressed(path, bits=bits, accepted=acc, i=c["i"].astype(np.int32), j=c["j"].astype(np.int32),
                        max_chi2=c["max_chi2"], chi2_total=c["chi2_total"].astype(np.float32),
                        iterations=c["iterations"].astype(np.int16), evals=c["evals"].astype(np.int16),
                        flags=c["flags"].astype(np.int8), wall_s=dt, solver_ms=sms, launches=launches, desc=desc)
    print("%s: %d cells, %.2f s wall, %.1f ms solver, %d launches, %d accepted -> %s" % (
        workload, len(c), dt, sms, launches, int(acc.sum()), path))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
