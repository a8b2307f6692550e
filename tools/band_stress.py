#!/usr/bin/env python
"""Repeats ipc_debug_band_solve on one system and reports how many distinct results come back (must be 1).
usage: python tools/band_stress.py nb m W workgroups repetitions"""
import ctypes as C
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

spec = importlib.util.spec_from_file_location("tb", os.path.join(ROOT, "tests", "test_gpu_band.py"))
tb = importlib.util.module_from_spec(spec)
spec.loader.exec_module(tb)


def main():
    nb, m, W, wgs, reps = (int(a) for a in sys.argv[1:6])
    from ipc_amd import capi
    lib = capi.load()
    S, rhs = tb._random_system(nb, m, W, 17)
    sysm = np.ascontiguousarray(tb._pack(S, rhs, nb, m, W))
    ref = np.linalg.solve(S, rhs)
    seen = {}
    worst = 0.0
    for _ in range(reps):
        x = np.zeros(nb + m - 1)
        info = C.c_int(0)
        capi.check(lib.ipc_debug_band_solve(nb, m, W, sysm.ctypes.data_as(C.c_void_p), wgs, x.ctypes.data_as(C.c_void_p), C.byref(info)))
        seen[x.tobytes()] = seen.get(x.tobytes(), 0) + 1
        worst = max(worst, np.abs(x - ref).max())
    print("nb %d m %d W %d workgroups %d: %d repetitions, %d distinct results %s, worst error vs numpy %.2e"
          % (nb, m, W, wgs, reps, len(seen), sorted(seen.values(), reverse=True), worst))


if __name__ == "__main__":
    main()
