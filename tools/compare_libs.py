"""A/B two builds of libipc_amd.so on one workload: per-cell results must match exactly when a
change is meant to be bit-identical.  Usage: python tools/compare_libs.py libA.so libB.so [C1|C2]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def dump(lib, which, out):
    import time
    import numpy as np
    from ipc_amd import capi, synth
    capi.LIB_PATH = lib
    from ipc_amd.consensus import IPC, Config
    n_out = {"C1": 100, "C2": 1000}[which]
    g = synth.inject_outliers(synth.intel_like(), n_out, seed=20260929)
    eng = IPC(g, Config(6.251, 50, 11.345, 100, 10.0))
    eng.run()
    t0 = time.perf_counter()
    bits, acc = eng.run()
    dt = time.perf_counter() - t0
    cells = eng.cell_info()
    order = np.lexsort((cells["j"], cells["i"]))
    np.savez(out, cells=cells[order], bits=bits, acc=acc, seconds=dt)


if __name__ == "__main__":
    if sys.argv[1] == "--dump":
        dump(sys.argv[2], sys.argv[3], sys.argv[4])
        sys.exit(0)
    import numpy as np
    a, b = sys.argv[1], sys.argv[2]
    which = sys.argv[3] if len(sys.argv) > 3 else "C1"
    outs = []
    for k, lib in enumerate((a, b)):
        out = "/tmp/cmp_%d.npz" % k
        subprocess.check_call([sys.executable, __file__, "--dump", os.path.abspath(lib), which, out])
        outs.append(np.load(out))
    A, B = outs
    ca, cb = A["cells"], B["cells"]
    res = dict(workload=which, cells=int(len(ca)), seconds_a=float(A["seconds"]), seconds_b=float(B["seconds"]),
               same_bits=bool(np.array_equal(A["bits"], B["bits"])), same_accepted=bool(np.array_equal(A["acc"], B["acc"])))
    for f in ("max_chi2", "chi2_total", "iterations", "tries", "flags", "evals"):
        x, y = ca[f], cb[f]
        if x.dtype.kind == "f":
            same = (x == y) | (np.isnan(x) & np.isnan(y))
        else:
            same = x == y
        res["diff_" + f] = int((~same).sum())
    pair = ca["i"] != ca["j"]
    res["pair_evals_mean_a"] = float(ca["evals"][pair].mean()); res["pair_evals_mean_b"] = float(cb["evals"][pair].mean())
    res["pair_iterations_mean"] = float(ca["iterations"][pair].mean()); res["pair_tries_mean"] = float(ca["tries"][pair].mean())
    print(json.dumps(res))
