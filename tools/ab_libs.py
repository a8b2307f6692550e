#!/usr/bin/env python
"""A/B builds of libipc_amd.so on one bench workload: solver time (best of --reps) and per-cell agreement with the
first build (bit-identical matrix / chi2 / iteration counts, or the number of differing decisions and the largest
relative chi2 difference).

usage (GPU box): python tools/ab_libs.py C2 ipc_amd/libipc_base.so ipc_amd/libipc_amd.so [--reps 3] [--env K=V ...]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def run(lib, workload, reps, out):
    from ipc_amd import capi
    capi.LIB_PATH = lib
    from bench import build_workload
    from ipc_amd.consensus import IPC
    g, cfg, _ = build_workload(workload)
    eng = IPC(g, cfg, device=0)
    best = 1e30
    for _ in range(reps + 1):
        bits, acc = eng.run()
        sms, launches = eng.solver_time_ms()
        best = min(best, sms)
    c = eng.cell_info()
    c = c[np.lexsort((c["j"], c["i"]))]
    th = np.where(c["i"] == c["j"], cfg.fast_reject_th, cfg.slow_reject_th)
    np.savez(out, bits=bits, acc=acc, chi=c["max_chi2"], it=c["iterations"], ev=c["evals"], ms=best, th=th, n=len(c),
             ci=c["i"], cj=c["j"], L=c["hi"] - c["lo"], fl=c["flags"])


if __name__ == "__main__":
    if sys.argv[1] == "--run":
        run(sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]); sys.exit(0)
    args = sys.argv[1:]
    reps = 3
    env = dict(os.environ)
    if "--reps" in args:
        k = args.index("--reps"); reps = int(args[k + 1]); del args[k:k + 2]
    if "--env" in args:
        k = args.index("--env")
        for kv in args[k + 1:]:
            a, b = kv.split("=", 1); env[a] = b
        del args[k:]
    workload, libs = args[0], args[1:]
    res = []
    for k, lib in enumerate(libs):
        out = "/tmp/ab_%d.npz" % k
        subprocess.check_call([sys.executable, __file__, "--run", os.path.abspath(lib), workload, str(reps), out], env=env)
        r = np.load(out)
        res.append(r)
        line = "%-34s %-5s %10.2f ms  cells %d  iterations %d  evals %d  accepted %d" % (
            os.path.basename(lib), workload, float(r["ms"]), int(r["n"]), int(r["it"].sum()), int(r["ev"].sum()), int(r["acc"].sum()))
        if k:
            a = res[0]
            ident = np.array_equal(a["bits"], r["bits"]) and np.array_equal(a["chi"], r["chi"], equal_nan=True) and np.array_equal(a["it"], r["it"])
            dec = int(((a["chi"] > a["th"]) != (r["chi"] > r["th"])).sum())
            rel = np.abs(a["chi"] - r["chi"]) / np.maximum(np.abs(a["chi"]), 1e-300)
            conv = (a["it"] == r["it"])
            line += "  | vs first: identical=%s decisions differing=%d accepted-set equal=%s max rel chi2 diff (same iteration count) %.2e" % (
                bool(ident), dec, bool(np.array_equal(a["acc"], r["acc"])), float(np.nanmax(rel[conv])) if conv.any() else 0.0)
        print(line, flush=True)
        if k and not ident:
            a = res[0]
            d = np.nonzero((a["it"] != r["it"]) | ~((a["chi"] == r["chi"]) | (np.isnan(a["chi"]) & np.isnan(r["chi"]))))[0]
            print("   %d cells differ; first ones (i j L | it chi2 evals flags first | second):" % len(d))
            for q in d[:12]:
                print("   %5d %5d %5d | %4d %.17g %4d %d | %4d %.17g %4d %d" % (a["ci"][q], a["cj"][q], a["L"][q], a["it"][q], a["chi"][q], a["ev"][q], a["fl"][q],
                                                                        r["it"][q], r["chi"][q], r["ev"][q], r["fl"][q]))
