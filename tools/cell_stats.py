#!/usr/bin/env python
"""Per-cell statistics of one solve on the GPU (iterations, trials, evaluations by chain length)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from bench import build_workload
from ipc_amd.consensus import IPC


def main(workload="C2"):
    g, cfg, desc = build_workload(workload)
    eng = IPC(g, cfg, device=0)
    eng.run()
    c = eng.cell_info()
    L = c["hi"] - c["lo"]
    out = {"workload": desc, "cells": int(len(c))}
    for name, sel in (("diag", c["i"] == c["j"]), ("pair", c["i"] != c["j"])):
        d = c[sel]
        l = L[sel]
        out[name] = {
            "n": int(len(d)), "L_mean": float(l.mean()), "L_max": int(l.max()),
            "iterations_mean": float(d["iterations"].mean()), "iterations_max": int(d["iterations"].max()),
            "iterations_hist": np.bincount(np.minimum(d["iterations"], 60)).tolist(),
            "tries_mean": float(d["tries"].mean()), "evals_mean": float(d["evals"].mean()),
            "evals_max": int(d["evals"].max()),
            "terminated_frac": float((d["flags"] & 1).mean()), "fail_frac": float(((d["flags"] & 2) != 0).mean()),
            "hit_iteration_cap": int((~(d["flags"] & 1).astype(bool)).sum()),
            "accepted_frac": float((d["max_chi2"] <= (cfg.fast_reject_th if name == "diag" else cfg.slow_reject_th)).mean()),
            "pose_iterations": float((l * d["iterations"]).sum()), "pose_evals": float((l * d["evals"]).sum()),
        }
    print(json.dumps(out))


if __name__ == "__main__":
    main(*sys.argv[1:])
