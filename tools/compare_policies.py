#!/usr/bin/env python
"""Run one workload under two kernel policies (IPC_SE3_POLICY / IPC_SE2_POLICY values) and compare the
per-cell results: decisions, max chi2, iteration counts.  usage: compare_policies.py C4m block default
(or environment settings: compare_policies.py C2 "IPC_TERMINATE_EPS=0" default)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from bench import build_workload
from ipc_amd.consensus import IPC


_touched = set()


def run(g, cfg, pol):
    """pol: "default", a policy string, or KEY=VALUE[,KEY=VALUE...] environment settings."""
    key = "IPC_SE%d_POLICY" % g.dim
    for k in _touched | {key}:
        os.environ.pop(k, None)
    if "=" in pol:
        for kv in pol.split(";"):
            k, v = kv.split("=", 1)
            os.environ[k] = v
            _touched.add(k)
    elif pol != "default":
        os.environ[key] = pol
    eng = IPC(g, cfg, device=0)
    t0 = time.perf_counter()
    bits, acc = eng.run()
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    bits, acc = eng.run()
    dt2 = time.perf_counter() - t0
    c = eng.cell_info()
    c = c[np.lexsort((c["j"], c["i"]))]
    sms, nl = eng.solver_time_ms()
    print("%-10s first %.3f s, second %.3f s (solver %.1f ms, %d launches), %d cells, %d accepted" % (
        pol, dt, dt2, sms, nl, len(c), int(acc.sum())), flush=True)
    eng.close()
    return bits, acc, c


def main(workload, pa, pb):
    g, cfg, desc = build_workload(workload)
    print(desc, flush=True)
    ba, aa, ca = run(g, cfg, pa)
    bb, ab, cb = run(g, cfg, pb)
    th = np.where(ca["i"] == ca["j"], cfg.fast_reject_th, cfg.slow_reject_th)
    da = ~(ca["max_chi2"] > th)
    db = ~(cb["max_chi2"] > th)
    rel = np.abs(ca["max_chi2"] - cb["max_chi2"]) / np.maximum(np.abs(ca["max_chi2"]), 1e-300)
    conv = (ca["flags"] & 1).astype(bool) & (cb["flags"] & 1).astype(bool)
    print("decisions differing: %d of %d; accepted sets equal: %s; bits equal: %s" % (
        int((da != db).sum()), len(ca), bool(np.array_equal(aa, ab)), bool(np.array_equal(ba, bb))))
    print("evals per cell: %.2f vs %.2f; iterations per cell %.2f vs %.2f" % (ca["evals"].mean(), cb["evals"].mean(),
                                                                             ca["iterations"].mean(), cb["iterations"].mean()))
    print("max rel chi2 diff (both terminated): %.3e; (all): %.3e; iterations equal on %.4f of the cells; NaN a/b %d/%d" % (
        float(np.nanmax(rel[conv])) if conv.any() else 0.0, float(np.nanmax(rel)), float((ca["iterations"] == cb["iterations"]).mean()),
        int(np.isnan(ca["max_chi2"]).sum()), int(np.isnan(cb["max_chi2"]).sum())))
    bad = np.argsort(-np.nan_to_num(rel, nan=1e9))[:8]
    for k in bad:
        print("  cell (%d,%d) L=%d chi2 %.9g vs %.9g  it %d/%d flags %d/%d" % (
            ca["i"][k], ca["j"][k], ca["hi"][k] - ca["lo"][k], ca["max_chi2"][k], cb["max_chi2"][k], ca["iterations"][k],
            cb["iterations"][k], ca["flags"][k], cb["flags"][k]))


if __name__ == "__main__":
    main(*sys.argv[1:4])
