#!/usr/bin/env python
"""Compare two tools/dump_matrix.py results of the same workload (e.g. two kernel generations)."""
import sys

import numpy as np


def main(pa, pb, th_fast=6.251, th_slow=6.251):
    a, b = np.load(pa), np.load(pb)
    assert np.array_equal(a["i"], b["i"]) and np.array_equal(a["j"], b["j"]), "different cell lists"
    th = np.where(a["i"] == a["j"], float(th_fast), float(th_slow))
    da, db = ~(a["max_chi2"] > th), ~(b["max_chi2"] > th)
    rel = np.abs(a["max_chi2"] - b["max_chi2"]) / np.maximum(np.abs(a["max_chi2"]), 1e-300)
    term = (a["flags"] & 1).astype(bool) & (b["flags"] & 1).astype(bool)
    print("cells %d; solver ms %.1f vs %.1f; accepted %d vs %d (sets equal: %s); matrices equal: %s" % (
        len(th), float(a["solver_ms"]), float(b["solver_ms"]), int(a["accepted"].sum()), int(b["accepted"].sum()),
        bool(np.array_equal(a["accepted"], b["accepted"])), bool(np.array_equal(a["bits"], b["bits"]))))
    print("decisions differing: %d; rel chi2 diff: max %.3e (both terminated: %.3e), > 1e-5 on %d cells; "
          "cap-hit cells %d vs %d; flags&2 %d vs %d" % (
              int((da != db).sum()), float(np.nanmax(rel)), float(np.nanmax(rel[term])), int((rel > 1e-5).sum()),
              int((~(a["flags"] & 1).astype(bool)).sum()), int((~(b["flags"] & 1).astype(bool)).sum()),
              int(((a["flags"] & 2) != 0).sum()), int(((b["flags"] & 2) != 0).sum())))
    for k in np.where(da != db)[0][:20]:
        print("  differs: cell (%d,%d) chi2 %.9g vs %.9g it %d/%d flags %d/%d" % (
            a["i"][k], a["j"][k], a["max_chi2"][k], b["max_chi2"][k], a["iterations"][k], b["iterations"][k],
            a["flags"][k], b["flags"][k]))
    print("iterations: mean %.2f vs %.2f; evals mean %.2f vs %.2f" % (
        a["iterations"].mean(), b["iterations"].mean(), a["evals"].mean(), b["evals"].mean()))


if __name__ == "__main__":
    main(*sys.argv[1:])
