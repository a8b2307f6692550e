#!/usr/bin/env python
"""How many persistent cluster solves really run abreast: reads a rocprofv3 kernel trace (rocpd SQLite) of a faithful
run and prints, for the `cluster_persist_kernel` launches, the duration statistics, the time-averaged number in flight
on the DEVICE (from the start / end stamps of the trace, not from the host's launch-to-collect clocks), the share of the
run with k solves in flight, the workgroups resident on average, and the idle gap of each stream between two launches.

usage: python tools/pipeline_concurrency.py <results.db>"""
import sqlite3
import sys

import numpy as np


def main(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    sel = "start, end, grid_x / workgroup_x" + (", " + qcol if qcol else ", 0")
    rows = db.execute("select " + sel + " from kernels where name like '%cluster_persist_kernel%' order by start").fetchall()
    if not rows:
        print("no cluster_persist_kernel launches in", path)
        return
    a = np.array(rows, dtype=np.int64)
    start, end, wgs, q = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    dur = (end - start) * 1e-3
    span = (end.max() - start.min()) * 1e-3
    print("launches %d   span %.1f ms   duration us: mean %.0f  median %.0f  p90 %.0f  max %.0f   workgroups per launch: mean %.1f max %d" % (
        len(a), span * 1e-3, dur.mean(), np.median(dur), np.percentile(dur, 90), dur.max(), wgs.mean(), wgs.max()))
    ev = sorted([(s, 1, w) for s, w in zip(start, wgs)] + [(e, -1, -w) for e, w in zip(end, wgs)])
    t_prev, k, w = ev[0][0], 0, 0
    hist, wg_time = {}, 0.0
    for t, dk, dw in ev:
        hist[k] = hist.get(k, 0) + (t - t_prev)
        wg_time += w * (t - t_prev)
        k += dk
        w += dw
        t_prev = t
    tot = float(sum(hist.values()))
    mean_k = sum(kk * v for kk, v in hist.items()) / tot
    print("in flight on the device: mean %.2f   resident workgroups: mean %.1f" % (mean_k, wg_time / tot))
    print("share of the run with k in flight: " + "  ".join("%d: %.1f%%" % (kk, 100.0 * hist[kk] / tot) for kk in sorted(hist)))
    if qcol:
        gaps = []
        for s in np.unique(q):
            m = q == s
            st, en = start[m], end[m]
            gaps.extend(((st[1:] - en[:-1]) * 1e-3).tolist())
        gaps = np.array(gaps) if gaps else np.zeros(1)
        print("%d %ss; gap between two launches of one of them, us: mean %.0f median %.0f p90 %.0f" % (
            len(np.unique(q)), qcol, gaps.mean(), np.median(gaps), np.percentile(gaps, 90)))


if __name__ == "__main__":
    main(sys.argv[1])
