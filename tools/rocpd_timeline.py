#!/usr/bin/env python
"""Dump the kernel (and memory-copy) intervals of a rocprofv3 rocpd result as a CSV timeline, relative to the first kernel.
usage: python tools/rocpd_timeline.py results.db [t0_ms t1_ms] > timeline.csv"""
import sqlite3
import sys


def cols(cur, table):
    return [r[1] for r in cur.execute("pragma table_info(%s)" % table).fetchall()]


def main(path, t0=None, t1=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    kc = cols(cur, "kernels")
    extra = [c for c in ("queue_id", "stream_id", "tid") if c in kc]
    base = cur.execute("select min(start) from kernels").fetchone()[0]
    q = "select start, end, name, grid_x / workgroup_x %s from kernels order by start" % "".join(", " + c for c in extra)
    print("# columns: kind,start_us,end_us,name,workgroups," + ",".join(extra))
    for r in cur.execute(q):
        a, b = (r[0] - base) * 1e-3, (r[1] - base) * 1e-3
        if t0 is not None and (b < t0 * 1e3 or a > t1 * 1e3):
            continue
        name = r[2].split("(")[0].replace("void ", "").replace("ipc::", "")
        print("k,%.1f,%.1f,%s,%d%s" % (a, b, name, r[3], "".join(",%s" % x for x in r[4:])))
    try:
        mc = cols(cur, "memory_copies")
        if mc:
            for r in cur.execute("select start, end, name, size from memory_copies order by start"):
                a, b = (r[0] - base) * 1e-3, (r[1] - base) * 1e-3
                if t0 is not None and (b < t0 * 1e3 or a > t1 * 1e3):
                    continue
                print("m,%.1f,%.1f,%s,%d" % (a, b, r[2], r[3]))
    except sqlite3.Error:
        pass


if __name__ == "__main__":
    a = sys.argv
    main(a[1], float(a[2]) if len(a) > 3 else None, float(a[3]) if len(a) > 3 else None)
