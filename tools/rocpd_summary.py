#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite result (kernel trace) into a small text/CSV table.

usage: python tools/rocpd_summary.py gpurun_out/prof_r1/c2_results.db > profiles/r1_c2_kernel_stats.csv
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
        "max(workgroup_x), sum(grid_x/workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("kernel,calls,total_ns,avg_ns,min_ns,max_ns,pct,arch_vgpr,accum_vgpr,sgpr,lds_bytes,scratch_bytes,wg_size,workgroups")
    for r in rows:
        print('"%s",%d,%d,%.0f,%d,%d,%.2f,%d,%d,%d,%d,%d,%d,%d' % (
            r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / total, r[6], r[7], r[8], r[9], r[10], r[11], r[12]))
    # The cell-solver launches of one step run on several streams and overlap (the tail of one bin
    # under the next bins), so their durations do not add up to wall time.  The figure that
    # corresponds to bench.py's HIP-event time is the busy time of their union, per step.
    iv = cur.execute("select start, end from kernels where name like '%_cells_kernel%' or name like '%_lds_kernel%' or name like '%_wave_kernel%' "
                     "or name like '%_group_kernel%' order by start").fetchall()
    # one k_scatter_bits launch per step (the two k_plan launches only run in the step that builds the cell lists: round 5)
    steps = cur.execute("select count(*) from kernels where name like 'k_scatter_bits%'").fetchone()[0]
    busy, lo, hi = 0, None, None
    for a, b in iv:
        if hi is None or a > hi:
            if hi is not None:
                busy += hi - lo
            lo, hi = a, b
        else:
            hi = max(hi, b)
    if hi is not None:
        busy += hi - lo
    if steps:
        print('"cell solver kernels: busy time of the union of their intervals",%d,%d,%.0f,,,,,,,,,,' % (steps, busy, busy / steps))


if __name__ == "__main__":
    main(sys.argv[1])
