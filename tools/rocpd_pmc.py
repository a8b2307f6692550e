#!/usr/bin/env python
"""Per-kernel PMC totals from a rocprofv3 rocpd SQLite result (one counter per pass)."""
import sqlite3
import sys


def main(paths):
    print("counter,kernel,dispatches,sum_value,avg_value_per_dispatch")
    for path in paths:
        cur = sqlite3.connect(path).cursor()
        for r in cur.execute("select counter_name, kernel_name, count(*), sum(value), avg(value) from "
                             "counters_collection group by counter_name, kernel_name order by sum(value) desc"):
            print('%s,"%s",%d,%.3f,%.3f' % r)


if __name__ == "__main__":
    main(sys.argv[1:])
