#!/usr/bin/env python
"""Critical path of a faithful run from the pipeline's event log (IPC_SPEC_LOG=<file>): for every accepted candidate, what
the time since the previous accept's state became available was spent on.
usage: python tools/spec_chain.py log.csv[.gz] [first_accept last_accept]
log lines: solve,t_launch_us,t_collect_us,pos,cand,expect,state_pos,stale(0 kept,1 stale,2 aborted),agree,iterations,cluster,device_us
           tentative,t_us,pos,cand      verdict,t_us,pos,cand,agree"""
import gzip
import sys
from collections import Counter


def main():
    path = sys.argv[1]
    f = gzip.open(path, "rt") if path.endswith(".gz") else open(path)
    solves, tent, verdicts = [], [], []
    for l in f:
        p = l.strip().split(",")
        if p[0] == "solve":
            solves.append(dict(t0=float(p[1]), t1=float(p[2]), pos=int(p[3]), cand=int(p[4]), expect=int(p[5]), state=int(p[6]),
                               stale=int(p[7]), agree=int(p[8]), it=int(p[9]), clu=int(p[10]), dev=float(p[11])))
        elif p[0] == "tentative":
            tent.append((float(p[1]), int(p[2])))
        elif p[0] == "verdict":
            verdicts.append((float(p[1]), int(p[2]), int(p[4])))
    acc_pos = [p for _, p, a in verdicts if a]
    t_tent = {}
    for t, p in tent:
        t_tent[p] = t                                   # the LAST time the state of an accept at p was made
    by_pos = {}
    for s in solves:
        by_pos.setdefault(s["pos"], []).append(s)
    lo = int(sys.argv[2]) if len(sys.argv) > 3 else 1
    hi = int(sys.argv[3]) if len(sys.argv) > 3 else len(acc_pos)
    cat = Counter()
    tot = 0.0
    rows = []
    for i in range(max(lo, 1), min(hi, len(acc_pos))):
        pp, p = acc_pos[i - 1], acc_pos[i]
        if pp not in t_tent or p not in t_tent:
            continue
        ta, tb = t_tent[pp], t_tent[p]
        link = tb - ta
        # the kept accept solve of p on the state of pp
        good = [s for s in by_pos.get(p, []) if s["state"] == pp and s["agree"] and s["stale"] == 0]
        if not good:
            cat["(no kept solve on the previous accept's state found)"] += link; tot += link
            continue
        g = good[-1]
        wait_launch = max(0.0, g["t0"] - ta)
        solve = g["t1"] - max(g["t0"], ta) if g["t0"] >= ta else g["t1"] - ta
        after = tb - g["t1"]
        tot += link
        cat["solve of the accepted candidate (launch to collect)"] += solve
        cat["collect -> its state made"] += after
        # what lay between the previous state and this solve's launch: other solves at positions in (pp, p) that the chain waited for
        between = [s for s in solves if pp < s["pos"] < p and s["state"] == pp and s["t1"] <= g["t0"] + 50 and s["t1"] >= ta]
        ea_rej = [s for s in between if s["expect"] == 1 and not s["agree"] and s["stale"] == 0]
        if wait_launch > 300 and ea_rej:
            cat["wait: expected accepts in between that rejected (%s)" % "serial"] += wait_launch
        elif wait_launch > 300:
            cat["wait: other (launch later than 0.3 ms after the state)"] += wait_launch
        else:
            cat["launch within 0.3 ms of the state"] += wait_launch
        rows.append((link, wait_launch, solve, after, p, g["expect"], len(ea_rej), g["it"], g["dev"]))
    print("accepts %d..%d: %.2f s on the chain, %.2f ms per accept" % (lo, hi, tot * 1e-6, tot * 1e-3 / max(1, len(rows))))
    for k, v in cat.most_common():
        print("  %-75s %8.2f s  %5.1f %%" % (k, v * 1e-6, 100.0 * v / tot))
    kept = [s for s in solves if s["stale"] == 0]
    print("kept solves: expected accept -> accept %d (%.1f ms on the device), expected accept -> reject %d (%.1f ms), expected reject -> accept %d, -> reject %d (%.1f ms)" % (
        sum(1 for s in kept if s["expect"] == 1 and s["agree"]), 1e-3 * sum(s["dev"] for s in kept if s["expect"] == 1 and s["agree"]) / max(1, sum(1 for s in kept if s["expect"] == 1 and s["agree"])),
        sum(1 for s in kept if s["expect"] == 1 and not s["agree"]), 1e-3 * sum(s["dev"] for s in kept if s["expect"] == 1 and not s["agree"]) / max(1, sum(1 for s in kept if s["expect"] == 1 and not s["agree"])),
        sum(1 for s in kept if s["expect"] == 0 and s["agree"]), sum(1 for s in kept if s["expect"] == 0 and not s["agree"]),
        1e-3 * sum(s["dev"] for s in kept if s["expect"] == 0 and not s["agree"]) / max(1, sum(1 for s in kept if s["expect"] == 0 and not s["agree"]))))
    rows.sort(reverse=True)
    print("longest links (ms): link, wait for launch, solve, collect->state, position, expectation, expected accepts rejected in between, iterations, device ms")
    for r in rows[:12]:
        print("  %8.1f %8.1f %8.1f %6.2f  pos %5d  expect %d  EA-rejected-between %d  it %d  dev %.1f" % (r[0] * 1e-3, r[1] * 1e-3, r[2] * 1e-3, r[3] * 1e-3, r[4], r[5], r[6], r[7], r[8] * 1e-3))


if __name__ == "__main__":
    main()
