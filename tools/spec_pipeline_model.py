"""Discrete-event model of the faithful mode's candidate pipeline on the oracle's per-candidate iteration counts
(tests/golden/{c1,c2}_incremental_expected.npz): how long the run takes with W solves in flight when a solve costs
`tau` seconds per dog-leg iteration and nothing else (no contention, no launch cost).

  window   -- round-3 mid version: at most W uncommitted candidates, all from the committed state; an accept throws the
              window away.
  pipeline -- the shipped one (engine.hip spec_pump): results are parked, a finished accept becomes the state the
              candidates behind it start from while earlier solves still run (assumed to reject).

Usage: python tools/spec_pipeline_model.py [c1|c2] [tau_ms]      (DESIGN.md 4.3 quotes c2 at 0.4 ms, c1 at 0.51 ms)"""
import heapq
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def window_model(dec, it, tau, W):
    N = len(dec)
    now, committed, ver = 0.0, 0, 0
    running, parked, ev, busy = {}, {}, [], 0.0
    while committed < N:
        while len(running) < W:
            k = committed
            while k < N and (k in running or k in parked):
                k += 1
            if k >= N or k - committed >= W:
                break
            d = it[k] * tau
            running[k] = (now + d, ver)
            heapq.heappush(ev, (now + d, k, ver))
            busy += d
        if not ev:
            break
        ft, k, v = heapq.heappop(ev)
        if k not in running or running[k] != (ft, v):
            continue
        now = ft
        del running[k]
        parked[k] = v
        while committed in parked:
            parked.pop(committed)
            if dec[committed]:
                ver += 1
                for j in list(running):
                    f, _ = running.pop(j)
                    busy -= max(0.0, f - now)
                parked.clear()
            committed += 1
    return now, busy / W


def pipeline_model(dec, it, tau, W, ahead=64):
    N = len(dec)
    truth = np.concatenate([[0], np.cumsum(dec)])            # accepts before position k
    acc_pos = [k for k in range(N) if dec[k]]
    now, committed, head = 0.0, 0, 0
    lineage = ()                                             # accepts (committed or assumed) the tip stands for
    running, parked, ev, busy = {}, {}, [], 0.0
    while committed < N:
        while len(running) < W:
            k = head
            while k < N and (k in running or k in parked):
                k += 1
            if k >= N or k - committed >= ahead:
                break
            d = it[k] * tau
            running[k] = (now + d, lineage)
            heapq.heappush(ev, (now + d, k, lineage))
            busy += d
            head = k + 1
        if not ev:
            break
        ft, k, lin = heapq.heappop(ev)
        if k not in running or running[k] != (ft, lin):
            continue
        now = ft
        del running[k]
        correct = lin == tuple(acc_pos[:truth[k]])
        if not correct:
            head = min(head, k)
        else:
            parked[k] = lin
            if dec[k]:
                lineage = lin + (k,)
                for j in list(running):
                    if j > k:
                        f, _ = running.pop(j)
                        busy -= max(0.0, f - now)
                for j in list(parked):
                    if j > k:
                        del parked[j]
                head = k + 1
        while committed in parked:
            parked.pop(committed)
            committed += 1
    return now, busy / W


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "c2"
    tau = float(sys.argv[2]) * 1e-3 if len(sys.argv) > 2 else (0.4e-3 if which == "c2" else 0.51e-3)
    z = np.load(os.path.join(ROOT, "tests", "golden", "%s_incremental_expected.npz" % which))
    dec, it = z["decision"].astype(int), z["iterations"].astype(float)
    print("%s: %d candidates, %d accepted, %d iterations, tau %.2f ms -> one at a time %.2f s" % (
        which, len(dec), int(dec.sum()), int(it.sum()), tau * 1e3, it.sum() * tau))
    for W in (2, 3, 4, 8, 12, 16):
        a, ab = window_model(dec, it, tau, W)
        b, bb = pipeline_model(dec, it, tau, W)
        print("  %2d in flight: window %.2f s (slots busy %.0f %%)   pipeline %.2f s (slots busy %.0f %%)" % (
            W, a, 100 * ab / a, b, 100 * bb / b))


if __name__ == "__main__":
    main()
