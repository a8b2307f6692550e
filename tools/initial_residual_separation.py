import sys, os
sys.path.insert(0, '.')
import numpy as np, bench
from ipc_amd.consensus import IPC
for wl in ("C2", "C1"):
    g, cfg, _ = bench.build_workload(wl)
    eng = IPC(g, cfg); order = eng.candidate_order(); eng.reset()
    rows = []
    for k in order:
        ok, info = eng.agreementCheck(int(k), with_info=True)
        rows.append((ok, info.chi2_initial, info.iterations, info.n_cluster_loops, info.max_chi2))
    r = np.array(rows, dtype=float)
    acc = r[:, 0] == 1
    print(wl, "accepted", int(acc.sum()), "chi2_initial of accepted: max %.3g  p99 %.3g  median %.3g" % (r[acc, 1].max(), np.percentile(r[acc, 1], 99), np.median(r[acc, 1])))
    for T in (1e2, 1e3, 1e4, 1e5):
        rej = ~acc
        print("   threshold %.0e: rejected above %d of %d (%.1f %% of reject iterations)   accepted above %d" % (
            T, int((r[rej, 1] > T).sum()), int(rej.sum()), 100 * r[rej & (r[:, 1] > T), 2].sum() / r[rej, 2].sum(), int((r[acc, 1] > T).sum())))
    eng.close()
