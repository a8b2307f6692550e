"""Which cells run long?  Concentration of the slow cells (iterations >= 200) over candidates."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from bench import build_workload
from ipc_amd.consensus import IPC
g, cfg, desc = build_workload(sys.argv[1] if len(sys.argv) > 1 else "C2")
eng = IPC(g, cfg, device=0)
eng.run()
c = eng.cell_info()
L = c["hi"] - c["lo"]
cost = c["iterations"].astype(np.float64) * L
slow = c[c["iterations"] >= 200]
print("cells", len(c), "slow", len(slow), "share of pose-iterations in slow cells %.3f" % (cost[c["iterations"] >= 200].sum() / cost.sum()))
cnt = np.bincount(np.concatenate([slow["i"], slow["j"]]), minlength=eng.N)
top = np.argsort(-cnt)[:15]
print("top candidates by slow-cell count:", [(int(k), int(cnt[k])) for k in top])
print("candidates covering 50%% / 90%% of slow cells: %d / %d of %d" % (
    np.searchsorted(np.cumsum(np.sort(cnt)[::-1]), 0.5 * cnt.sum()) + 1,
    np.searchsorted(np.cumsum(np.sort(cnt)[::-1]), 0.9 * cnt.sum()) + 1, eng.N))
d = c[c["i"] == c["j"]]
dit = np.zeros(eng.N); dit[d["i"]] = d["iterations"]
dchi = np.zeros(eng.N); dchi[d["i"]] = d["max_chi2"]
p = c[c["i"] != c["j"]]
pred = np.maximum(dit[p["i"]], dit[p["j"]])
print("corr(pair iterations, max diag iterations) = %.3f" % np.corrcoef(p["iterations"], pred)[0, 1])
print("corr(pair iterations, log max diag chi2) = %.3f" % np.corrcoef(p["iterations"], np.log1p(np.maximum(dchi[p["i"]], dchi[p["j"]])))[0, 1])
