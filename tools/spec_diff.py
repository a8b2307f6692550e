"""Per-candidate results of the faithful run under two IPC_SPEC_WINDOW settings (debugging aid): which candidates differ."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ipc_amd.consensus import IPC

def run(g, cfg, w):
    os.environ["IPC_SPEC_WINDOW"] = str(w)
    eng = IPC(g, cfg)
    order = eng.candidate_order()
    eng.reset()
    out = []
    for k in order:
        ok, info = eng.agreementCheck(k, with_info=True)
        out.append((int(ok), info.lo, info.hi, info.n_cluster_loops, info.iterations, info.tries, info.max_chi2))
    eng.close()
    return order, out

which = sys.argv[1] if len(sys.argv) > 1 else "C2"
wa, wb = int(sys.argv[2]) if len(sys.argv) > 2 else 1, int(sys.argv[3]) if len(sys.argv) > 3 else 8
g, cfg, _ = bench.build_workload(which)
order, a = run(g, cfg, wa)
for rep in range(int(sys.argv[4]) if len(sys.argv) > 4 else 2):
    _, b = run(g, cfg, wb)
    bad = [p for p in range(len(a)) if a[p] != b[p]]
    print("run", rep, "differing positions:", len(bad))
    for p in bad[:12]:
        prev_acc = max([q for q in range(p) if a[q][0]], default=-1)
        print("  pos", p, "cand", int(order[p]), "last accept before at pos", prev_acc, "\n    w%d" % wa, a[p], "\n    w%d" % wb, b[p])
