import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
from ipc_amd import capi
capi.LIB_PATH = os.path.join(os.getcwd(), "ipc_amd", "libipc_dbg_timing.so")
from bench import build_workload
from ipc_amd.consensus import IPC
g, cfg, desc = build_workload(sys.argv[1] if len(sys.argv) > 1 else "C1")
eng = IPC(g, cfg, device=0)
eng.run()
out = np.zeros(32, dtype=np.uint64)
eng.lib.ipc_dbg_read(eng.h, out.ctypes.data_as(C.c_void_p), 32)
tA, tB, tC, tT, its, ev = [float(x) for x in out[:6]]
tot = tA + tB + tC + tT
print("cycles (thread 0 of every cell): A %.1f%%  B %.1f%%  C %.1f%%  trials+eval %.1f%%" % (100*tA/tot, 100*tB/tot, 100*tC/tot, 100*tT/tot))
print("per iteration: A %.0f  B %.0f  C %.0f cycles; per evaluation: %.0f cycles; iterations %.0f evals %.0f" % (tA/its, tB/its, tC/its, tT/ev, its, ev))
names = ["update+publish", "barrier1", "neighbour+errors", "loop_eval+wave_sum", "barrier2", "gather"]
for tag, base in (("wave 0", 8), ("last wave", 16)):
    t = [float(x) for x in out[base:base + 6]]
    print(tag, "per evaluation:", "  ".join("%s %.0f" % (n, v / ev) for n, v in zip(names, t)), " total %.0f" % (sum(t) / ev))
tb = [float(x) for x in out[24:29]]
print("wave 0 per iteration, phase B:", "  ".join("%s %.0f" % (n, v / its) for n, v in zip(["partials", "packed reduce", "barrier", "solve", "barrier"], tb)))
