"""Phase split of the SE3 LDS-pose kernels (debug build with -DIPC_PHASE_TIMING, see wave_phase_timing.py):
s_memtime ticks (100 MHz) of lane 0 of the first and of the last wave of every team, summed per variant.
usage (GPU box): python tools/se3_phase_timing.py C4m"""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
from ipc_amd import capi
capi.LIB_PATH = os.path.join(os.getcwd(), "ipc_amd", "libipc_dbg_timing.so")
from bench import build_workload
from ipc_amd.consensus import IPC
g, cfg, desc = build_workload(sys.argv[1] if len(sys.argv) > 1 else "C4m")
eng = IPC(g, cfg, device=0)
eng.run()
out = np.zeros(4096, dtype=np.uint64)
eng.lib.ipc_dbg_read(eng.h, out.ctypes.data_as(C.c_void_p), 4096)
print("variant wave   cells    iters    %A   %B1(sweep) %B2(solve)  %C  %trials %commit | barrier%(of all)  ticks/iter  ticks/(iter*pose)*1e3  evals/iter")
for W in (1, 4):
    for M in range(1, 16):
        for half, name in ((0, "first"), (16, "last")):
            d = out[1024 + 32 * (M + 16 * (1 if W > 1 else 0)) + half:][:16].astype(np.float64)
            if d[10] == 0 or (W == 1 and half):
                continue
            tot = d[0] + d[1] + d[2] + d[3] + d[4] + d[5]
            print("%s%-3d %-5s %8d %9d  %5.1f %5.1f %5.1f %5.1f %5.1f %5.1f  | %5.1f   %9.1f  %8.2f  %5.2f" % (
                {1: "w", 4: "g"}[W], M, name, d[10], d[7], 100 * d[0] / tot, 100 * d[1] / tot, 100 * d[2] / tot,
                100 * d[3] / tot, 100 * d[4] / tot, 100 * d[5] / tot, 100 * d[6] / tot, tot / d[7], 1e3 * tot / d[9], d[8] / d[7]))
