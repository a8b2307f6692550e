"""Phase split of the SE2 wave / pair / quad kernels (debug build with -DIPC_PHASE_TIMING:
  python -c "import __graft_entry__ as g; g._build_lib('ipc_amd/libipc_dbg_timing.so', ['-DIPC_PHASE_TIMING'], 'build/dbg')"
then   python tools/wave_phase_timing.py C2   on the GPU box).  Cycles are s_memtime ticks (100 MHz)
of lane 0 / wave 0 of every cell, summed per kernel variant."""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
from ipc_amd import capi
capi.LIB_PATH = os.path.join(os.getcwd(), "ipc_amd", "libipc_dbg_timing.so")
from bench import build_workload
from ipc_amd.consensus import IPC
g, cfg, desc = build_workload(sys.argv[1] if len(sys.argv) > 1 else "C2")
eng = IPC(g, cfg, device=0)
eng.run()
out = np.zeros(4096, dtype=np.uint64)
eng.lib.ipc_dbg_read(eng.h, out.ctypes.data_as(C.c_void_p), 4096)
print("variant   cells   iters   %A   %B1(partials)  %B2(solve)  %C   %trials  %commit  wait%(of all)   ticks/iter  ticks/(iter*pose)*1e3  rejected-trials/iter  GN-trials/iter  GN-rejected/iter  big-sweeps/eval  ticks/big-sweep  ticks/small-sweep")
for W in (1, 2, 4):
    for M in range(1, 16):
        d = out[64 + 16 * (M + 16 * (W - 1)):][:16].astype(np.float64)
        if d[9] == 0:
            continue
        tot = d[0] + d[1] + d[2] + d[3] + d[4] + d[10]
        print("%s%-3d %8d %9d  %5.1f %5.1f %5.1f %5.1f %5.1f %5.1f   %5.1f   %8.1f  %8.2f  %6.3f  %6.3f  %6.3f  %6.3f  %8.0f %8.0f" % (
            {1: "w", 2: "p", 4: "q"}[W], M, d[9], d[6], 100 * d[0] / tot, 100 * d[1] / tot, 100 * d[2] / tot,
            100 * d[3] / tot, 100 * d[4] / tot, 100 * d[10] / tot, 100 * d[5] / tot, tot / d[6], 1e3 * tot / d[8], d[11] / d[6], d[12] / d[6], d[13] / d[6], d[14] / max(d[7] - d[9], 1), d[15] / max(d[14], 1), float(out[64 + 16 * (M + 16 * (W - 1)) + 1024]) / max(d[7] - d[9] - d[14], 1)))
