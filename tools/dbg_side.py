import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
from ipc_amd import graphio, capi, synth
capi.LIB_PATH = sys.argv[1]
from ipc_amd.consensus import IPC, Config
g = synth._se2_graph(700, 24, seed=102, laps=4.0, name="medium")
g = synth.inject_outliers(g, 16, seed=2)
g = g.subset(np.array([14]))
eng = IPC(g, Config(), device=0)
eng.run()
cells = eng.cell_info()
print(cells)
if "side" in sys.argv[1]:
    out = np.zeros(8 * 4)
    eng.lib.ipc_dbg_read(eng.h, out.ctypes.data_as(C.c_void_p), out.size)
    print("rho newChi curChi linGain hHh bh hgnNorm delta")
    print(out.reshape(-1, 8)[:2])
