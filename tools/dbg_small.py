import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from ipc_amd import graphio, capi
if len(sys.argv) > 1:
    capi.LIB_PATH = sys.argv[1]
from ipc_amd.consensus import IPC, Config, unpack_bits
g = graphio.read_g2o("tests/golden/small_se2_spoiled_n6_seed3.g2o")
exp = np.load("tests/golden/small_se2_expected.npz")
eng = IPC(g, Config(), device=0)
bits, acc = eng.run()
C = unpack_bits(bits, eng.N)
print(sys.argv[1:], (C != exp["okmat"]).sum(), "bit diffs")
cells = eng.cell_info()
bad = 0
for c in cells:
    ref = exp["maxchi2"][c["i"], c["j"]]
    rel = abs(ref - c["max_chi2"]) / max(abs(ref), 1e-12)
    if rel > 1e-6:
        bad += 1
        if bad < 15: print(c, ref, rel)
print(bad, "of", len(cells))
