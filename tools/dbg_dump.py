import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from ipc_amd import graphio, capi
capi.LIB_PATH = sys.argv[1]
from ipc_amd.consensus import IPC, Config
g = graphio.read_g2o("tests/golden/small_se2_spoiled_n6_seed3.g2o")
eng = IPC(g, Config(), device=0)
eng.run()
cells = eng.cell_info()
for c in cells:
    if c["i"] == c["j"] and c["i"] in (0, 3, 11):
        print(sys.argv[1][-9:], c)
