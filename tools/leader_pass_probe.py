"""Leader phase clocks of the persistent solver against the chain length of the solves (one at a time, IPC_PERSIST_PROF=1):
how much of a chain pass is fixed cost and how much grows with the items per thread.  usage: python tools/leader_pass_probe.py <V>"""
import os, sys
sys.path.insert(0, '.')
os.environ["IPC_SPEC_WINDOW"] = "1"; os.environ["IPC_PERSIST_PROF"] = "1"
from ipc_amd import synth
from ipc_amd.consensus import IPC, Config
V = int(sys.argv[1])
g = synth._se2_graph(V, max(24, V // 16), seed=81, laps=3.0, name="inc")
g = synth.inject_outliers(g, 40, seed=2)
eng = IPC(g, Config()); order = eng.candidate_order(); eng.reset()
its = 0; spans = []
for k in order:
    ok, info = eng.agreementCheck(int(k), with_info=True); its += info.iterations; spans.append(info.hi - info.lo)
import numpy as np
print("V", V, "candidates", len(order), "iterations", its, "mean window", np.mean(spans), "max cluster", flush=True)
eng.close()
