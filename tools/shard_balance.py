"""Per-rank solve time of the row shards, emulated on ONE GPU (each rank's shard solved in turn;
no collective): shows the load balance of the row partition (IPC_ROW_BALANCE=cost|cyclic) and the fixed per-step cost.
usage: [IPC_ROW_BALANCE=cyclic] python tools/shard_balance.py [C2] [world ...]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from bench import build_workload
from ipc_amd.consensus import IPC
from ipc_amd.dist import EngineBackend

wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
worlds = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8]
if worlds[0] != 1:
    worlds = [1] + worlds                             # (the speed-up is quoted against the one-rank step)
g, cfg, desc = build_workload(wl)
eng = IPC(g, cfg, device=0)
b = EngineBackend(eng)
for world in worlds:
    rpr = (eng.N + world - 1) // world
    gathered = b.empty_words(world * rpr * eng.words)
    bits, acc = b.empty_words(eng.N * eng.words), b.empty_bytes(eng.N)
    times, first = [], []
    for r in range(world):
        upper = gathered[r * rpr * eng.words:(r + 1) * rpr * eng.words]
        best = 1e9
        for rep in range(3):
            # rep 0 is a SINGLE-SHOT matrix: (rank, world) has just changed, so the step plans its cell lists (two passes and a
            # host read-back) and solves them in list order; reps 1, 2 are repeated steps on the cached lists, slow cells first
            torch.cuda.synchronize(); t = time.perf_counter()
            with b.stream_ctx():
                b.solve_rows(r, world, upper)
                b.assemble(gathered, world, bits)
                b.set_max(bits, acc)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            if rep == 0:
                first.append(dt * 1e3)
            else:
                best = min(best, dt)
        times.append(best * 1e3)
    if world == 1:
        t1, f1 = max(times), max(first)
    print("world %d: per-rank step ms min %.1f max %.1f mean %.1f -> bound on the speed-up over world 1: %.2fx | single-shot (planning included, "
          "no slow-cells-first order): max %.1f ms -> %.2fx" % (world, min(times), max(times), sum(times) / world, t1 / max(times), max(first), f1 / max(first)))
