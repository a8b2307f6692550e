#!/usr/bin/env python
"""ONE run of the CPU oracle over EVERY solved cell of a bench workload on all granted host cores (the
cpu_baseline leg of bench.py times a stratified sample of ~1 % of C2's cells and extrapolates; BASELINE.md promised C1 - C2 in
full "when the host has the cores").  Prints one JSON line: the whole-workload rate, the rate the sample of bench.py gives in
the same process, decisions differing from the GPU over ALL cells, chi2 differences per class.
usage: python tools/cpu_full_workload.py [C2] > profiles/r6_c2_cpu_full.json     (C2: ~8 min on 16 cores)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import bench
from ipc_amd.consensus import IPC
from oracle import oracle as O


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
    g, cfg, desc = bench.build_workload(wl)
    eng = IPC(g, cfg, device=0)
    eng.run()
    cells = eng.cell_info()
    eng.run()
    eng.synchronize()
    t0 = time.perf_counter()
    eng.run()
    eng.synchronize()
    gpu_s = time.perf_counter() - t0
    cores = bench.effective_cores()
    poses = O.propagate(g.dim, g.odom_meas)
    L = (cells["hi"] - cells["lo"]).astype(np.int64)
    idx = np.argsort(-L, kind="stable")
    t0 = time.perf_counter()
    mx, its, used = O.pair_cells_mt(g.dim, g.odom_meas, g.odom_info, cfg.s_factor, poses, g.loop_ids, g.loop_meas, g.loop_info,
                                    cells["i"][idx], cells["j"][idx], cfg.fast_reject_iter_base, cfg.slow_reject_iter_base, cores)
    t_all = time.perf_counter() - t0
    th = np.where(cells["i"] == cells["j"], cfg.fast_reject_th, cfg.slow_reject_th)[idx]
    nl = np.where(cells["i"] == cells["j"], 1, 2)[idx]
    cap = np.where(cells["i"][idx] == cells["j"][idx], cfg.fast_reject_iter_base, cfg.slow_reject_iter_base) * np.where(L[idx] + nl > 100, 5, 1)
    gm = cells["max_chi2"][idx]
    mism = int(((~(mx > th)) != (~(gm > th))).sum())
    rel = np.abs(mx - gm) / np.maximum(np.abs(mx), 1e-300)
    conv = (its < cap) & (cells["iterations"][idx] < cap)
    sample = bench.cpu_baseline(g, cfg, cells, 30.0)
    out = dict(workload=wl, desc=desc, solved_cells=int(len(cells)), threads=int(used), logical_cpus=os.cpu_count(), granted_cores=cores,
               whole_workload_seconds=round(t_all, 2), whole_workload_cells_per_s=len(cells) / t_all,
               sample_leg_of_bench_cells_per_s=sample["value"], sample_leg=sample["sample"],
               sample_over_whole=sample["value"] / (len(cells) / t_all),
               gpu_ms_per_matrix=gpu_s * 1e3, gpu_over_cpu_all_cores_whole_workload=(len(cells) / gpu_s) / (len(cells) / t_all),
               decisions_differing_from_gpu_over_all_cells=mism,
               max_rel_chi2_diff={"converged": float(np.nanmax(rel[conv])), "at_iteration_cap": float(np.nanmax(rel[~conv])) if (~conv).any() else 0.0,
                                  "cells_converged": int(conv.sum()), "cells_at_iteration_cap": int((~conv).sum())},
               note="CPU = oracle/ipc_oracle.c (-O3, this repo's restatement of the reference path: g2o cannot be built here), one shared work "
                    "queue, longest chains first, ONE run over every solved cell")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
