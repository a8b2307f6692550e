#!/usr/bin/env python
"""Compare EVERY solved cell of a bench workload (GPU, through the C ABI) with the CPU oracle on all host cores:
decisions (max chi2 <= threshold) and max chi2.  Too long for the test suite (C2: 531 545 cells, ~8 min on the
16 CPUs a GPU box grants), so it is run by hand and its summary is committed under profiles/.

usage (GPU box): python tools/full_oracle_sweep.py C2 out.json [--max-seconds 1500] [--seed 0]
Cells are visited in random order in chunks, so a run cut short by --max-seconds is an unbiased sample."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    from bench import build_workload
    from ipc_amd.consensus import IPC
    from oracle import oracle as O
    workload, out_path = sys.argv[1], sys.argv[2]
    max_s = float(sys.argv[sys.argv.index("--max-seconds") + 1]) if "--max-seconds" in sys.argv else 1e9
    g, cfg, desc = build_workload(workload)
    eng = IPC(g, cfg, device=0)
    bits, acc = eng.run()
    cells = eng.cell_info()
    O.build()
    poses = O.propagate(g.dim, g.odom_meas)
    import bench
    cores = bench.effective_cores()                    # (the GPU boxes show 256 logical CPUs and grant 16)
    seed = int(sys.argv[sys.argv.index("--seed") + 1]) if "--seed" in sys.argv else 0
    rng = np.random.default_rng(seed)
    order = rng.permutation(len(cells))
    th_all = np.where(cells["i"] == cells["j"], cfg.fast_reject_th, cfg.slow_reject_th)
    L = cells["hi"] - cells["lo"]
    nl = np.where(cells["i"] == cells["j"], 1, 2)
    cap_all = np.where(cells["i"] == cells["j"], cfg.fast_reject_iter_base, cfg.slow_reject_iter_base) * np.where(L + nl > 100, 5, 1)
    done = diff = n_cap = n_nan = 0
    worst_conv = worst_cap = 0.0
    examples = []
    t0 = time.perf_counter()
    chunk = max(2048, 32 * cores)
    for s in range(0, len(order), chunk):
        idx = order[s:s + chunk]
        mx, its, used = O.pair_cells_mt(g.dim, g.odom_meas, g.odom_info, cfg.s_factor, poses, g.loop_ids, g.loop_meas,
                                        g.loop_info, cells["i"][idx], cells["j"][idx], cfg.fast_reject_iter_base,
                                        cfg.slow_reject_iter_base, cores)
        c = cells[idx]
        th = th_all[idx]
        d = (~(mx > th)) != (~(c["max_chi2"] > th))
        for k in np.nonzero(d)[0][:4]:
            examples.append(dict(i=int(c["i"][k]), j=int(c["j"][k]), oracle=float(mx[k]), gpu=float(c["max_chi2"][k]),
                                 threshold=float(th[k]), oracle_iterations=int(its[k]), gpu_iterations=int(c["iterations"][k])))
        diff += int(d.sum())
        conv = (its < cap_all[idx]) & (c["iterations"] < cap_all[idx])
        rel = np.abs(mx - c["max_chi2"]) / np.maximum(np.abs(mx), 1e-12)
        n_nan += int(np.isnan(rel).sum())
        rel = np.nan_to_num(rel, nan=0.0)
        if conv.any():
            worst_conv = max(worst_conv, float(rel[conv].max()))
        if (~conv).any():
            worst_cap = max(worst_cap, float(rel[~conv].max()))
            n_cap += int((~conv).sum())
        done += len(idx)
        if time.perf_counter() - t0 > max_s:
            break
    dt = time.perf_counter() - t0
    out = dict(workload=workload, desc=desc, order_seed=seed, solved_cells=int(len(cells)), cells_compared=done,
               decisions_differing=diff, worst_rel_chi2_diff_converged=worst_conv,
               cells_at_iteration_cap_on_either_side=n_cap, worst_rel_chi2_diff_at_cap=worst_cap, nan_cells=n_nan,
               oracle_threads=cores, oracle_seconds=dt, oracle_cells_per_s=done / dt, first_differences=examples[:16],
               accepted=int(acc.sum()))
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
