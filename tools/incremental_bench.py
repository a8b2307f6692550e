"""Faithful incremental mode (reference agreementCheck loop) on the GPU vs the CPU oracle, and its
agreement with the matrix mode.  Usage: python tools/incremental_bench.py [C1|C2] [--cpu]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from ipc_amd.consensus import IPC
    which = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "C1"
    g, cfg, _ = bench.build_workload(which)            # the bench's own workloads (same seeds as the committed fixtures)
    eng = IPC(g, cfg)
    order = eng.candidate_order()
    eng.reset()
    eng.agreementCheck(order[0])            # warm-up (allocations, first launches)
    eng.reset()
    acc = np.zeros(g.N, dtype=np.uint8)
    infos = []
    t0 = time.perf_counter()
    for k in order:
        ok, info = eng.agreementCheck(k, with_info=True)
        acc[k] = ok
        infos.append((info.hi - info.lo, info.n_cluster_loops, info.iterations, info.tries))
    t_gpu = time.perf_counter() - t0
    t0 = time.perf_counter()
    poses, finfo = eng.final_optimize(acc)
    t_fin = time.perf_counter() - t0
    t0 = time.perf_counter()
    _, acc_m = eng.run()
    t_mat = time.perf_counter() - t0
    infos = np.array(infos)
    out = dict(workload=which, N=int(g.N), V=int(g.V), gpu_incremental_s=t_gpu,
               gpu_checks_per_s=g.N / t_gpu, accepted_incremental=int(acc.sum()),
               max_cluster_loops=int(infos[:, 1].max()), total_iterations=int(infos[:, 2].sum()),
               total_tries=int(infos[:, 3].sum()),
               final_map_s=t_fin, final_chi2=finfo.chi2_total, final_iterations=finfo.iterations,
               matrix_mode_s=t_mat, accepted_matrix=int(acc_m.sum()),
               agreement_matrix_vs_incremental=float((acc == acc_m).mean()))
    if "--cpu" in sys.argv:
        from oracle import oracle as O
        O.build()
        inc = O.IncrementalIPC(g.dim, g.odom_meas, g.odom_info, cfg.s_factor, cfg.fast_reject_th,
                               cfg.fast_reject_iter_base, cfg.slow_reject_th, cfg.slow_reject_iter_base,
                               g.loop_ids, g.loop_meas, g.loop_info)
        t0 = time.perf_counter()
        acc_c = inc.run()
        out["cpu_oracle_incremental_s"] = time.perf_counter() - t0
        out["decisions_differing_from_cpu"] = int((acc_c != acc).sum())
    print(json.dumps(out))
    eng.close()                                        # (IPC_PERSIST_PROF=1: the leader's phase clocks go to stderr here)


if __name__ == "__main__":
    main()
