import os, time, sys
sys.path.insert(0, '.')
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
os.system("lscpu | head -25; cat /proc/loadavg")
import numpy as np
import bench
from oracle import oracle as O
g, cfg, _ = bench.build_workload("C1")
poses = O.propagate(g.dim, g.odom_meas)
N = g.N
ii = np.arange(N, dtype=np.int32)
lo = g.loop_ids.min(axis=1); hi = g.loop_ids.max(axis=1)
L = hi - lo
idx = np.argsort(L)[: 2048].astype(np.int32)   # short diagonal cells
for T in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    n = min(len(idx), 64 * T)
    t0 = time.perf_counter()
    O.pair_cells_mt(g.dim, g.odom_meas, g.odom_info, cfg.s_factor, poses, g.loop_ids, g.loop_meas, g.loop_info, idx[:n] if False else np.resize(idx, n), np.resize(idx, n), cfg.fast_reject_iter_base, cfg.slow_reject_iter_base, T)
    dt = time.perf_counter() - t0
    print("threads %3d cells %5d  %.3f s  %.1f cells/s  per-thread %.2f" % (T, n, dt, n / dt, n / dt / T))
