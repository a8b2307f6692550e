#!/usr/bin/env python
"""Generates the committed golden fixtures.  Runs ONLY in the authoring container (it executes
the reference's own outlier injector from /root/reference/scripts); the fixtures it writes are
data (inputs and expected outputs) and travel with the repo, the reference does not.

  *_clean.g2o                         synthetic clean graphs (ipc_amd.synth, fixed seeds)
  *_spoiled_*.g2o                     output of the reference's scripts/generateDataset.py on them
  *_expected.npz                      CPU-oracle results on the spoiled graphs (regression pin)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import numpy as np

from ipc_amd import graphio, synth
from oracle import oracle as O

REF_SCRIPT = "/root/reference/scripts/generateDataset.py"

CASES = [
    # name, clean generator, outliers, seed, extra flags, s_factor, fast_th, fast_it, slow_th, slow_it
    ("small_se2", synth.small_se2, 6, 3, [], 10.0, 6.251, 50, 11.345, 100),
    ("small_se2_local", synth.small_se2, 5, 9, ["--local"], 10.0, 6.251, 50, 11.345, 100),
    ("small_se2_group", synth.small_se2, 3, 5, ["-g", "2"], 10.0, 6.251, 50, 11.345, 100),
    ("small_se3", synth.small_se3, 5, 4, [], 50.0, 6.251, 50, 6.251, 100),
]


def main():
    for name, gen, n_out, seed, flags, s, fth, fit, sth, sit in CASES:
        g = gen()
        clean = os.path.join(HERE, name.split("_local")[0].split("_group")[0] + "_clean.g2o")
        graphio.write_g2o(clean, g)
        spoiled = os.path.join(HERE, "%s_spoiled_n%d_seed%d.g2o" % (name, n_out, seed))
        subprocess.check_call([sys.executable, REF_SCRIPT, "-i", clean, "-o", spoiled, "-n", str(n_out),
                               "--seed", str(seed)] + flags, stdout=subprocess.DEVNULL)
        gs = graphio.read_g2o(spoiled)
        ok, mx = O.consistency_matrix(gs.dim, gs.odom_meas, gs.odom_info, s, gs.loop_ids, gs.loop_meas,
                                      gs.loop_info, fth, fit, sth, sit)
        order = O.candidate_order(gs.loop_ids)
        acc = O.set_max(ok, order)
        inc = O.IncrementalIPC(gs.dim, gs.odom_meas, gs.odom_info, s, fth, fit, sth, sit, gs.loop_ids,
                               gs.loop_meas, gs.loop_info).run()
        np.savez_compressed(os.path.join(HERE, name + "_expected.npz"), okmat=ok, maxchi2=mx, order=order,
                            accepted=acc, incremental_accepted=inc,
                            params=np.array([s, fth, fit, sth, sit]), n_outliers=n_out, seed=seed)
        print(name, "N=%d accepted=%d incremental=%d" % (gs.N, acc.sum(), inc.sum()))


if __name__ == "__main__":
    main()
