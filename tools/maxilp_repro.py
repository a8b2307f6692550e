#!/usr/bin/env python
"""Reduced reproducer of the scheduling-strategy hazard.  The SE2 PAIR kernels (two waves per cell, LDS
mailbox) whose errors are recomputed (M > 8), built with `-mllvm -amdgpu-sched-strategy=max-ilp` and PLAIN mailbox
payload accesses (-DIPC_MAILBOX_PLAIN), solve nearly all their cells wrongly and differently from run to run; with the
relaxed-atomic payload accesses the library ships, both scheduling strategies give bit-identical results.  One
instantiation (REPRO_M, default p9), one workload (REPRO_WORKLOAD, default T700), ~20 s per build.

  build:  python tools/maxilp_repro.py build name=flag,flag ...   (each name -> ipc_amd/librepro_<name>_m<M>.so;
                                                                   `mllvm:X` stands for `-mllvm X`)
  run:    python tools/maxilp_repro.py run name ...               (GPU; the first name is the reference)
  all:    python tools/maxilp_repro.py demo                       (GPU box: builds and runs the four corners)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
M = int(os.environ.get("REPRO_M", "9"))
WORKLOAD = os.environ.get("REPRO_WORKLOAD", "T700")
POLICY = "w1,p%d,16x4,16x8,16x16" % M   # every chain of 65 .. 128 M poses goes through the pM pair kernel


def lib_of(name):
    return os.path.join(ROOT, "ipc_amd", "librepro_%s_m%d.so" % (name, M))


def build(specs):
    procs = []
    for spec in specs:
        name, _, fl = spec.partition("=")
        flags = ["-DIPC_PAIR_ONLY_M=%d" % M] + [f for f in fl.split(",") if f]
        flags = [x for f in flags for x in (["-mllvm", f[6:]] if f.startswith("mllvm:") else [f])]
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "build_variant.py"), lib_of(name), "se2_pair.hip"] + flags))
    for p in procs:
        assert p.wait() == 0


def run_one(lib, out):
    import numpy as np
    from ipc_amd import capi
    capi.LIB_PATH = lib
    os.environ["IPC_SE2_POLICY"] = POLICY
    from bench import build_workload
    from ipc_amd.consensus import IPC
    g, cfg, _ = build_workload(WORKLOAD)
    eng = IPC(g, cfg, device=0)
    bits, acc = eng.run()
    c = eng.cell_info()
    c = c[np.lexsort((c["j"], c["i"]))]
    np.savez(out, c=c, acc=acc)


def run(names):
    import numpy as np
    res = []
    for k, n in enumerate(names):
        out = "/tmp/repro_%d.npz" % k
        subprocess.check_call([sys.executable, __file__, "--one", lib_of(n), out])
        r = np.load(out)
        res.append(r)
        c = r["c"]
        L = c["hi"] - c["lo"]
        pair = (L > 64) & (L <= 128 * M)
        line = "%-22s cells %d (pM: %d) accepted %d iterations %d" % (n, len(c), int(pair.sum()), int(r["acc"].sum()), int(c["iterations"].sum()))
        if k:
            a = res[0]["c"]
            same = (a["max_chi2"] == c["max_chi2"]) & (a["iterations"] == c["iterations"])
            d = np.nonzero(~same)[0]
            line += " | differing from %s: %d cells (pM: %d, others: %d)" % (names[0], len(d), int((~same & pair).sum()), int((~same & ~pair).sum()))
            print(line)
            for q in d[:4]:
                print("     cell (%d, %d) L=%d: it %d chi2 %.17g | it %d chi2 %.17g" % (a["i"][q], a["j"][q], L[q], a["iterations"][q], a["max_chi2"][q],
                                                                                 c["iterations"][q], c["max_chi2"][q]))
        else:
            print(line)


def demo():
    ilp = "mllvm:-amdgpu-sched-strategy=max-ilp"
    build(["def=", "ilp=" + ilp, "plain_def=-DIPC_MAILBOX_PLAIN", "plain_ilp=-DIPC_MAILBOX_PLAIN," + ilp])
    run(["def", "ilp", "plain_def", "plain_ilp"])


if __name__ == "__main__":
    if sys.argv[1] == "demo":
        demo()
    elif sys.argv[1] == "--one":
        run_one(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run(sys.argv[2:])
