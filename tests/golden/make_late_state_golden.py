#!/usr/bin/env python
"""Oracle expectations for LATE states of the faithful runs of BASELINE configs[3] / [4] (round 6).

The CPU oracle's own run of IPC::agreementCheck over every candidate (reference src/simulation.cpp:34-47) gets through
1 205 of C4's 4 450 candidates and 3 239 of C5's 25 000 in 45 min each (make_incremental_golden.py); the part behind that
-- clusters of 1 000 to 3 328 accepted loops, 12 000 to 180 000 pose unknowns in the oracle's own (pose-space) system --
it cannot reach by running.  But the state of the reference's IPC object is only the vertex estimates and
_max_consensus_set (include/ipc/consensus.hpp:23-32), and ONE agreementCheck from a given state is seconds to hours.

  1. tools/late_state_dump.py (on the GPU box) runs the whole loop on the GPU and dumps (window poses, consensus set)
     in front of chosen late checks -> gpurun_out/late_states_<tag>.npz
  2. this script (build container, CPU only) puts the oracle's IncrementalIPC into each dumped state
     (oracle_ipc_set_state) and runs oracle_ipc_agreement_check for that candidate: computeIndependentSubgraph
     (src/consensus.cpp:124-171) over the dumped set, the dog-leg over the whole cluster (src/consensus_utils.cpp:7-22),
     the per-edge chi2 test.  One process per position; results are appended to a .jsonl as they finish.
  3. `--assemble` writes tests/golden/<tag>_late_states.npz: inputs (candidate, consensus set, window poses) and the
     oracle's outputs (decision, cluster span and size, iterations, max edge chi2) + what the GPU run recorded there.

tests/test_gpu_late_states.py injects the same states into the engine (ipc_incremental_set_state) and compares.

The oracle runs these with oracle_set_wide_dots(1): the same envelope Cholesky, dot products in eight partial sums (the
serial sum is a 4-cycle dependency chain per multiply-add: C5's 180 000-unknown systems would take a minute per
factorisation).  test_oracle_kat.py holds the two forms against each other.

usage: python tests/golden/make_late_state_golden.py c4|c5 [--workers 6] [--only q,q,...] [--max-iterations-gpu 200]
       python tests/golden/make_late_state_golden.py c4|c5 --assemble
"""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import numpy as np

WORKLOAD = {"c4": "C4", "c5": "C5", "c4m": "C4m", "r2k": "R2k"}


def window_to_poses(dim, w):
    """The fixture's window form -> engine / oracle poses ([n, 3] or [n, 12] R row-major, t)."""
    if dim == 2:
        return w
    q = w[:, :4] / np.linalg.norm(w[:, :4], axis=1, keepdims=True)
    qw, qx, qy, qz = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.stack([1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw),
                  2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw),
                  2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)], axis=1)
    return np.concatenate([R, w[:, 4:7]], axis=1)


DUMP_SUFFIX = os.environ.get("IPC_LATE_DUMP_SUFFIX", "")       # (a second dump of the same run: evenly spaced re-synchronisation points)


def dump_path(tag):
    return os.path.join(ROOT, "gpurun_out", "late_states_%s%s.npz" % (tag, DUMP_SUFFIX))


def work(args):
    tag, i = args
    from oracle import oracle as O
    import bench
    d = np.load(dump_path(tag))
    g, cfg, _ = bench.build_workload(WORKLOAD[tag])
    q = int(d["positions"][i])
    k = int(d["order"][q])
    rec = d["records"][q]
    lo, hi = int(rec["lo"]), int(rec["hi"])
    cns = d["cns"][d["cns_off"][i]:d["cns_off"][i + 1]]
    w = d["window"][d["window_off"][i]:d["window_off"][i + 1]]
    assert w.shape[0] == hi - lo + 1
    O.set_wide_dots(True)
    inc = O.IncrementalIPC(g.dim, g.odom_meas, g.odom_info, cfg.s_factor, cfg.fast_reject_th, cfg.fast_reject_iter_base,
                           cfg.slow_reject_th, cfg.slow_reject_iter_base, g.loop_ids, g.loop_meas, g.loop_info)
    poses = inc.poses()                                   # open loop (outside the window nothing is read)
    poses[lo:hi + 1] = window_to_poses(g.dim, w)
    inc.set_state(poses, cns)
    t0 = time.perf_counter()
    ok, info = inc.agreement_check(k)
    dt = time.perf_counter() - t0
    out = dict(tag=tag, i=int(i), q=q, k=k, decision=bool(ok), lo=info["lo"], hi=info["hi"], cluster=info["cluster"],
               iterations=info["iterations"], max_chi2=info["max_chi2"], seconds=round(dt, 1),
               gpu=dict(decision=bool(rec["ok"]), lo=lo, hi=hi, cluster=int(rec["cluster"]), iterations=int(rec["iterations"]),
                        max_chi2=float(rec["max_chi2"]), flags=int(rec["flags"])))
    with open(os.path.join(ROOT, "gpurun_out", "late_oracle_%s.jsonl" % tag), "a") as f:
        f.write(json.dumps(out) + "\n")
    return out


def forward(args):
    """From dumped state i the oracle runs ON ITS OWN through the candidates that follow (its state evolves by its own
    accepts) up to position `stop`: one record per candidate, appended to late_forward_<tag>.jsonl.  Stops early when its
    decision differs from the GPU run's (from there on the two states differ and nothing compares)."""
    tag, i, stop = args
    from oracle import oracle as O
    import bench
    d = np.load(dump_path(tag))
    g, cfg, _ = bench.build_workload(WORKLOAD[tag])
    O.set_wide_dots(True)
    inc = O.IncrementalIPC(g.dim, g.odom_meas, g.odom_info, cfg.s_factor, cfg.fast_reject_th, cfg.fast_reject_iter_base,
                           cfg.slow_reject_th, cfg.slow_reject_iter_base, g.loop_ids, g.loop_meas, g.loop_info)
    order = d["order"]
    if i < 0:
        # from the END of the oracle's own prefix run (c4_ / c5_incremental_expected.npz): its poses up to the last vertex a
        # checked candidate touches, odometry beyond (nothing behind the last accept's window was ever optimised)
        exp = np.load(os.path.join(HERE, "%s_incremental_expected.npz" % tag))
        poses = inc.poses()
        m = exp["poses"].shape[0]
        poses[:m] = exp["poses"]
        for v in range(m, poses.shape[0]):
            poses[v] = O.pose_mul(g.dim, poses[v - 1], O.meas_to_pose(g.dim, g.odom_meas[v - 1]))
        inc.set_state(poses, exp["consensus"])
        q0 = len(exp["order"])
    else:
        q0 = int(d["positions"][i])
        rec = d["records"][q0]
        lo, hi = int(rec["lo"]), int(rec["hi"])
        if "window_lo" in d.files:                        # (evenly spaced re-synchronisation points dump a wider window)
            lo, hi = int(d["window_lo"][i]), int(d["window_hi"][i])
        poses = inc.poses()
        # (the window of THIS candidate's cluster; the candidates that follow reach at most a little further, into poses
        # that are pure odometry on top of the window's last pose -- as in the run itself, see below)
        poses[lo:hi + 1] = window_to_poses(g.dim, d["window"][d["window_off"][i]:d["window_off"][i + 1]])
        for v in range(hi + 1, poses.shape[0]):
            poses[v] = O.pose_mul(g.dim, poses[v - 1], O.meas_to_pose(g.dim, g.odom_meas[v - 1]))
        inc.set_state(poses, d["cns"][d["cns_off"][i]:d["cns_off"][i + 1]])
    path = os.path.join(ROOT, "gpurun_out", "late_forward_%s.jsonl" % tag)
    n, t0 = 0, time.perf_counter()
    for q in range(q0, stop):
        k = int(order[q])
        t1 = time.perf_counter()
        ok, info = inc.agreement_check(k)
        r = d["records"][q]
        out = dict(start=int(i), q=q, k=k, decision=bool(ok), lo=info["lo"], hi=info["hi"], cluster=info["cluster"],
                   iterations=info["iterations"], max_chi2=info["max_chi2"], seconds=round(time.perf_counter() - t1, 2),
                   gpu_decision=bool(r["ok"]), gpu_max_chi2=float(r["max_chi2"]))
        with open(path, "a") as f:
            f.write(json.dumps(out) + "\n")
        n += 1
        if bool(ok) != bool(r["ok"]):
            break
    return dict(tag=tag, start=int(i), first=q0, done=n, stop=stop, seconds=round(time.perf_counter() - t0, 1))


def assemble(tag):
    d = np.load(dump_path(tag))
    res = {}
    with open(os.path.join(ROOT, "gpurun_out", "late_oracle_%s.jsonl" % tag)) as f:
        for line in f:
            r = json.loads(line)
            res[r["i"]] = r
    idx = sorted(res)
    cns, cns_off, win, win_off = [], [0], [], [0]
    for i in idx:
        c = d["cns"][d["cns_off"][i]:d["cns_off"][i + 1]]
        w = d["window"][d["window_off"][i]:d["window_off"][i + 1]]
        cns.append(c); cns_off.append(cns_off[-1] + len(c))
        win.append(w); win_off.append(win_off[-1] + len(w))
    R = [res[i] for i in idx]
    # the forward stretches (--forward): from each dumped state, and from the end of its own prefix run, the oracle went on BY
    # ITSELF through the candidates that follow -- one record per position, the first one kept where stretches overlap
    fwd = {}
    fpath = os.path.join(ROOT, "gpurun_out", "late_forward_%s.jsonl" % tag)
    if os.path.exists(fpath):
        for line in open(fpath):
            r = json.loads(line)
            fwd.setdefault(r["q"], r)
    fq = sorted(fwd)
    F = [fwd[q] for q in fq]
    path = os.path.join(HERE, "%s_late_states.npz" % tag)
    np.savez_compressed(
        path, workload=WORKLOAD[tag], position=np.array([r["q"] for r in R], dtype=np.int32),
        candidate=np.array([r["k"] for r in R], dtype=np.int32), reason=np.array([str(d["reasons"][i]) for i in idx]),
        cns=np.concatenate(cns).astype(np.int32), cns_off=np.array(cns_off), window=np.concatenate(win), window_off=np.array(win_off),
        decision=np.array([r["decision"] for r in R], dtype=np.uint8), lo=np.array([r["lo"] for r in R], dtype=np.int32),
        hi=np.array([r["hi"] for r in R], dtype=np.int32), cluster=np.array([r["cluster"] for r in R], dtype=np.int32),
        iterations=np.array([r["iterations"] for r in R], dtype=np.int32), max_chi2=np.array([r["max_chi2"] for r in R]),
        oracle_seconds=np.array([r["seconds"] for r in R]),
        gpu_decision=np.array([r["gpu"]["decision"] for r in R], dtype=np.uint8),
        gpu_iterations=np.array([r["gpu"]["iterations"] for r in R], dtype=np.int32),
        gpu_max_chi2=np.array([r["gpu"]["max_chi2"] for r in R]), gpu_flags=np.array([r["gpu"]["flags"] for r in R], dtype=np.int32),
        fwd_position=np.array(fq, dtype=np.int32), fwd_start=np.array([r["start"] for r in F], dtype=np.int32),
        fwd_decision=np.array([r["decision"] for r in F], dtype=np.uint8), fwd_lo=np.array([r["lo"] for r in F], dtype=np.int32),
        fwd_hi=np.array([r["hi"] for r in F], dtype=np.int32), fwd_cluster=np.array([r["cluster"] for r in F], dtype=np.int32),
        fwd_iterations=np.array([r["iterations"] for r in F], dtype=np.int32), fwd_max_chi2=np.array([r["max_chi2"] for r in F]),
        fwd_oracle_seconds=float(sum(r["seconds"] for r in F)),
        # the whole run the states were taken from (GPU): per-candidate records in processing order, for the full-run test
        run_decision=d["records"]["ok"], run_cluster=d["records"]["cluster"], run_lo=d["records"]["lo"], run_hi=d["records"]["hi"],
        run_max_chi2=d["records"]["max_chi2"], run_final_consensus=d["final_consensus"],
        loop_ids_checksum=d["loop_ids_checksum"], meas_checksum=d["meas_checksum"])
    if F:
        fw = max(abs(r["max_chi2"] - r["gpu_max_chi2"]) / max(abs(r["max_chi2"]), 1e-12) for r in F if r["max_chi2"] == r["max_chi2"])
        print("%s forward stretches: %d positions (%d ... %d), decisions differing from the GPU run: %d, worst relative chi2 difference %.2e, "
              "%.0f s of oracle time" % (tag, len(F), fq[0], fq[-1], sum(r["decision"] != r["gpu_decision"] for r in F), fw,
                                         sum(r["seconds"] for r in F)))
    worst = max(abs(r["max_chi2"] - r["gpu"]["max_chi2"]) / max(abs(r["max_chi2"]), 1e-12) for r in R)
    print("%s: %d positions, decisions differing from the GPU run: %d, (lo, hi, cluster) differing: %d, worst relative chi2 "
          "difference %.2e, largest cluster %d loops, %.0f s of oracle time; %s (%d bytes)" %
          (tag, len(R), sum(r["decision"] != r["gpu"]["decision"] for r in R),
           sum((r["lo"], r["hi"], r["cluster"]) != (r["gpu"]["lo"], r["gpu"]["hi"], r["gpu"]["cluster"]) for r in R), worst,
           max(r["cluster"] for r in R), sum(r["seconds"] for r in R), os.path.relpath(path, ROOT), os.path.getsize(path)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--workers", type=int, default=6)
    ap.add_argument("--only", default="")
    ap.add_argument("--max-iterations-gpu", type=int, default=10 ** 9,
                    help="skip positions whose GPU solve took more dog-leg iterations than this (oracle time)")
    ap.add_argument("--assemble", action="store_true")
    ap.add_argument("--forward", type=int, default=0,
                    help="from every dumped state (and from the end of the oracle's own prefix run) the oracle runs on by itself "
                         "through at most this many candidates, or up to the next dumped state")
    a = ap.parse_args()
    if a.assemble:
        return assemble(a.tag)
    if a.forward > 0:
        d = np.load(dump_path(a.tag))
        pos = [int(q) for q in d["positions"]]
        n = len(d["order"])
        jobs = []
        exp = os.path.join(HERE, "%s_incremental_expected.npz" % a.tag)
        if os.path.exists(exp):
            q0 = len(np.load(exp)["order"])
            if q0 < pos[0]:
                jobs.append((a.tag, -1, min(pos[0], q0 + a.forward)))
        for i, q in enumerate(pos):
            nxt = pos[i + 1] if i + 1 < len(pos) else n
            jobs.append((a.tag, i, min(nxt, q + a.forward)))
        # (a stretch whose every position already has a record -- an earlier, interrupted invocation -- is not run again)
        fpath = os.path.join(ROOT, "gpurun_out", "late_forward_%s.jsonl" % a.tag)
        have = set(json.loads(line)["q"] for line in open(fpath)) if os.path.exists(fpath) else set()
        first = lambda j: pos[j[1]] if j[1] >= 0 else len(np.load(exp)["order"])
        jobs = [j for j in jobs if any(q not in have for q in range(first(j), j[2]))]
        jobs.sort(key=lambda j: -(j[2] - first(j)))
        print("%s: %d forward stretches on %d workers" % (a.tag, len(jobs), a.workers), flush=True)
        from multiprocessing import Pool
        with Pool(a.workers) as pool:
            for out in pool.imap_unordered(forward, jobs):
                print(json.dumps(out), flush=True)
        return
    d = np.load(dump_path(a.tag))
    todo = list(range(len(d["positions"])))
    if a.only:
        want = set(int(x) for x in a.only.split(","))
        todo = [i for i in todo if int(d["positions"][i]) in want]
    todo = [i for i in todo if int(d["records"][int(d["positions"][i])]["iterations"]) <= a.max_iterations_gpu]
    done = set()
    jl = os.path.join(ROOT, "gpurun_out", "late_oracle_%s.jsonl" % a.tag)
    if os.path.exists(jl):
        done = set(json.loads(line)["i"] for line in open(jl))
    todo = [i for i in todo if i not in done]
    # cheapest first (GPU iteration count x chain length), so results arrive early
    todo.sort(key=lambda i: int(d["records"][int(d["positions"][i])]["iterations"]) *
              (int(d["window_off"][i + 1]) - int(d["window_off"][i])))
    print("%s: %d positions to do on %d workers" % (a.tag, len(todo), a.workers), flush=True)
    from multiprocessing import Pool
    with Pool(a.workers) as pool:
        for out in pool.imap_unordered(work, [(a.tag, i) for i in todo]):
            print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
