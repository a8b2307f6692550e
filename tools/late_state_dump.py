#!/usr/bin/env python
"""Dumps late states of a full faithful run (IPC::agreementCheck over every candidate, reference src/simulation.cpp:34-47)
for the CPU oracle to continue from (tests/golden/make_late_state_golden.py).

The oracle cannot reach the late part of BASELINE configs[3] / [4] on its own (C4: 1 205 of 4 450 candidates in 45 min,
C5: 3 239 of 25 000), but ONE check from a given state is minutes.  The state of the reference's IPC object is the vertex
estimates + _max_consensus_set (include/ipc/consensus.hpp:23-32), so this tool runs the whole loop on the GPU twice:

  pass 1  records every candidate's outcome and picks the positions: clusters >= --min-cluster loops, the largest cluster
          (an accept and a reject), candidates at the iteration cap, rejects whose own span is far wider than the band
          (arrowhead rows), accepts and rejects spread over the late range;
  pass 2  the same run again (bitwise: the digests are compared), downloading ipc_current_poses + ipc_consensus_set in
          front of each chosen check.

Output: gpurun_out/late_states_<workload>.npz -- per position: q, candidate, GPU outcome, consensus set, the poses of the
cluster's window lo..hi as (unit quaternion w x y z, translation) for SE3 / (x y theta) for SE2.
usage: python tools/late_state_dump.py <workload> [--positions N] [--min-cluster M] [--limit Q]
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from ipc_amd import capi
capi.export_recommended_environment()


def rot_to_quat(R):
    """[n, 9] row-major rotations -> [n, 4] unit quaternions (w x y z), w >= 0 (scipy; numpy only)."""
    from scipy.spatial.transform import Rotation
    q = Rotation.from_matrix(R.reshape(-1, 3, 3)).as_quat()          # x y z w
    q = np.concatenate([q[:, 3:4], q[:, :3]], axis=1)
    q[q[:, 0] < 0] *= -1
    return q


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


def run(eng, order, dump_at=None, every=1000, tag=""):
    eng.reset()
    eng.agreementCheck(int(order[0]))
    eng.reset()
    h = hashlib.sha256()
    rec = np.zeros(len(order), dtype=[("ok", "u1"), ("lo", "i4"), ("hi", "i4"), ("cluster", "i4"), ("iterations", "i4"),
                                      ("tries", "i4"), ("flags", "i4"), ("max_chi2", "f8"), ("chi2_total", "f8"),
                                      ("chi2_initial", "f8")])
    dumps = {}
    t0 = time.perf_counter()
    for q, k in enumerate(order):
        if dump_at is not None and q in dump_at:
            dumps[q] = (eng.current_poses(), eng.getMaxConsensusSet().copy())
        ok, info = eng.agreementCheck(int(k), with_info=True)
        rec[q] = (ok, info.lo, info.hi, info.n_cluster_loops, info.iterations, info.tries, info.flags, info.max_chi2,
                  info.chi2_total, info.chi2_initial)
        h.update(np.array([ok, info.iterations, info.tries], dtype=np.int64).tobytes())
        h.update(np.float64(info.max_chi2).tobytes())
        if (q + 1) % every == 0:
            print("  %s %6d / %d  %.1f s  accepted %d  cluster %d" % (tag, q + 1, len(order), time.perf_counter() - t0,
                                                                       int(rec["ok"][:q + 1].sum()), info.n_cluster_loops), flush=True)
    return rec, dumps, time.perf_counter() - t0, h.hexdigest()[:16]


def pick(rec, ids, order, cfg, want, min_cluster, n_slow=3, with_cap=True):
    n = len(rec)
    span = np.abs(ids[order, 1] - ids[order, 0])
    late = np.nonzero(rec["cluster"] >= min_cluster)[0]
    if late.size == 0:
        late = np.nonzero(rec["cluster"] >= rec["cluster"].max() // 2)[0]
    chosen, why = [], {}

    def add(q, reason):
        q = int(q)
        if q not in why:
            chosen.append(q)
            why[q] = reason

    acc, rej = late[rec["ok"][late] == 1], late[rec["ok"][late] == 0]
    big = rec["cluster"].max()
    near = np.nonzero(rec["cluster"] >= big - 2)[0]
    for ok, name in ((1, "largest cluster, accept"), (0, "largest cluster, reject")):
        c = near[rec["ok"][near] == ok]
        if c.size:
            add(c[-1], name)
    cap = cfg.slow_reject_iter_base * 5
    capped = late[(rec["iterations"][late] >= cap) & ((rec["flags"][late] & 1) == 0)]
    if capped.size and with_cap:
        add(capped[len(capped) // 2], "at the iteration cap (%d iterations)" % cap)
    typical = np.median(span[rec["ok"] == 1]) if (rec["ok"] == 1).any() else 1
    wide = rej[span[rej] > 8 * typical]
    if wide.size:
        add(wide[np.argmax(span[wide])], "arrowhead: rejected candidate of span %d against a typical %d" % (int(span[wide].max()), int(typical)))
        add(wide[len(wide) // 2], "arrowhead reject")
    slow = rej[np.argsort(rec["iterations"][rej])[-n_slow:]] if rej.size and n_slow > 0 else []
    for q in slow:
        add(q, "reject with many iterations (%d)" % int(rec["iterations"][q]))
    quick = rej[rec["iterations"][rej] <= np.percentile(rec["iterations"][rej], 30)] if rej.size else []
    # the rest: accepts and rejects spread evenly over the late range
    left = max(0, want - len(chosen))
    na = (left + 1) // 2
    for q in acc[np.linspace(0, len(acc) - 1, na).astype(int)] if len(acc) else []:
        add(q, "accept")
    left = max(0, want - len(chosen))
    pool = quick if len(quick) else rej
    for q in pool[np.linspace(0, len(pool) - 1, left).astype(int)] if len(pool) else []:
        add(q, "reject")
    chosen.sort()
    return chosen, why


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload")
    ap.add_argument("--positions", type=int, default=24)
    ap.add_argument("--min-cluster", type=int, default=1000)
    ap.add_argument("--limit", type=int, default=-1)
    ap.add_argument("--slow", type=int, default=3, help="rejects with the most iterations to include")
    ap.add_argument("--no-cap", action="store_true", help="no candidate at the iteration cap (hours of oracle time on C5)")
    ap.add_argument("--even", type=int, default=0, help="instead of the picks above: this many positions evenly spaced from --even-from to the end "
                                                        "(re-synchronisation points of the oracle's forward stretches, make_late_state_golden.py --forward)")
    ap.add_argument("--even-from", type=int, default=0)
    ap.add_argument("--suffix", default="", help="appended to the output file name")
    a = ap.parse_args()
    import bench
    from ipc_amd.consensus import IPC
    g, cfg, desc = bench.build_workload(a.workload)
    eng = IPC(g, cfg)
    order = eng.candidate_order()
    if a.limit > 0:
        order = order[:a.limit]
    ids = np.asarray(g.loop_ids, dtype=np.int64).reshape(-1, 2)
    rec1, _, dt1, dig1 = run(eng, order, tag="pass 1")
    if a.even > 0:
        chosen = sorted(set(int(q) for q in np.linspace(a.even_from, len(order) - 1, a.even + 1)[:-1]))
        why = {q: "evenly spaced" for q in chosen}
    else:
        chosen, why = pick(rec1, ids, order, cfg, a.positions, a.min_cluster, a.slow, not a.no_cap)
    print("pass 1: %.1f s, accepted %d, largest cluster %d, digest %s; dumping %d positions" %
          (dt1, int(rec1["ok"].sum()), int(rec1["cluster"].max()), dig1, len(chosen)), flush=True)
    rec2, dumps, dt2, dig2 = run(eng, order, dump_at=set(chosen), tag="pass 2")
    same = dig1 == dig2 and rec1.tobytes() == rec2.tobytes()
    print("pass 2: %.1f s, digest %s, records bitwise equal to pass 1: %s" % (dt2, dig2, same), flush=True)
    # the final consensus set is a fixed point of computeIndependentSubgraph's rule: every accepted candidate in order
    final = eng.getMaxConsensusSet()
    out = dict(workload=a.workload, desc=desc, order=order, records=rec1, passes_bitwise_equal=same, digest=dig1,
               seconds=np.array([dt1, dt2]), final_consensus=final, positions=np.array(chosen, dtype=np.int32),
               reasons=np.array([why[q] for q in chosen]),
               loop_ids_checksum=np.int64(ids.sum()), meas_checksum=float(np.asarray(g.loop_meas).sum()))
    cns_all, cns_off, win_all, win_off = [], [0], [], [0]
    wlo_all, whi_all = [], []
    stride = (len(order) - a.even_from) // max(a.even, 1) + 1
    for q in chosen:
        poses, cns = dumps[q]
        lo, hi = int(rec1["lo"][q]), int(rec1["hi"][q])
        if a.even > 0:
            # a re-synchronisation point of a forward stretch: every pose a check of the stretch can read -- from the first
            # vertex of any cluster the next candidates meet (pass 1 knows them) to the last vertex any accept has optimised
            lo = int(rec1["lo"][q:q + stride + 1].min())
            hi = int(max(rec1["hi"][:q + 1].max(), hi))
        wlo_all.append(lo); whi_all.append(hi)
        w = poses[lo:hi + 1]
        if g.dim == 3:
            w = np.concatenate([rot_to_quat(w[:, :9]), w[:, 9:12]], axis=1)
        cns_all.append(cns.astype(np.int32)); cns_off.append(cns_off[-1] + len(cns))
        win_all.append(w); win_off.append(win_off[-1] + w.shape[0])
    out.update(window_lo=np.array(wlo_all, dtype=np.int32), window_hi=np.array(whi_all, dtype=np.int32))
    out.update(cns=np.concatenate(cns_all) if cns_all else np.zeros(0, np.int32), cns_off=np.array(cns_off),
               window=np.concatenate(win_all) if win_all else np.zeros((0, 7)), window_off=np.array(win_off))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", "late_states_%s%s.npz" % (a.workload.lower(), a.suffix))
    np.savez_compressed(path, **out)
    print(json.dumps(dict(workload=a.workload, candidates=len(order), seconds=[round(dt1, 2), round(dt2, 2)],
                          accepted=int(rec1["ok"].sum()), largest_cluster=int(rec1["cluster"].max()), digest=dig1,
                          bitwise=bool(same), positions=[int(q) for q in chosen],
                          file=os.path.relpath(path, ROOT), bytes=os.path.getsize(path))), flush=True)
    # state injection (ipc_incremental_set_state) reproduces the run: a second engine, one check at a time, put into three
    # of the dumped states -- with the poses as downloaded bit for bit, with the fixture's quaternion form to rounding
    os.environ["IPC_SPEC_WINDOW"] = "1"
    e1 = IPC(g, cfg)
    del os.environ["IPC_SPEC_WINDOW"]
    open_loop = e1.initial_poses()
    for q in [chosen[0], chosen[len(chosen) // 2], chosen[-1]] if chosen else []:
        poses, cns = dumps[q]
        k, r = int(order[q]), rec1[q]
        e1.set_state(poses, cns, q)
        ok, info = e1.agreementCheck(k, with_info=True)
        exact = (ok, info.lo, info.hi, info.n_cluster_loops, info.iterations, info.tries) == (bool(r["ok"]), r["lo"], r["hi"], r["cluster"], r["iterations"], r["tries"]) \
            and np.float64(info.max_chi2).tobytes() == np.float64(r["max_chi2"]).tobytes()
        i = chosen.index(q)
        lo, hi = wlo_all[i], whi_all[i]
        inj = open_loop.copy()
        inj[lo:hi + 1] = window_to_poses(g.dim, win_all[i])
        e1.set_state(inj, cns, q)
        ok2, info2 = e1.agreementCheck(k, with_info=True)
        rel = abs(info2.max_chi2 - r["max_chi2"]) / max(abs(r["max_chi2"]), 1e-300)
        print("  injected q=%d: bitwise %s; from the quaternion window: decision %s, iterations %d vs %d, chi2 rel diff %.2e" %
              (q, exact, ok2 == bool(r["ok"]), info2.iterations, r["iterations"], rel), flush=True)
    e1.close()
    for q in chosen:
        r = rec1[q]
        print("  q=%d k=%d %s cluster=%d chain=%d..%d it=%d flags=%d chi2=%.6g  [%s]" %
              (q, int(order[q]), "accept" if r["ok"] else "reject", r["cluster"], r["lo"], r["hi"], r["iterations"], r["flags"],
               r["max_chi2"], why[q]))
    eng.close()


if __name__ == "__main__":
    main()
