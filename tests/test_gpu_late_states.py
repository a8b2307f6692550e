"""The WHOLE faithful runs of BASELINE configs[3] (C4) and configs[4] (C5) under the driver, and their late part under the
oracle.

The reference loops IPC::agreementCheck over every candidate (src/simulation.cpp:34-47 -> src/consensus.cpp:43-75).  The
CPU oracle's own run reaches candidate 1 205 of C4's 4 450 and 3 239 of C5's 25 000 in 45 min each
(tests/golden/c{4,5}_incremental_expected.npz); behind that the clusters grow to 1 992 / 3 328 accepted loops -- the
banded, split solver of cluster_band.hpp -- and until round 6 nothing but the engine itself had looked at those checks.

  * full runs: every candidate through the look-ahead pipeline; decisions, cluster spans and sizes equal to the committed
    records of the run the late states were taken from, accepted counts, the final consensus set; a late window of the run
    repeated one check at a time (IPC_SPEC_WINDOW=1) from the pipeline run's own state: bit for bit the same records;
    the consensus set a fixed point of computeIndependentSubgraph's rule (src/consensus.cpp:124-171) in both of its forms;
  * STATE-INJECTED ORACLE CHECKS: the state of the reference's IPC object is the vertex estimates and _max_consensus_set
    (include/ipc/consensus.hpp:23-32).  tools/late_state_dump.py took that pair in front of late checks of the GPU runs;
    tests/golden/make_late_state_golden.py put the oracle into each state and let it check the candidate (minutes to hours
    per position in the build container); here the engine is put into the same states (ipc_incremental_set_state) and must
    take the oracle's decision, find the oracle's cluster (lo, hi, size) and end on its max edge chi2 within 1e-5.
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL = 1e-5
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _engine(g, cfg, **env):
    from ipc_amd.consensus import IPC
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return IPC(g, cfg, device=0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def _window_to_poses(dim, w):
    """tests/golden/make_late_state_golden.py::window_to_poses (the oracle was given exactly these numbers)."""
    if dim == 2:
        return w
    q = w[:, :4] / np.linalg.norm(w[:, :4], axis=1, keepdims=True)
    qw, qx, qy, qz = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.stack([1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw),
                  2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw),
                  2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)], axis=1)
    return np.concatenate([R, w[:, 4:7]], axis=1)


def _load(tag):
    import bench
    fx = np.load(os.path.join(GOLD, "%s_late_states.npz" % tag))
    g, cfg, _ = bench.build_workload(str(fx["workload"]))
    assert int(np.asarray(g.loop_ids, dtype=np.int64).sum()) == int(fx["loop_ids_checksum"]), "workload changed"
    assert abs(float(np.asarray(g.loop_meas).sum()) - float(fx["meas_checksum"])) < 1e-9, "workload changed"
    return fx, g, cfg


def _record(ok, info):
    return (ok, info.lo, info.hi, info.n_cluster_loops, info.iterations, info.tries, info.flags, info.max_chi2, info.chi2_total,
            info.chi2_initial)


def _bitwise(a, b):
    return a[:7] == b[:7] and all(np.float64(x).tobytes() == np.float64(y).tobytes() or (x != x and y != y) for x, y in zip(a[7:], b[7:]))


def _late_states(tag, min_positions, min_cluster):
    fx, g, cfg = _load(tag)
    n = len(fx["position"])
    assert n >= min_positions
    assert int(fx["cluster"].max()) >= min_cluster
    assert fx["decision"].any() and not fx["decision"].all()                # accepts AND rejects
    eng = _engine(g, cfg, IPC_SPEC_WINDOW=1)
    order = eng.candidate_order()
    open_loop = eng.initial_poses()
    worst, differ = 0.0, []
    for i in range(n):
        q, k = int(fx["position"][i]), int(fx["candidate"][i])
        assert int(order[q]) == k
        lo, hi = int(fx["lo"][i]), int(fx["hi"][i])
        poses = open_loop.copy()
        poses[lo:hi + 1] = _window_to_poses(g.dim, fx["window"][fx["window_off"][i]:fx["window_off"][i + 1]])
        eng.set_state(poses, fx["cns"][fx["cns_off"][i]:fx["cns_off"][i + 1]], q)
        ok, info = eng.agreementCheck(k, with_info=True)
        assert (info.lo, info.hi, info.n_cluster_loops) == (lo, hi, int(fx["cluster"][i])), (i, q, k)
        ref = float(fx["max_chi2"][i])
        err = abs(info.max_chi2 - ref) / max(abs(ref), 1e-12)
        worst = max(worst, err)
        if ok != bool(fx["decision"][i]) or err > REL:
            differ.append((i, q, k, ok, bool(fx["decision"][i]), info.max_chi2, ref, info.iterations, int(fx["iterations"][i]), str(fx["reason"][i])))
    assert not differ, differ
    print("\n[%s late states] %d positions (clusters of %d ... %d loops, %d accepts / %d rejects), worst relative chi2 difference "
          "from the oracle %.2e" % (tag, n, int(fx["cluster"].min()), int(fx["cluster"].max()), int(fx["decision"].sum()),
                                    int(n - fx["decision"].sum()), worst))
    eng.close()


def test_c4_late_states_against_the_oracle():
    """BASELINE configs[3]: >= 16 late positions, clusters of >= 1 000 loops, the largest cluster of the run included."""
    _late_states("c4", 16, 1900)


def test_c5_late_states_against_the_oracle():
    """BASELINE configs[4]: >= 8 late positions (chains of up to 30 000 poses = 180 000 unknowns in the oracle's system)."""
    _late_states("c5", 8, 3000)


def _full_run(tag, window_at, window_len):
    from ipc_amd import capi
    fx, g, cfg = _load(tag)
    eng = _engine(g, cfg)
    order = eng.candidate_order()
    n = len(order)
    assert n == len(fx["run_decision"])
    eng.reset()
    recs = []
    saved = None
    for q in range(n):
        if q == window_at:
            saved = (eng.current_poses(), eng.getMaxConsensusSet().copy())
        ok, info = eng.agreementCheck(int(order[q]), with_info=True)
        recs.append(_record(ok, info))
    dec = np.array([r[0] for r in recs], dtype=np.uint8)
    clu = np.array([r[3] for r in recs])
    # the run the oracle-checked states were taken from: same decisions, same clusters, same chi2 to rounding
    assert np.array_equal(dec, fx["run_decision"]), np.nonzero(dec != fx["run_decision"])[0][:10]
    assert np.array_equal(clu, fx["run_cluster"])
    assert np.array_equal(np.array([r[1] for r in recs]), fx["run_lo"]) and np.array_equal(np.array([r[2] for r in recs]), fx["run_hi"])
    mx = np.array([r[7] for r in recs])
    both = np.isfinite(mx) & np.isfinite(fx["run_max_chi2"])
    assert np.abs(mx[both] - fx["run_max_chi2"][both]).max() <= 1e-6 * np.maximum(np.abs(fx["run_max_chi2"][both]), 1e-9).max()
    cns = eng.getMaxConsensusSet()
    assert np.array_equal(cns, fx["run_final_consensus"])
    assert np.array_equal(cns, order[dec[np.arange(n)] == 1])               # the accepted candidates, in processing order
    # the oracle's OWN run of the first candidates (tests/golden/<tag>_incremental_expected.npz: as far as it got in 45 min)
    pre = np.load(os.path.join(GOLD, "%s_incremental_expected.npz" % tag))
    npre = len(pre["order"])
    assert np.array_equal(order[:npre], pre["order"])
    worst_pre = 0.0
    for q in range(npre):
        r = recs[q]
        assert r[0] == bool(pre["decision"][q]) and (r[1], r[2], r[3]) == (int(pre["lo"][q]), int(pre["hi"][q]), int(pre["cluster"][q])), q
        ref = float(pre["max_chi2"][q])
        err = abs(r[7] - ref) / max(abs(ref), 1e-12)
        worst_pre = max(worst_pre, err)
        assert err <= REL, (q, r[7], ref)
    print("\n[%s full run] the oracle's own prefix run: %d candidates, worst relative chi2 difference %.2e" % (tag, npre, worst_pre))
    # at the oracle-checked positions the live run agrees with the ORACLE as well
    for i, q in enumerate(fx["position"]):
        r = recs[int(q)]
        assert r[0] == bool(fx["decision"][i]) and (r[1], r[2], r[3]) == (int(fx["lo"][i]), int(fx["hi"][i]), int(fx["cluster"][i]))
        assert abs(r[7] - float(fx["max_chi2"][i])) <= REL * max(abs(float(fx["max_chi2"][i])), 1e-12)
    # ... and at every position of the oracle's FORWARD STRETCHES: from each of those states (and from the end of its own prefix
    # run) the oracle went on by itself, its state evolving by its own accepts (make_late_state_golden.py --forward)
    if "fwd_position" in fx.files and len(fx["fwd_position"]):
        worst = 0.0
        for i, q in enumerate(fx["fwd_position"]):
            r = recs[int(q)]
            assert r[0] == bool(fx["fwd_decision"][i]), (int(q), r, float(fx["fwd_max_chi2"][i]))
            assert (r[1], r[2], r[3]) == (int(fx["fwd_lo"][i]), int(fx["fwd_hi"][i]), int(fx["fwd_cluster"][i])), int(q)
            ref = float(fx["fwd_max_chi2"][i])
            if ref == ref:
                err = abs(r[7] - ref) / max(abs(ref), 1e-12)
                worst = max(worst, err)
                assert err <= REL, (int(q), r[7], ref)
        print("\n[%s full run] %d of %d positions checked against the oracle's forward stretches, worst relative chi2 difference %.2e"
              % (tag, len(fx["fwd_position"]), n, worst))
    # the consensus set is a fixed point of computeIndependentSubgraph's rule, in the reference's re-scan form and in the
    # engine's one-sweep form: for a handful of members, both find the same cluster inside the final set
    lib = capi.load()
    ids = np.asarray(g.loop_ids, dtype=np.int32).reshape(-1, 2)
    lo_all = np.ascontiguousarray(ids[cns].min(axis=1).astype(np.int32))
    hi_all = np.ascontiguousarray(ids[cns].max(axis=1).astype(np.int32))
    for m in np.linspace(0, len(cns) - 1, 6).astype(int):
        outs = []
        for sweep in (0, 1):
            mem = np.zeros(len(cns), dtype=np.int32)
            nm, olo, ohi = C.c_int(0), C.c_int(0), C.c_int(0)
            capi.check(lib.ipc_debug_absorbed_edges(int(lo_all[m]), int(hi_all[m]), len(cns), lo_all.ctypes.data_as(C.c_void_p),
                                                    hi_all.ctypes.data_as(C.c_void_p), sweep, mem.ctypes.data_as(C.c_void_p),
                                                    C.byref(nm), C.byref(olo), C.byref(ohi)))
            outs.append((set(mem[:nm.value].tolist()), olo.value, ohi.value))
        assert outs[0] == outs[1] and m in outs[0][0]
    # a late window again, one check at a time, from the pipeline run's own state: bit for bit the pipeline's records
    e1 = _engine(g, cfg, IPC_SPEC_WINDOW=1)
    e1.set_state(saved[0], saved[1], window_at)
    for q in range(window_at, window_at + window_len):
        ok, info = e1.agreementCheck(int(order[q]), with_info=True)
        assert _bitwise(_record(ok, info), recs[q]), (q, _record(ok, info), recs[q])
    e1.close()
    eng.close()
    return int(dec.sum()), int(clu.max())


def test_c4_full_faithful_run():
    """All 4 450 candidates of BASELINE configs[3] (58 s in round 5)."""
    acc, big = _full_run("c4", 3900, 48)
    assert big >= 1900 and acc >= 1900
    print("\n[C4 full run] %d accepted, largest cluster %d loops" % (acc, big))


def test_c5_full_faithful_run():
    """All 25 000 candidates of BASELINE configs[4] (144 s in round 5)."""
    acc, big = _full_run("c5", 23000, 120)
    assert big >= 3000 and acc >= 4500
    print("\n[C5 full run] %d accepted, largest cluster %d loops" % (acc, big))


# ---- Levenberg retry and lost launches at band sizes ---------------------------------------------------
def _rank_deficient(info_row):
    """EDGE_SE3 information (21 upper-triangular values, x y z qx qy qz order) with the rotation rows and columns zeroed:
    rank 3, no inverse -- the capacitance formulation of the cluster solvers (which needs the covariance) cannot take it."""
    out = np.zeros(21)
    keep = [0, 1, 2, 6, 7, 11]                                   # (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
    out[keep] = info_row[keep]
    return out


@pytest.mark.parametrize("tag,min_cluster,min_unknowns", [("c4", 1500, 10000), ("c5", 2500, 100000)])
def test_levenberg_retry_inside_a_large_cluster_against_the_oracle(oracle, tag, min_cluster, min_unknowns):
    """g2o's "dl_var" (reference src/utils.cpp:104-105) retries a failed factorisation with lambda on the diagonal at ANY
    size.  Rounds 3 - 5 could only do that on a dense copy of the normal equations, up to 24 000 pose unknowns: in C5's
    clusters (10 000 - 34 000 poses) such a check ended in Fail = reject, in C4's it took a 1.8 GB dense factorisation per
    retry.  A late state of the committed runs (cluster of >= 1 500 / 2 500 accepted loops), the candidate's information
    made rank deficient (no rotation part): the engine must take the oracle's decision and end on its chi2 -- through the
    banded literal system (cluster_literal_band.hpp)."""
    import bench
    O = oracle
    fx, g, cfg = _load(tag)
    pick = [i for i in range(len(fx["position"])) if fx["decision"][i] and fx["cluster"][i] >= min_cluster]
    assert pick
    i = pick[0]
    q, k = int(fx["position"][i]), int(fx["candidate"][i])
    lo, hi = int(fx["lo"][i]), int(fx["hi"][i])
    assert 6 * (hi - lo) >= min_unknowns
    li = np.array(g.loop_info, dtype=np.float64, copy=True)
    li[k] = _rank_deficient(li[k])
    g.loop_info = li
    cns = fx["cns"][fx["cns_off"][i]:fx["cns_off"][i + 1]]
    window = _window_to_poses(g.dim, fx["window"][fx["window_off"][i]:fx["window_off"][i + 1]])
    # the oracle: g2o's literal normal equations, no trouble with a singular information matrix
    inc = O.IncrementalIPC(g.dim, g.odom_meas, g.odom_info, cfg.s_factor, cfg.fast_reject_th, cfg.fast_reject_iter_base,
                           cfg.slow_reject_th, cfg.slow_reject_iter_base, g.loop_ids, g.loop_meas, g.loop_info)
    poses = inc.poses()
    poses[lo:hi + 1] = window
    inc.set_state(poses, cns)
    try:
        O.set_wide_dots(True)
        ok_ref, ref = inc.agreement_check(k)
    finally:
        O.set_wide_dots(False)
    eng = _engine(g, cfg, IPC_SPEC_WINDOW=1)
    gp = eng.initial_poses()
    gp[lo:hi + 1] = window
    eng.set_state(gp, cns, q)
    ok, info = eng.agreementCheck(k, with_info=True)
    c = eng.incremental_counters()
    assert (info.lo, info.hi, info.n_cluster_loops) == (ref["lo"], ref["hi"], ref["cluster"])
    assert not (info.flags & 2), "the optimisation ended in Fail"
    assert info.flags & 4, "no damped solve"
    assert c["host_solver_fallbacks"] >= 1 and c["literal_band_solves"] >= 1, c
    assert ok == ok_ref, (info.max_chi2, ref)
    assert abs(info.max_chi2 - ref["max_chi2"]) <= REL * max(abs(ref["max_chi2"]), 1e-12), (info.max_chi2, ref)
    print("\n[%s, rank-deficient candidate in a cluster of %d loops, %d pose unknowns] decision %s, chi2 %.9g (oracle %.9g), %d iterations "
          "(oracle %d), %d banded literal solves" % (tag, info.n_cluster_loops, 6 * (hi - lo), ok, info.max_chi2, ref["max_chi2"],
                                                     info.iterations, ref["iterations"], c["literal_band_solves"]))
    eng.close()


def test_lost_launches_are_launched_again():
    """The persistent solver kernels meet at grid barriers, so every workgroup of a launch must be resident; beside a foreign
    tenant's kernels one may never be, the barrier gives up and the launch is LOST (PersistOut::error 1).  Round 5 sent such
    a check to the host-driven dense solver.  Now it is launched again -- alone, when its turn comes.  Every third solve of
    every solver instance reported lost (IPC_PERSIST_FAULT_EVERY=3: the host side of the time-out path; the device side is a
    3 s spin nobody wants in a test suite), the small sphere through the pipeline and one check at a time, through the dense
    kernel and (IPC_BAND_MIN_N=0) the banded one: the records of the undisturbed run, bit for bit, and no check ever reaches
    the host-driven solver."""
    import bench
    g, cfg, _ = bench.build_workload("C4s")
    for band in ({}, dict(IPC_BAND_MIN_N=0)):
        ref = _engine(g, cfg, **band)
        order = ref.candidate_order()

        def records(eng):
            eng.reset()
            out = []
            for k in order:
                ok, info = eng.agreementCheck(int(k), with_info=True)
                out.append(_record(ok, info))
            return out

        r0 = records(ref)
        for env in (dict(IPC_PERSIST_FAULT_EVERY=3), dict(IPC_PERSIST_FAULT_EVERY=3, IPC_SPEC_WINDOW=1)):
            eng = _engine(g, cfg, **env, **band)
            r1 = records(eng)
            c = eng.incremental_counters()
            assert c["lost_launches"] >= 10, c
            assert c["host_solver_fallbacks"] == 0, c
            if "IPC_SPEC_WINDOW" in env:
                assert c["relaunches"] == c["lost_launches"], c
            for q, (a, b) in enumerate(zip(r0, r1)):
                assert _bitwise(a, b), (band, env, q, a, b)
            assert np.array_equal(eng.current_poses().view(np.uint64), ref.current_poses().view(np.uint64))
            eng.close()
        ref.close()
