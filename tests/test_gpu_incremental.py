"""GPU parity of the faithful incremental mode (IPC::agreementCheck, reference
src/consensus.cpp:43-75) and of the final map (reference src/simulation.cpp:50-65) against the
CPU oracle, through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL = 1e-5      # chi2 tolerance (relative) of BASELINE.json's north star


def _engine(g, **kw):
    from ipc_amd.consensus import IPC, Config
    cfg = Config(**kw)
    return IPC(g, cfg, device=0), cfg


def _oracle_inc(O, g, cfg):
    return O.IncrementalIPC(2, g.odom_meas, g.odom_info, cfg.s_factor, cfg.fast_reject_th,
                            cfg.fast_reject_iter_base, cfg.slow_reject_th, cfg.slow_reject_iter_base,
                            g.loop_ids, g.loop_meas, g.loop_info)


def _run_both(O, g, eng, cfg):
    inc = _oracle_inc(O, g, cfg)
    order = eng.candidate_order()
    eng.reset()
    worst = 0.0
    for k in order:
        ok_ref, ref = inc.agreement_check(k)
        ok, info = eng.agreementCheck(k, with_info=True)
        assert (info.lo, info.hi, info.n_cluster_loops) == (ref["lo"], ref["hi"], ref["cluster"]), (k, ref)
        assert ok == ok_ref, (k, ref, info.max_chi2)
        err = abs(info.max_chi2 - ref["max_chi2"]) / max(abs(ref["max_chi2"]), 1e-12)
        worst = max(worst, err)
        assert err <= REL, (k, ref, info.max_chi2)
        # (the iteration at which the dog-leg gives up near convergence is round-off driven, so
        # the counts are not compared)
    assert np.array_equal(eng.getMaxConsensusSet(), inc.consensus())
    ref_poses = inc.poses()
    got = eng.current_poses()
    assert np.allclose(got[:, :2], ref_poses[:, :2], rtol=0, atol=1e-6)
    dth = np.angle(np.exp(1j * (got[:, 2] - ref_poses[:, 2])))
    assert np.abs(dth).max() <= 1e-7
    return worst, inc


def test_incremental_small_step_by_step(oracle):
    from ipc_amd import synth
    g = synth.inject_outliers(synth.small_se2(), 6, seed=3)
    eng, cfg = _engine(g)
    _run_both(oracle, g, eng, cfg)


@pytest.mark.parametrize("seed", [5, 11])
def test_incremental_medium_clusters(oracle, seed):
    """3 laps: clusters absorb tens of accepted loops (capacitance systems up to ~60 unknowns and
    the x5 iteration rule)."""
    from ipc_amd import synth
    g = synth._se2_graph(400, 24, seed=70 + seed, laps=3.0, name="inc")
    g = synth.inject_outliers(g, 16, seed=seed)
    eng, cfg = _engine(g)
    _, inc = _run_both(oracle, g, eng, cfg)
    assert len(inc.consensus()) >= 10


def test_consensus_set_editing(oracle):
    """removeEdgeFromCnS / addEdgeToCnS (reference src/consensus.cpp:77-119)."""
    from ipc_amd import synth
    g = synth.small_se2()
    eng, cfg = _engine(g)
    order = eng.candidate_order()
    eng.reset()
    for k in order:
        eng.agreementCheck(k)
    cs = list(eng.getMaxConsensusSet())
    assert len(cs) >= 2
    k = cs[0]
    assert eng.removeEdgeFromCnS(k) is True
    assert list(eng.getMaxConsensusSet()) == cs[1:]
    assert eng.removeEdgeFromCnS(k) is False
    eng.addEdgeToCnS(k)
    after = list(eng.getMaxConsensusSet())
    assert sorted(after) == sorted(cs)
    hi = np.maximum(g.loop_ids[:, 0], g.loop_ids[:, 1])
    assert list(hi[after]) == sorted(hi[after])            # cmpEdgesTime order
    eng.addEdgeToCnS(k)                                     # same vertex pair: no duplicate
    assert list(eng.getMaxConsensusSet()) == after


def _final_reference(O, g, cfg, acc, iter_base):
    s = cfg.s_factor
    info = (np.asarray(g.odom_info) * s) / s                # setInformation(information()/s) after *s
    poses = O.propagate(2, g.odom_meas)
    order = O.candidate_order(g.loop_ids)
    sel = [k for k in order if acc[k]]
    return O.solve_cell(2, g.odom_meas, info, 1.0, poses, 0, g.V - 1, g.loop_ids[sel], g.loop_meas[sel],
                        g.loop_info[sel], iter_base, want_poses=True)


@pytest.mark.parametrize("which", ["small", "laps"])
def test_final_map_matches_oracle(oracle, which):
    from ipc_amd import synth
    if which == "small":
        g = synth.inject_outliers(synth.small_se2(), 6, seed=3)
    else:
        g = synth.inject_outliers(synth._se2_graph(400, 24, seed=81, laps=3.0, name="fin"), 10, seed=2)
    eng, cfg = _engine(g)
    _, acc = eng.run()
    assert acc.sum() >= 2
    n_edges = (g.V - 1) + int(acc.sum())
    iters = 1000
    # the oracle's cell solve applies the reference's x5 rule to iter_base when #edges > 100
    base = iters // 5 if n_edges > 100 else iters
    ref = _final_reference(oracle, g, cfg, acc, base)
    poses, info = eng.final_optimize(acc, iterations=iters)
    assert abs(info.chi2_total - ref["chi2_final"]) <= REL * max(ref["chi2_final"], 1e-12)
    assert abs(info.max_chi2 - ref["max_chi2"]) <= REL * max(ref["max_chi2"], 1e-12)
    assert np.allclose(poses[:, :2], ref["poses"][:, :2], rtol=0, atol=1e-6)
    dth = np.angle(np.exp(1j * (poses[:, 2] - ref["poses"][:, 2])))
    assert np.abs(dth).max() <= 1e-7


def test_final_map_without_loops_is_open_loop(oracle):
    from ipc_amd import synth
    g = synth.inject_outliers(synth.small_se2(), 4, seed=9)
    eng, _ = _engine(g)
    poses, info = eng.final_optimize(np.zeros(g.N, dtype=np.uint8))
    assert np.allclose(poses, eng.initial_poses(), rtol=0, atol=0)
    assert info.chi2_total == 0.0


# ---- SE3 ---------------------------------------------------------------------------------
def _rot_angle(Ra, Rb):
    """Angle of Ra^T Rb for [n, 9] row-major rotations."""
    A, B = Ra.reshape(-1, 3, 3), Rb.reshape(-1, 3, 3)
    tr = np.einsum("nij,nij->n", A, B)
    return np.arccos(np.clip((tr - 1.0) / 2.0, -1.0, 1.0))


def _run_both_se3(O, g, eng, cfg):
    inc = O.IncrementalIPC(3, g.odom_meas, g.odom_info, cfg.s_factor, cfg.fast_reject_th,
                           cfg.fast_reject_iter_base, cfg.slow_reject_th, cfg.slow_reject_iter_base,
                           g.loop_ids, g.loop_meas, g.loop_info)
    eng.reset()
    for k in eng.candidate_order():
        ok_ref, ref = inc.agreement_check(k)
        ok, info = eng.agreementCheck(k, with_info=True)
        assert (info.lo, info.hi, info.n_cluster_loops) == (ref["lo"], ref["hi"], ref["cluster"]), (k, ref)
        assert ok == ok_ref, (k, ref, info.max_chi2)
        err = abs(info.max_chi2 - ref["max_chi2"]) / max(abs(ref["max_chi2"]), 1e-12)
        assert err <= REL, (k, ref, info.max_chi2)
    assert np.array_equal(eng.getMaxConsensusSet(), inc.consensus())
    ref_poses, got = inc.poses(), eng.current_poses()
    assert np.allclose(got[:, 9:], ref_poses[:, 9:], rtol=0, atol=1e-6)
    assert _rot_angle(got[:, :9], ref_poses[:, :9]).max() <= 1e-6
    return inc


def test_incremental_se3_small_step_by_step(oracle):
    from ipc_amd import synth
    g = synth.inject_outliers(synth.small_se3(), 5, seed=4)
    eng, cfg = _engine(g, s_factor=50.0, slow_reject_th=6.251)
    _run_both_se3(oracle, g, eng, cfg)


def test_incremental_se3_clusters(oracle):
    """A small sphere: clusters absorb several accepted loops (12 x 12 and larger capacitance systems)."""
    from ipc_amd import synth
    g = synth.sphere_like(rings=8, per_ring=16, radius=8.0)
    keep = np.arange(0, g.N, max(1, g.N // 24))
    g = synth.inject_outliers(g.subset(keep), 8, seed=6)
    eng, cfg = _engine(g, s_factor=50.0, slow_reject_th=6.251)
    inc = _run_both_se3(oracle, g, eng, cfg)
    assert len(inc.consensus()) >= 5


def test_final_map_se3_matches_oracle(oracle):
    from ipc_amd import synth
    O = oracle
    g = synth.sphere_like(rings=8, per_ring=16, radius=8.0)
    keep = np.arange(0, g.N, max(1, g.N // 24))
    g = synth.inject_outliers(g.subset(keep), 6, seed=3)
    eng, cfg = _engine(g, s_factor=50.0, slow_reject_th=6.251)
    _, acc = eng.run()
    assert acc.sum() >= 3
    s = cfg.s_factor
    info = (np.asarray(g.odom_info) * s) / s
    poses0 = O.propagate(3, g.odom_meas)
    sel = [k for k in O.candidate_order(g.loop_ids) if acc[k]]
    n_edges = (g.V - 1) + len(sel)
    iters = 1000
    base = iters // 5 if n_edges > 100 else iters
    ref = O.solve_cell(3, g.odom_meas, info, 1.0, poses0, 0, g.V - 1, g.loop_ids[sel], g.loop_meas[sel],
                       g.loop_info[sel], base, want_poses=True)
    poses, inf = eng.final_optimize(acc, iterations=iters)
    assert abs(inf.chi2_total - ref["chi2_final"]) <= REL * max(ref["chi2_final"], 1e-12)
    assert abs(inf.max_chi2 - ref["max_chi2"]) <= REL * max(ref["max_chi2"], 1e-12)
    assert np.allclose(poses[:, 9:], ref["poses"][:, 9:], rtol=0, atol=1e-6)
    assert _rot_angle(poses[:, :9], ref["poses"][:, :9]).max() <= 1e-6


@pytest.mark.parametrize("workload", ["C5", "T2400"])
def test_tail_after_accepts_matches_the_serial_compose_on_long_chains(workload):
    """k_apply_accept rebuilds the tail behind an accepted window as ONE rigid transform of the open-loop poses
    (D = T (+) P^-1, round 4) where the reference composes pose by pose (propagateCurrentGuess, src/consensus_utils.cpp:61-71).
    On long tails (V = 50 000 SE3, V = 2 400 SE2) after several accepts: equal to the oracle's serial compose to 1e-12 of
    the trajectory's extent, and the SE3 rotations stay orthonormal (two unnormalised 3 x 3 products per accept)."""
    import bench
    from ipc_amd.consensus import IPC
    from oracle import oracle as O
    g, cfg, _ = bench.build_workload(workload)
    eng = IPC(g, cfg, device=0)
    inc = O.IncrementalIPC(g.dim, g.odom_meas, g.odom_info, cfg.s_factor, cfg.fast_reject_th, cfg.fast_reject_iter_base,
                           cfg.slow_reject_th, cfg.slow_reject_iter_base, g.loop_ids, g.loop_meas, g.loop_info)
    order = eng.candidate_order()
    eng.reset()
    accepts = 0
    for k in order[:400]:
        ok = eng.agreementCheck(int(k))
        ok_ref, _ = inc.agreement_check(int(k))
        assert ok == ok_ref
        accepts += ok
        if accepts >= 6:
            break
    assert accepts >= 3
    # behind the last accepted window the trajectory is pure odometry: pose i = pose i-1 (+) z_{i-1}, whatever the window's
    # own poses are (those agree with the oracle's to solver tolerance only)
    got = eng.current_poses()
    hi_last = int(max(g.loop_ids[int(c)].max() for c in eng.getMaxConsensusSet()))
    idx = np.arange(hi_last + 1, g.V)
    assert len(idx) > 1000
    if g.dim == 3:
        from ipc_amd.synth import _quat_to_R
        R = got[:, :9].reshape(-1, 3, 3)
        t = got[:, 9:]
        extent = np.abs(t).max()
        Rz = np.stack([_quat_to_R(q / np.linalg.norm(q)) for q in g.odom_meas[idx - 1, 3:]])
        t_pred = t[idx - 1] + np.einsum("nij,nj->ni", R[idx - 1], g.odom_meas[idx - 1, :3])
        R_pred = np.einsum("nij,njk->nik", R[idx - 1], Rz)
        assert np.abs(t[idx] - t_pred).max() <= 2e-12 * max(extent, 1.0), np.abs(t[idx] - t_pred).max()
        assert np.abs(R[idx] - R_pred).max() <= 1e-12
        assert np.abs(np.einsum("nij,nkj->nik", R, R) - np.eye(3)).max() <= 1e-12
    else:
        x, y, th = got[:, 0], got[:, 1], got[:, 2]
        extent = max(np.abs(x).max(), np.abs(y).max())
        z = g.odom_meas[idx - 1]
        c, s_ = np.cos(th[idx - 1]), np.sin(th[idx - 1])
        assert np.abs(x[idx] - (x[idx - 1] + c * z[:, 0] - s_ * z[:, 1])).max() <= 2e-12 * max(extent, 1.0)
        assert np.abs(y[idx] - (y[idx - 1] + s_ * z[:, 0] + c * z[:, 1])).max() <= 2e-12 * max(extent, 1.0)
        assert np.abs(np.angle(np.exp(1j * (th[idx] - th[idx - 1] - z[:, 2])))).max() <= 1e-12
    eng.close()


# ---- resume from a saved state (ipc_incremental_set_state, round 6) -----------------------------------------------------
@pytest.mark.parametrize("workload", ["C1", "C4s"])
def test_a_run_resumed_from_a_saved_state_continues_as_the_uninterrupted_run(workload):
    """The state of the reference's IPC object is the vertex estimates + _max_consensus_set (include/ipc/consensus.hpp:23-32):
    (ipc_current_poses, ipc_consensus_set) taken in the middle of a run and handed to ANOTHER engine continue that run --
    SE3 bit for bit (the rotation matrices are the state), SE2 to rounding (cos / sin of theta are recomputed).  Bad
    arguments are refused."""
    import bench
    from ipc_amd import capi
    from ipc_amd.consensus import IPC
    g, cfg, _ = bench.build_workload(workload)
    a, b = IPC(g, cfg, device=0), IPC(g, cfg, device=0)
    order = a.candidate_order()
    cut = len(order) // 2
    a.reset()
    ra = []
    for q, k in enumerate(order):
        if q == cut:
            b.set_state(a.current_poses(), a.getMaxConsensusSet(), cut)
            assert np.array_equal(b.getMaxConsensusSet(), a.getMaxConsensusSet())
        ok, info = a.agreementCheck(int(k), with_info=True)
        ra.append((ok, info.lo, info.hi, info.n_cluster_loops, info.max_chi2))
    assert sum(r[0] for r in ra[:cut]) >= 5
    for q in range(cut, len(order)):
        ok, info = b.agreementCheck(int(order[q]), with_info=True)
        r = ra[q]
        assert (ok, info.lo, info.hi, info.n_cluster_loops) == r[:4], q
        if g.dim == 3:
            assert np.float64(info.max_chi2).tobytes() == np.float64(r[4]).tobytes(), (q, info.max_chi2, r[4])
        else:
            assert abs(info.max_chi2 - r[4]) <= 1e-6 * max(abs(r[4]), 1e-12), (q, info.max_chi2, r[4])    # (the dog-leg's last steps are rounding-driven)
    assert np.array_equal(a.getMaxConsensusSet(), b.getMaxConsensusSet())
    pa, pb = a.current_poses(), b.current_poses()
    assert np.abs(pa - pb).max() <= (0.0 if g.dim == 3 else 1e-7)
    with pytest.raises(capi.IpcError):
        b.set_state(pa, np.array([g.N + 3], dtype=np.int32), 0)            # not a candidate
    with pytest.raises(capi.IpcError):
        b.set_state(pa, a.getMaxConsensusSet(), g.N + 1)                    # resume position beyond the list
    a.close(); b.close()
