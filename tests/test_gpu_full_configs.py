"""GPU parity at the full size of every BASELINE config (C1..C5): size-independent properties of the
matrix / set-max, and the HIP path against the CPU oracle on samples that are AIMED at the fragile
cells -- every sampled set contains the cells that ran to the iteration cap, the cells whose
capacitance solve failed (flags & 2) and the cells whose max chi2 lies within 1 % of the threshold,
next to a chain-length-stratified random part.  The oracle sweep runs on all host cores
(oracle_pair_cells_mt)."""
import os
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _properties(O, g, eng, bits, acc):
    """Symmetry, the non-overlap rule, set-max == greedy clique on the GPU's own matrix, clique-ness."""
    from ipc_amd.consensus import unpack_bits
    C = unpack_bits(bits, eng.N)
    assert np.array_equal(C, C.T)
    lo, hi = g.loop_ids.min(1), g.loop_ids.max(1)
    ov = (np.minimum(hi[:, None], hi[None, :]) - np.maximum(lo[:, None], lo[None, :])) > 0
    d = np.diag(C).astype(bool)
    free = ~ov & ~np.eye(eng.N, dtype=bool)
    assert np.array_equal(C[free].astype(bool), (d[:, None] & d[None, :])[free])   # consensus.cpp:157-159 rule
    order = O.candidate_order(g.loop_ids)
    assert np.array_equal(acc, O.set_max(C, order))
    A = np.nonzero(acc)[0]
    assert np.all(C[np.ix_(A, A)] == 1)
    return C


def _fragile(cells, cfg, cap_fast, cap_slow):
    th = np.where(cells["i"] == cells["j"], cfg.fast_reject_th, cfg.slow_reject_th)
    L = cells["hi"] - cells["lo"]
    nl = np.where(cells["i"] == cells["j"], 1, 2)
    cap = np.where(cells["i"] == cells["j"], cap_fast, cap_slow) * np.where(L + nl > 100, 5, 1)
    at_cap = cells["iterations"] >= cap
    failed = (cells["flags"] & 2) != 0
    near = np.abs(cells["max_chi2"] - th) <= 1e-2 * th
    return th, at_cap, failed, near


def _sample(cells, cfg, n_random, max_cap, seed=0, max_near=400):
    th, at_cap, failed, near = _fragile(cells, cfg, cfg.fast_reject_iter_base, cfg.slow_reject_iter_base)
    rng = np.random.default_rng(seed)
    L = cells["hi"] - cells["lo"]
    order = np.argsort(L, kind="stable")
    strat = order[np.linspace(0, len(order) - 1, min(n_random, len(order))).astype(np.int64)]
    capi = np.nonzero(at_cap)[0]
    if len(capi) > max_cap:                                      # the cheapest ones when there are too many
        capi = capi[np.argsort(L[capi], kind="stable")[:max_cap]]
    neari = np.nonzero(near)[0]
    if len(neari) > max_near:
        neari = rng.choice(neari, max_near, replace=False)
    pick = np.unique(np.concatenate([strat, capi, np.nonzero(failed)[0], neari]))
    rng.shuffle(pick)
    return pick, dict(cap=int(at_cap.sum()), cap_sampled=int(len(capi)), failed=int(failed.sum()),
                      near=int(near.sum()), near_sampled=int(len(neari)), random=int(len(strat)))


def _oracle_compare(O, g, cfg, cells, pick, rel=1e-5):
    """Decisions must agree on every sampled cell; max chi2 within `rel` wherever both sides converged
    (a cell that ran to the iteration cap stops at a rounding-dependent point of a still-moving
    trajectory: there the decision and a loose 1e-2 bound are checked)."""
    poses = O.propagate(g.dim, g.odom_meas)
    t0 = time.perf_counter()
    mx, its, used = O.pair_cells_mt(g.dim, g.odom_meas, g.odom_info, cfg.s_factor, poses, g.loop_ids, g.loop_meas,
                                    g.loop_info, cells["i"][pick], cells["j"][pick], cfg.fast_reject_iter_base,
                                    cfg.slow_reject_iter_base, os.cpu_count() or 1)
    dt = time.perf_counter() - t0
    c = cells[pick]
    th = np.where(c["i"] == c["j"], cfg.fast_reject_th, cfg.slow_reject_th)
    assert not np.isnan(mx).any()
    dec_o, dec_g = ~(mx > th), ~(c["max_chi2"] > th)
    bad = np.nonzero(dec_o != dec_g)[0]
    assert len(bad) == 0, [(int(c["i"][k]), int(c["j"][k]), float(mx[k]), float(c["max_chi2"][k])) for k in bad[:8]]
    L = c["hi"] - c["lo"]
    nl = np.where(c["i"] == c["j"], 1, 2)
    cap = np.where(c["i"] == c["j"], cfg.fast_reject_iter_base, cfg.slow_reject_iter_base) * np.where(L + nl > 100, 5, 1)
    conv = (its < cap) & (c["iterations"] < cap)
    err = np.abs(mx - c["max_chi2"]) / np.maximum(np.abs(mx), 1e-12)
    worst = float(err[conv].max()) if conv.any() else 0.0
    assert worst <= rel, (worst, [(int(c["i"][k]), int(c["j"][k]), float(mx[k]), float(c["max_chi2"][k]))
                                  for k in np.nonzero(conv & (err > rel))[0][:8]])
    if (~conv).any():
        assert float(err[~conv].max()) <= 1e-2
    print("oracle sweep: %d cells on %d threads in %.1f s; worst rel chi2 diff %.2e (converged), %d at the cap" % (
        len(pick), used, dt, worst, int((~conv).sum())))
    return worst


def _run(name):
    from bench import build_workload
    from ipc_amd.consensus import IPC
    g, cfg, _ = build_workload(name)
    eng = IPC(g, cfg, device=0)
    bits, acc = eng.run()
    return g, cfg, eng, bits, acc


def test_full_size_c1_whole_matrix_against_the_oracle(oracle):
    """C1 (INTEL-like + 100 outliers): every solved cell against the oracle when the host has the cores for
    it (45 k cells at ~50 cells/s per core), a 6000-cell aimed sample otherwise."""
    g, cfg, eng, bits, acc = _run("C1")
    _properties(oracle, g, eng, bits, acc)
    cells = eng.cell_info()
    cores = os.cpu_count() or 1
    if cores >= 48:
        pick = np.arange(len(cells))
        np.random.default_rng(0).shuffle(pick)
    else:
        pick, _ = _sample(cells, cfg, n_random=min(6000, 120 * cores), max_cap=8 * cores)
    _oracle_compare(oracle, g, cfg, cells, pick)


def test_full_size_c2_properties_and_aimed_parity(oracle):
    """C2, the headline config (INTEL-like + 1000 outliers, 789 396 cells)."""
    g, cfg, eng, bits, acc = _run("C2")
    _properties(oracle, g, eng, bits, acc)
    bits2, acc2 = eng.run()
    assert np.array_equal(bits, bits2) and np.array_equal(acc, acc2)          # run-to-run determinism
    cells = eng.cell_info()
    cores = os.cpu_count() or 1
    pick, info = _sample(cells, cfg, n_random=max(200, 40 * cores), max_cap=4 * cores)
    print("C2 sample:", info)
    assert info["cap"] > 0 and len(pick) >= 200
    _oracle_compare(oracle, g, cfg, cells, pick)


def test_full_size_c3_properties_and_aimed_parity(oracle):
    """C3 (MIT-like + 5000 outliers, 12.6 M cells)."""
    g, cfg, eng, bits, acc = _run("C3")
    _properties(oracle, g, eng, bits, acc)
    cells = eng.cell_info()
    cores = os.cpu_count() or 1
    pick, info = _sample(cells, cfg, n_random=max(200, 30 * cores), max_cap=2 * cores, max_near=40 * cores)
    print("C3 sample:", info)
    _oracle_compare(oracle, g, cfg, cells, pick)


def test_full_size_c4_properties_and_aimed_parity(oracle):
    """C4 (sphere2500-like SE3 + 2000 outliers, 9.9 M cells of which 3.2 M are solved, chains up to 2499
    poses): the LDS-pose kernels at their full capacity."""
    g, cfg, eng, bits, acc = _run("C4")
    _properties(oracle, g, eng, bits, acc)
    cells = eng.cell_info()
    assert (cells["hi"] - cells["lo"]).max() > 2304               # the W = 4, M = 10 variant is exercised
    assert int(((cells["flags"] & 2) != 0).sum()) == 0
    cores = os.cpu_count() or 1
    pick, info = _sample(cells, cfg, n_random=max(64, 3 * cores), max_cap=max(2, cores // 8), max_near=2 * cores)
    print("C4 sample:", info)
    _oracle_compare(oracle, g, cfg, cells, pick)


def test_full_size_c5_properties_and_aimed_parity(oracle):
    """C5 (50 000-pose SE3 chain, 5000 true loops + 20 000 local outliers, 312.5 M cells) on one GPU."""
    g, cfg, eng, bits, acc = _run("C5")
    from ipc_amd.consensus import unpack_bits
    N = eng.N
    # the dense N x N byte matrix would be 625 MB: check the properties on the bit rows directly
    b = np.ascontiguousarray(bits)
    rng = np.random.default_rng(5)
    rows = rng.choice(N, 512, replace=False)
    sub = unpack_bits(b[rows], N)                                # [512, N]
    cols = np.stack([((b[:, r >> 6] >> np.uint64(r & 63)) & np.uint64(1)).astype(np.uint8) for r in rows])
    assert np.array_equal(sub, cols)                              # symmetry on 512 rows / columns
    lo, hi = g.loop_ids.min(1), g.loop_ids.max(1)
    kk = np.arange(N)
    d = ((b[kk, kk >> 6] >> (kk & 63).astype(np.uint64)) & np.uint64(1)).astype(bool)
    for r, k in enumerate(rows[:128]):
        free = (np.minimum(hi[k], hi) - np.maximum(lo[k], lo)) <= 0
        free[k] = False
        assert np.array_equal(sub[r][free].astype(bool), (d[k] & d)[free])   # consensus.cpp:157-159 rule
    # set-max == greedy clique in the cmpTime order, on the bit rows
    mask = np.zeros(b.shape[1], dtype=np.uint64)
    ref = np.zeros(N, dtype=np.uint8)
    for k in oracle.candidate_order(g.loop_ids):
        if d[k] and np.array_equal(b[k] & mask, mask):
            ref[k] = 1
            mask[k >> 6] |= np.uint64(1) << np.uint64(k & 63)
    assert np.array_equal(acc, ref)
    cells = eng.cell_info()
    cores = os.cpu_count() or 1
    pick, info = _sample(cells, cfg, n_random=max(200, 40 * cores), max_cap=4 * cores)
    print("C5 sample:", info)
    _oracle_compare(oracle, g, cfg, cells, pick)


@pytest.mark.parametrize("name", ["C1", "C4s"])
def test_convergence_test_against_the_literal_g2o_loop(name, monkeypatch):
    """IPC_TERMINATE_EPS=0 runs g2o's literal loop (every solve ends through ~40 failing trials); the default
    stops a solve in the Newton regime once one more Gauss-Newton step cannot move an edge's chi2 by more
    than 2 sqrt(1e-13) (Se2View::term_eps).  Same decisions, same accepted set, chi2 within 1e-6."""
    from bench import build_workload
    from ipc_amd.consensus import IPC
    g, cfg, _ = build_workload(name)
    res = {}
    for mode, val in (("literal", "0"), ("default", None)):
        if val is None:
            monkeypatch.delenv("IPC_TERMINATE_EPS", raising=False)
        else:
            monkeypatch.setenv("IPC_TERMINATE_EPS", val)
        eng = IPC(g, cfg, device=0)
        bits, acc = eng.run()
        c = eng.cell_info()
        res[mode] = (bits.copy(), acc.copy(), c[np.lexsort((c["j"], c["i"]))])
        eng.close()
    a, b = res["literal"], res["default"]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    rel = np.abs(a[2]["max_chi2"] - b[2]["max_chi2"]) / np.maximum(np.abs(a[2]["max_chi2"]), 1e-300)
    assert float(rel.max()) <= 1e-6
    assert b[2]["evals"].sum() < 0.8 * a[2]["evals"].sum()        # the test does remove residual passes


def test_borderline_cells_are_solved_again_by_the_literal_loop(monkeypatch):
    """A cell whose max chi2 ends within IPC_BORDERLINE_BAND (default 4 sqrt(IPC_TERMINATE_EPS), relative) of its
    threshold is solved again with the convergence test off, so the test cannot have changed a decision.  With the band
    opened to 5 % the pass has cells to work on (C1): they are reported, carry the literal loop's chi2 (to 1e-6: two solvers, and
    the trial sequence that ends g2o's loop is rounding-driven) and nothing else of the matrix moves; a malformed IPC_TERMINATE_EPS is refused, not read as 0."""
    from bench import build_workload
    from ipc_amd.consensus import IPC
    g, cfg, _ = build_workload("C1")
    res = {}
    for mode, env in (("literal", {"IPC_TERMINATE_EPS": "0"}), ("default", {}), ("wide", {"IPC_BORDERLINE_BAND": "0.05"}),
                      ("off", {"IPC_BORDERLINE_BAND": "0"})):
        for k in ("IPC_TERMINATE_EPS", "IPC_BORDERLINE_BAND"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = IPC(g, cfg, device=0)
        bits, acc = eng.run()
        c = eng.cell_info()
        res[mode] = (bits.copy(), acc.copy(), c[np.lexsort((c["j"], c["i"]))], eng.solve_report())
        eng.close()
    lit, wide, off, dflt = res["literal"], res["wide"], res["off"], res["default"]
    assert lit[3]["literal_cells"] == 0 and off[3]["literal_cells"] == 0
    th = np.where(wide[2]["i"] == wide[2]["j"], cfg.fast_reject_th, cfg.slow_reject_th)
    near = np.abs(off[2]["max_chi2"] - th) <= 0.05 * th
    assert wide[3]["literal_cells"] == int(near.sum()) > 0, (wide[3], int(near.sum()))
    assert dflt[3]["literal_cells"] <= 2                          # a 1.3e-6 band: a cell or two of 50 000 at most
    for r in (wide, off, dflt):
        assert np.array_equal(r[0], lit[0]) and np.array_equal(r[1], lit[1])
    rel = np.abs(wide[2]["max_chi2"] - lit[2]["max_chi2"]) / np.maximum(np.abs(lit[2]["max_chi2"]), 1e-300)
    assert float(rel[near].max()) <= 1e-6, float(rel[near].max())
    assert np.array_equal(wide[2]["max_chi2"][~near], off[2]["max_chi2"][~near])
    monkeypatch.setenv("IPC_TERMINATE_EPS", "1e-13x")
    with pytest.raises(Exception, match="IPC_TERMINATE_EPS"):
        IPC(g, cfg, device=0)


def test_c5_row_shards_of_eight_ranks_reassemble_to_the_single_gpu_matrix():
    """BASELINE configs[4] in its own form -- SE3, N = 25 000, 8 ranks -- on one GPU: every rank's shard
    (ipc_solve_rows with the library's cost-balanced row assignment) is computed in turn into the layout the RCCL
    all-gather produces (rank-major, 8 x 3 125 rows x 391 words = 78 MB), then assembled and reduced: the same matrix
    bits and the same consensus set as the one-rank run.  (The collective itself only runs in the driver's multi-GPU bench.)"""
    from ipc_amd.dist import EngineBackend
    g, cfg, eng, ref_bits, ref_acc = _run("C5")
    world = 8
    b = EngineBackend(eng)
    rpr = (eng.N + world - 1) // world
    gathered = b.empty_words(world * rpr * eng.words)
    per_rank = []
    with b.stream_ctx():
        for r in range(world):
            b.solve_rows(r, world, gathered[r * rpr * eng.words:(r + 1) * rpr * eng.words])
            per_rank.append(len(eng.cell_info()))
        bits = b.empty_words(eng.N * eng.words)
        acc = b.empty_bytes(eng.N)
        b.assemble(gathered, world, bits)
        b.set_max(bits, acc)
    b.stream.synchronize()
    got = bits.cpu().numpy().view(np.uint64).reshape(eng.N, eng.words)
    assert np.array_equal(got, ref_bits)
    assert np.array_equal(acc.cpu().numpy(), ref_acc)
    total = sum(per_rank)
    assert max(per_rank) <= 1.25 * total / world, per_rank         # solved cells per rank: the cost balance holds on SE3 too
    print("C5, 8 ranks: solved cells per rank", per_rank)


def test_a_wide_borderline_band_costs_a_fraction_of_a_step(monkeypatch):
    """Round 5: the borderline cells are solved again by the cell kernels themselves over compact lists built on the device
    (no per-cell host copies, no host-driven solves).  With the band opened to 5 % (hundreds of literal cells on C1) a
    step stays within 1.5 x of the default step; the literal cells carry the bits of a run with the convergence test off."""
    import time
    from bench import build_workload
    from ipc_amd.consensus import IPC
    g, cfg, _ = build_workload("C1")

    def timed(env):
        for k in ("IPC_TERMINATE_EPS", "IPC_BORDERLINE_BAND"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = IPC(g, cfg, device=0)
        eng.run()                                    # plan, buffers, code objects
        eng.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            bits, acc = eng.run()
            ts.append(time.perf_counter() - t0)
        c = eng.cell_info()
        rep = eng.solve_report()
        eng.close()
        return float(np.median(ts)), bits, acc, c[np.lexsort((c["j"], c["i"]))], rep

    t_def, b_def, a_def, c_def, r_def = timed({})
    t_wide, b_wide, a_wide, c_wide, r_wide = timed({"IPC_BORDERLINE_BAND": "0.05"})
    t_lit, b_lit, a_lit, c_lit, _ = timed({"IPC_TERMINATE_EPS": "0"})
    assert r_wide["literal_cells"] >= 100, r_wide
    print("\n[borderline band 5 %%] %d literal cells, step %.1f ms against %.1f ms by default" % (r_wide["literal_cells"], 1e3 * t_wide, 1e3 * t_def))
    assert t_wide <= 1.5 * t_def, (t_wide, t_def)
    assert np.array_equal(b_wide, b_lit) and np.array_equal(a_wide, a_lit)
    th = np.where(c_def["i"] == c_def["j"], cfg.fast_reject_th, cfg.slow_reject_th)
    near = np.abs(c_def["max_chi2"] - th) <= 0.05 * th
    assert int(near.sum()) == r_wide["literal_cells"]
    assert np.array_equal(c_wide["max_chi2"][near].view(np.uint64), c_lit["max_chi2"][near].view(np.uint64))
    assert np.array_equal(c_wide["iterations"][near], c_lit["iterations"][near])


def test_repeated_steps_reuse_the_cell_lists_and_an_appended_candidate_renews_them():
    """ipc_solve_rows keeps the cell lists of (rank, world) between steps (one host wait per step instead of two); the lists
    are rebuilt when the candidates change (ipc_append_candidate), when another rank's rows are asked for, and by the
    set-only phases, which plan from the step's diagonal bits."""
    from bench import build_workload
    from ipc_amd import synth
    from ipc_amd.consensus import IPC
    g, cfg, _ = build_workload("tiny")
    eng = IPC(g, cfg, device=0)
    b0, a0 = eng.run()
    for _ in range(3):
        b, a = eng.run()
        assert np.array_equal(b, b0) and np.array_equal(a, a0)
    acc_so, _ = eng.run_set_only()
    assert np.array_equal(acc_so, a0)
    b, a = eng.run()
    assert np.array_equal(b, b0) and np.array_equal(a, a0)
    # one more candidate: the matrix of the longer list = a fresh engine on that list
    extra = synth.inject_outliers(g, 1, seed=77)
    k = eng.append_candidate(extra.loop_ids[-1], extra.loop_meas[-1], extra.loop_info[-1])
    assert k == g.N
    b1, a1 = eng.run()
    ref = IPC(extra, cfg, device=0)
    br, ar = ref.run()
    assert np.array_equal(b1, br) and np.array_equal(a1, ar)
    eng.close(); ref.close()
