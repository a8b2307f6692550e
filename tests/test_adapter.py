"""include/ipc/consensus_amd.hpp -- the IPC<EDGE, VERTEX> class of the reference (include/ipc/consensus.hpp:5-33)
over the C ABI -- compiled against a small mock of the g2o types it touches (tests/mock_ref/), since g2o is
not installed here; on a GPU box the same binary replays the reference's harness loop on the golden inputs."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
MOCK = os.path.join(ROOT, "tests", "mock_ref")


def _compile(out, link):
    # this repo's include/ in FRONT of the (mock of the) reference's: "ipc/consensus.hpp" resolves to the shim
    cmd = ["g++", "-O1", "-std=c++14", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-I" + MOCK,
           os.path.join(MOCK, "adapter_main.cpp")]
    if link:
        cmd += ["-o", out, "-L" + os.path.join(ROOT, "ipc_amd"), "-lipc_amd", "-Wl,-rpath," + os.path.join(ROOT, "ipc_amd"),
                "-Wl,-rpath,/opt/rocm/lib"]
    else:
        cmd += ["-c", "-o", out]
    subprocess.check_call(cmd)


def test_adapter_header_compiles_as_cxx14_for_both_pose_types(tmp_path):
    """The reference builds with -std=c++14 (CMakeLists.txt:4); adapter_main instantiates
    IPC<EdgeSE2, VertexSE2> and IPC<EdgeSE3, VertexSE3> with every member function."""
    _compile(str(tmp_path / "adapter_main.o"), link=False)


def _run_adapter(tmp_path, dim, path, prm, order="ref", where="graph", env=None, inliers=0):
    exe = str(tmp_path / "adapter_main")
    if not os.path.exists(exe):
        _compile(exe, link=True)
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([exe, str(dim), path] + [str(v) for v in prm] + [order, where, str(inliers)], capture_output=True, text=True, env=e)
    assert r.returncode == 0, r.stderr
    return dict(l.split(" ", 1) for l in r.stdout.strip().splitlines())


def _python_path(g, prm, order):
    """The same calls through the Python mirror, in the order the harness used."""
    from ipc_amd.consensus import IPC, Config
    eng = IPC(g, Config(prm[1], prm[2], prm[3], prm[4], prm[0]))
    eng.reset()
    dec = [int(eng.agreementCheck(int(k))) for k in order]
    eng.close()
    return dec


SMALL = [(2, "small_se2_spoiled_n6_seed3.g2o"), (3, "small_se3_spoiled_n5_seed4.g2o")]


def _prm(dim):
    return (10.0, 6.251, 50, 11.345, 100) if dim == 2 else (50.0, 6.251, 50, 6.251, 100)


@pytest.mark.gpu
@pytest.mark.parametrize("dim,spoiled", SMALL)
@pytest.mark.parametrize("order", ["ref", "stable"])
def test_adapter_replays_the_harness_loop(tmp_path, dim, spoiled, order):
    """src/simulation.cpp:28-56 with the harness literally unchanged: IPC(problem, cfg), then the candidates one by one,
    never announced.  order = ref: the harness's own std::sort + cmpTime over the graph's (address-ordered) edge set,
    which the adapter's constructor predicts; stable: ties in file order, a caller the prediction gets wrong on ties."""
    from ipc_amd import graphio
    from ipc_amd.consensus import IPC, Config
    path = os.path.join(GOLD, spoiled)
    prm = _prm(dim)
    lines = _run_adapter(tmp_path, dim, path, prm, order=order)
    g = graphio.read_g2o(path)
    called = [int(x) for x in lines["order"].split()]
    assert sorted(called) == list(range(g.N))
    last = g.loop_ids.max(axis=1)[called]
    assert np.all(np.diff(last) >= 0)                              # cmpTime
    if order == "stable":
        assert called == list(graphio.candidate_order(g.loop_ids))
    want = _python_path(g, prm, called)
    assert [int(x) for x in lines["decisions"].split()] == want
    assert int(lines["set"]) == sum(want)
    # ... and with the list announced first
    called2 = [int(x) for x in lines["announced_order"].split()]
    assert [int(x) for x in lines["announced"].split()] == (want if called2 == called else _python_path(g, prm, called2))
    # the constructor scaled the caller's odometry information in place (robustifyVoters, src/consensus.cpp:21) and the
    # harness's own division (src/simulation.cpp:56) brings back what the file holds, before its final optimize(1000)
    assert float(lines["ctor_info_scale"]) <= 1e-15
    assert float(lines["harness_info_restore"]) <= 4e-16
    assert lines["harness_vertices_propagated"] == "1"
    if sum(want):
        assert lines["removed"] == "1 -> %d" % (sum(want) - 1)     # by an edge OBJECT the engine never saw, joining the
        assert lines["added"] == "-> %d" % sum(want)               # pair the other way round: (min id, max id) decide
        assert lines["added_twin"] == "-> %d" % sum(want)          # a member joins that pair already
    eng = IPC(g, Config(prm[1], prm[2], prm[3], prm[4], prm[0]))
    _, acc = eng.run()
    assert int(lines["matrix"].split()[1]) == int(acc.sum())
    assert lines["cleared"] == "0"                                 # ~IPC clears the caller's graph (src/consensus.cpp:38)


@pytest.mark.gpu
@pytest.mark.parametrize("dim,spoiled", SMALL)
def test_adapter_with_candidates_that_are_not_in_the_graph(tmp_path, dim, spoiled):
    """Every candidate is an edge object the engine has never seen (the graph holds the odometry only): each
    agreementCheck appends one record (ipc_append_candidate) and checks it."""
    from ipc_amd import graphio
    path = os.path.join(GOLD, spoiled)
    prm = _prm(dim)
    g = graphio.read_g2o(path)
    for order in ("ref", "stable"):
        lines = _run_adapter(tmp_path, dim, path, prm, order=order, where="foreign")
        called = [int(x) for x in lines["order"].split()]
        assert [int(x) for x in lines["decisions"].split()] == _python_path(g, prm, called)


@pytest.mark.gpu
@pytest.mark.parametrize("dim,stem", [(2, "small_se2_dup_pairs"), (3, "small_se3_dup_pairs")])
@pytest.mark.parametrize("where", ["graph", "foreign"])
def test_adapter_checks_the_edge_it_is_given_on_duplicate_vertex_pairs(tmp_path, where, dim, stem):
    """Three vertex pairs carry two candidates each, with different measurements and different verdicts
    (tests/golden/make_dup_pair_golden.py, SE2 and SE3): agreementCheck judges the object it is handed
    (src/consensus.cpp:43-56), not an earlier candidate on the same pair.  Expected decisions: the CPU oracle's (committed fixture)."""
    from ipc_amd import graphio
    path = os.path.join(GOLD, stem + ".g2o")
    exp = np.load(os.path.join(GOLD, stem + "_expected.npz"))
    prm = _prm(dim)
    g = graphio.read_g2o(path)
    lines = _run_adapter(tmp_path, dim, path, prm, order="stable", where=where)
    called = [int(x) for x in lines["order"].split()]
    assert called == list(exp["order"])
    got = [int(x) for x in lines["decisions"].split()]
    assert got == list(exp["decision"])
    by_index = dict(zip(called, got))
    for first, second in exp["duplicates"]:
        assert tuple(sorted(g.loop_ids[first])) == tuple(sorted(g.loop_ids[second]))
    assert [by_index[int(a)] != by_index[int(b)] for a, b in exp["duplicates"][:2]] == [True, True]
    lines = _run_adapter(tmp_path, dim, path, prm, order="ref", where=where)
    called = [int(x) for x in lines["order"].split()]
    assert [int(x) for x in lines["decisions"].split()] == _python_path(g, prm, called)


@pytest.mark.gpu
@pytest.mark.parametrize("dim,spoiled", SMALL)
def test_adapter_writes_the_estimates_back_on_request(tmp_path, dim, spoiled):
    """src/consensus.cpp:69-71 leaves the optimised window and the re-propagated tail in the caller's graph; the adapter
    does so behind setWriteBackEstimates(true)."""
    lines = _run_adapter(tmp_path, dim, os.path.join(GOLD, spoiled), _prm(dim), env={"IPC_ADAPTER_WRITE_BACK": "1", "IPC_ADAPTER_LOOP_ONLY": "1"})
    assert int(lines["set"]) > 0
    assert float(lines["written_back"]) == 0.0


def _bench_graph_file(tmp_path, workload):
    import bench
    from ipc_amd import graphio
    g, cfg, _ = bench.build_workload(workload)
    path = str(tmp_path / (workload + ".g2o"))
    graphio.write_g2o(path, g)
    g2 = graphio.read_g2o(path)
    assert np.array_equal(g2.loop_meas, g.loop_meas) and np.array_equal(g2.odom_info, g.odom_info)
    prm = (cfg.s_factor, cfg.fast_reject_th, cfg.fast_reject_iter_base, cfg.slow_reject_th, cfg.slow_reject_iter_base)
    return g2, path, prm


@pytest.mark.gpu
def test_unchanged_harness_over_all_of_c2(tmp_path, capsys):
    """BASELINE configs[1] through the adapter binary with the harness's own sequence (constructor, then agreementCheck
    per candidate, nothing announced): decisions of the committed oracle run, at pipeline speed -- the constructor read the
    loop edges from the graph.  Then the same with the candidates kept OUT of the graph (one append per check)."""
    g, path, prm = _bench_graph_file(tmp_path, "C2")
    exp = np.load(os.path.join(GOLD, "c2_incremental_expected.npz"))
    env = {"IPC_ADAPTER_LOOP_ONLY": "1"}
    rates = {}
    lines = _run_adapter(tmp_path, 2, path, prm, order="stable", env=env)
    assert [int(x) for x in lines["order"].split()] == list(exp["order"])
    assert [int(x) for x in lines["decisions"].split()] == list(exp["decision"])
    rates["graph, ties in file order"] = float(lines["seconds"].split()[2])
    lines = _run_adapter(tmp_path, 2, path, prm, order="ref", env=env, inliers=256)
    called = [int(x) for x in lines["order"].split()]
    assert [int(x) for x in lines["decisions"].split()] == _python_path(g, prm, called)
    rates["graph, the harness's own order"] = float(lines["seconds"].split()[2])
    lines = _run_adapter(tmp_path, 2, path, prm, order="stable", where="foreign", env=env)
    assert [int(x) for x in lines["decisions"].split()] == list(exp["decision"])
    rates["candidates not in the graph (append per check)"] = float(lines["seconds"].split()[2])
    with capsys.disabled():
        print("\n[adapter, C2, 1256 candidates] candidates/s:", rates)
    assert rates["graph, the harness's own order"] >= 500.0
    assert rates["graph, ties in file order"] >= 300.0             # (a caller the constructor's prediction gets wrong on ties)


@pytest.mark.gpu
def test_unchanged_harness_over_all_of_c3_with_its_duplicate_pairs(tmp_path, capsys):
    """BASELINE configs[2] (MIT-like, 5020 candidates, 36 vertex pairs that carry two candidates each)."""
    g, path, prm = _bench_graph_file(tmp_path, "C3")
    pairs = {}
    for k, (a, b) in enumerate(g.loop_ids):
        pairs.setdefault((int(a), int(b)), []).append(k)
    dups = [v for v in pairs.values() if len(v) > 1]
    assert len(dups) >= 30
    lines = _run_adapter(tmp_path, 2, path, prm, order="ref", env={"IPC_ADAPTER_LOOP_ONLY": "1"}, inliers=20)
    called = [int(x) for x in lines["order"].split()]
    got = [int(x) for x in lines["decisions"].split()]
    want = _python_path(g, prm, called)
    assert got == want
    by_index = dict(zip(called, got))
    differing = sum(len({by_index[k] for k in v}) > 1 for v in dups)
    with capsys.disabled():
        print("\n[adapter, C3, %d candidates] %s candidates/s, %d accepted, %d duplicate pairs (%d with differing verdicts)"
              % (g.N, lines["seconds"].split()[2], sum(got), len(dups), differing))
