"""C++ host mirror (ipc_amd/host): the testers keep the reference's CLI, config keys and output
files (examples/ipc_tester_2D.cpp:13-17, src/utils.cpp:316-337, src/simulation.cpp:91-105)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
HOST = os.path.join(ROOT, "ipc_amd", "host")

CFG = """name : "{name}"
dataset : "{dataset}"
ground_truth : "unused.txt"
output : "{output}"
visualize : 0
canonic_inliers : {inl}
fast_reject_th : {fth}
fast_reject_iter_base : 50
slow_reject_th : {sth}
slow_reject_iter_base : 100
s_factor : {s}          # injected by the reference's bash drivers with yq
use_best_k_buddies : false
k_buddies : 2
use_recovery : true
"""


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    return {d: os.path.join(HOST, "ipc_tester_%dD" % d) for d in (2, 3)}


def _write_cfg(tmp_path, dim, **over):
    vals = dict(name="small", dataset=os.path.join(GOLD, "small_se2_spoiled_n6_seed3.g2o" if dim == 2 else
                                                   "small_se3_spoiled_n5_seed4.g2o"),
                output=str(tmp_path / "res.txt"), inl=8 if dim == 2 else 6, fth=6.251,
                sth=11.345 if dim == 2 else 6.251, s=10.0 if dim == 2 else 50.0)
    vals.update(over)
    p = tmp_path / "cfg.yaml"
    p.write_text(CFG.format(**vals))
    return str(p), vals


def test_usage_and_loud_errors(built, tmp_path):
    r = subprocess.run([built[2]], capture_output=True, text=True)
    assert r.returncode == 2 and "-c <cfg.yaml>" in r.stderr
    r = subprocess.run([built[2], "-c", str(tmp_path / "nope.yaml")], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot open" in r.stderr
    # the shipped reference YAMLs lack s_factor & co: readConfig must refuse them like yaml-cpp does
    bad = tmp_path / "bad.yaml"
    bad.write_text("\n".join(l for l in CFG.format(name="x", dataset="d", output="o.txt", inl=1, fth=1, sth=1, s=1)
                             .splitlines() if not l.startswith("s_factor")))
    r = subprocess.run([built[2], "-c", str(bad)], capture_output=True, text=True)
    assert r.returncode == 1 and "missing key 's_factor'" in r.stderr
    # wrong dimension for the binary
    cfg, _ = _write_cfg(tmp_path, 3)
    r = subprocess.run([built[2], "-c", cfg], capture_output=True, text=True)
    assert r.returncode == 1 and "not a 2D graph" in r.stderr


GEN_CASES = [
    ("small_se2_clean.g2o", "small_se2_spoiled_n6_seed3.g2o", ["-n", "6", "--seed", "3"]),
    ("small_se2_clean.g2o", "small_se2_local_spoiled_n5_seed9.g2o", ["-n", "5", "--seed", "9", "--local"]),
    ("small_se2_clean.g2o", "small_se2_group_spoiled_n3_seed5.g2o", ["-n", "3", "--seed", "5", "-g", "2"]),
    ("small_se3_clean.g2o", "small_se3_spoiled_n5_seed4.g2o", ["-n", "5", "--seed", "4"]),
]


@pytest.mark.parametrize("clean,spoiled,flags", GEN_CASES)
def test_generate_dataset_reproduces_the_reference_script_byte_for_byte(built, tmp_path, clean, spoiled, flags):
    """The golden *_spoiled_* files are outputs of the reference's scripts/generateDataset.py
    (tests/golden/make_golden.py); the C++ generator must write the same bytes."""
    out = str(tmp_path / "out.g2o")
    r = subprocess.run([os.path.join(HOST, "generateDataset"), "-i", os.path.join(GOLD, clean), "-o", out] + flags,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(out, "rb").read() == open(os.path.join(GOLD, spoiled), "rb").read()


@pytest.mark.parametrize("dim,n,seed,extra", [(2, 1000, 1000, []), (3, 300, 77, []), (2, 50, 5, ["-p"]),
                                               (2, 40, 6, ["--information=42"])])
def test_generate_dataset_matches_python_generator(built, tmp_path, dim, n, seed, extra):
    """Same draws as ipc_amd.synth.sample_outliers (which wraps CPython's own random module)."""
    from ipc_amd import graphio, synth
    g = synth.small_se2() if dim == 2 else synth.small_se3()
    clean, out = str(tmp_path / "clean.g2o"), str(tmp_path / "out.g2o")
    graphio.write_g2o(clean, g)
    r = subprocess.run([os.path.join(HOST, "generateDataset"), "-i", clean, "-o", out, "-n", str(n), "--seed",
                        str(seed)] + extra, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = graphio.read_g2o(out)
    ids, meas = synth.sample_outliers(dim, g.V, n, seed, perfect="-p" in extra)
    assert got.N == g.N + n
    assert np.array_equal(got.loop_ids[g.N:], ids)
    assert np.array_equal(got.loop_meas[g.N:], meas)
    if "--information=42" in extra:
        assert np.array_equal(got.loop_info[g.N:], np.tile([42.0, 0, 0, 42.0, 0, 42.0], (n, 1)))
    else:
        assert np.array_equal(got.loop_info[g.N:], np.tile(g.loop_info[0], (n, 1)))


@pytest.mark.gpu
@pytest.mark.parametrize("dim", [2, 3])
def test_tester_outputs_match_python_path(built, tmp_path, dim):
    cfg, vals = _write_cfg(tmp_path, dim)
    env = dict(os.environ, IPC_AMD_MODE="matrix")                  # the batched formulation (opt-in)
    r = subprocess.run([built[dim], "-c", cfg], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert "Size of MAX consistent set" in r.stdout and "Precision" in r.stdout and "Recall" in r.stdout
    exp = np.load(os.path.join(GOLD, ("small_se2" if dim == 2 else "small_se3") + "_expected.npz"))
    acc = exp["accepted"]
    inl = vals["inl"]
    tp, fp = int(acc[:inl].sum()), int(acc[inl:].sum())
    fn = inl - tp
    pr = open(str(tmp_path / "res.PR")).read().split()
    assert float(pr[0]) == pytest.approx(tp / (tp + fp), rel=1e-5)
    assert float(pr[1]) == pytest.approx(tp / (tp + fn), rel=1e-5)
    assert float(pr[3]) == pytest.approx(float(pr[2]) / len(acc), rel=1e-3)
    assert ("Size of MAX consistent set = %d" % int(acc.sum())) in r.stdout
    traj = np.loadtxt(str(tmp_path / "res.txt"))
    assert traj.shape[1] == (3 if dim == 2 else 7)
    assert np.allclose(traj[0][:3], 0)
    # the trajectory is the final map (reference src/simulation.cpp:50-65,91-98); the file
    # carries operator<<'s 6 significant digits
    from ipc_amd import graphio
    from ipc_amd.consensus import IPC, Config
    g = graphio.read_g2o(vals["dataset"])
    eng = IPC(g, Config(vals["fth"], 50, vals["sth"], 100, vals["s"]))
    poses, _ = eng.final_optimize(acc)
    if dim == 2:
        assert np.allclose(traj, poses, rtol=2e-5, atol=2e-5)
        assert not np.allclose(traj, eng.initial_poses(), atol=1e-3)
    else:
        assert np.allclose(traj[:, :3], poses[:, 9:], rtol=2e-5, atol=2e-5)
        # writeVertex prints Quaterniond(R) as qx qy qz qw (src/utils.cpp:248-258)
        q = traj[:, 3:]
        x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                      2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                      2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], axis=1)
        assert np.allclose(R, poses[:, :9], atol=1e-4)


@pytest.mark.gpu
def test_tester_incremental_mode_matches_python_path(built, tmp_path):
    """The tester's default is the reference's own per-candidate loop (src/simulation.cpp:34-47)."""
    from ipc_amd import graphio
    from ipc_amd.consensus import IPC, Config
    cfg, vals = _write_cfg(tmp_path, 2)
    env = {k: v for k, v in os.environ.items() if k != "IPC_AMD_MODE"}
    r = subprocess.run([built[2], "-c", cfg], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    g = graphio.read_g2o(vals["dataset"])
    eng = IPC(g, Config(vals["fth"], 50, vals["sth"], 100, vals["s"]))
    eng.reset()
    acc = np.zeros(g.N, dtype=np.uint8)
    for k in eng.candidate_order():
        acc[k] = eng.agreementCheck(k)
    assert ("Size of MAX consistent set = %d" % int(acc.sum())) in r.stdout
    traj = np.loadtxt(str(tmp_path / "res.txt"))
    poses, _ = eng.final_optimize(acc)
    assert np.allclose(traj, poses, rtol=2e-5, atol=2e-5)
