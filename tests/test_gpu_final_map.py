"""The harness's final map at BASELINE sizes (north_star: "final chi2 within 1e-5
relative").  After the agreementCheck loop the reference un-scales the odometry information, adds the accepted loops and
runs optimize(1000) over the whole graph (src/simulation.cpp:50-65): ipc_final_optimize.  Rounds 2 - 5 tested it on graphs
of V <= 500; here on the faithful runs' accepted sets of C1 (254 loops), C2 (160) and C4 (1 993 loops over the whole
2 500-pose sphere: the banded solver), against the oracle's optimisation of the same graph
(tests/golden/make_final_map_golden.py): total chi2 and max edge chi2 within 1e-5, poses within 1e-6."""
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL = 1e-5
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rot_angle(Ra, Rb):
    Ra, Rb = Ra.reshape(-1, 3, 3), Rb.reshape(-1, 3, 3)
    tr = np.einsum("nij,nij->n", Ra, Rb)
    return np.arccos(np.clip((tr - 1.0) / 2.0, -1.0, 1.0))


@pytest.mark.parametrize("tag", ["c1", "c2", "c4", "c5"])
def test_final_map_of_the_faithful_accepted_set_against_the_oracle(tag):
    import bench
    from ipc_amd.consensus import IPC
    path = os.path.join(GOLD, "%s_final_map_expected.npz" % tag)
    if not os.path.exists(path):
        pytest.skip("no oracle fixture for %s" % tag)
    fx = np.load(path)
    g, cfg, _ = bench.build_workload(str(fx["workload"]))
    assert int(np.asarray(g.loop_ids, dtype=np.int64).sum()) == int(fx["loop_ids_checksum"]), "workload changed"
    assert abs(float(np.asarray(g.loop_meas).sum()) - float(fx["meas_checksum"])) < 1e-9, "workload changed"
    eng = IPC(g, cfg, device=0)
    eng.final_optimize(fx["accepted"], iterations=1000)                      # (warm-up: workspaces)
    t0 = time.perf_counter()
    poses, info = eng.final_optimize(fx["accepted"], iterations=1000)
    dt = time.perf_counter() - t0
    assert info.n_cluster_loops == int(fx["accepted"].sum())
    assert (info.flags & 2) == 0
    ref_tot, ref_max = float(fx["chi2_total"]), float(fx["max_chi2"])
    assert abs(info.chi2_initial - float(fx["chi2_initial"])) <= 1e-9 * float(fx["chi2_initial"])
    assert abs(info.chi2_total - ref_tot) <= REL * ref_tot, (info.chi2_total, ref_tot)
    assert abs(info.max_chi2 - ref_max) <= REL * ref_max, (info.max_chi2, ref_max)
    ref = fx["poses"]
    if g.dim == 2:
        assert np.abs(poses[:, :2] - ref[:, :2]).max() <= 1e-6
        assert np.abs(np.angle(np.exp(1j * (poses[:, 2] - ref[:, 2])))).max() <= 1e-6
    else:
        scale = max(1.0, np.abs(ref[:, 9:]).max())
        assert np.abs(poses[:, 9:] - ref[:, 9:]).max() <= 1e-6 * scale
        assert _rot_angle(poses[:, :9], ref[:, :9]).max() <= 1e-6
    print("\n[%s final map] %d loops, chi2 %.6g -> %.9g (oracle %.9g), %d iterations (oracle %d), %.3f s on the GPU, oracle %.1f s"
          % (tag, info.n_cluster_loops, info.chi2_initial, info.chi2_total, ref_tot, info.iterations, int(fx["iterations"]), dt,
             float(fx["oracle_seconds_authoring_container"])))
    eng.close()
