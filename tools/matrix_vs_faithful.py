"""Lists, for the bench workloads C1 and C2, the candidates on which the batched matrix + set-max formulation and the
reference's own incremental algorithm (ipc_agreement_check; fixture = the CPU oracle's run) decide differently, and
writes tests/golden/matrix_vs_faithful_expected.json.  Run on the GPU box: python tools/matrix_vs_faithful.py [out.json]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from ipc_amd.consensus import IPC
    out = {}
    for wl, tag in (("C1", "c1"), ("C2", "c2")):
        g, cfg, desc = bench.build_workload(wl)
        exp = np.load(os.path.join(ROOT, "tests", "golden", "%s_incremental_expected.npz" % tag))
        eng = IPC(g, cfg)
        order = eng.candidate_order()
        faithful = np.zeros(g.N, dtype=bool)
        faithful[order] = exp["decision"].astype(bool)
        # the GPU's own faithful run must be the fixture's
        eng.reset()
        gpu = np.zeros(g.N, dtype=bool)
        for k in order:
            gpu[k] = eng.agreementCheck(int(k))
        assert np.array_equal(gpu, faithful), "GPU faithful run differs from the oracle fixture"
        _, acc = eng.run()
        m = acc.astype(bool)
        inl = cfg.canonic_inliers
        out[wl] = dict(
            workload=desc, candidates=int(g.N), true_loops=int(inl),
            accepted_matrix=int(m.sum()), accepted_reference_algorithm=int(faithful.sum()),
            accepted_by_matrix_mode_only=[int(k) for k in np.nonzero(m & ~faithful)[0]],
            accepted_by_the_reference_algorithm_only=[int(k) for k in np.nonzero(~m & faithful)[0]],
            injected_outliers_accepted_by_matrix_mode=int(m[inl:].sum()),
            injected_outliers_accepted_by_the_reference_algorithm=int(faithful[inl:].sum()))
        eng.close()
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "matrix_vs_faithful_expected.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if not isinstance(vv, list)} for k, v in out.items()}))


if __name__ == "__main__":
    main()
