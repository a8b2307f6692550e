"""Host-side mirror of the reference's consensus interface over the C ABI.

`IPC` follows the surface of the reference's `template<class EDGE, class VERTEX> class IPC`
(reference include/ipc/consensus.hpp:5-33): constructed from the open-loop problem and a
Config, it owns the consensus set.  The batched entry points (`consistency_matrix`,
`max_consensus_set`) are the MI355X re-formulation of the per-candidate `agreementCheck` loop
(reference src/simulation.cpp:34-47): all cells of the N x N consistency matrix are solved in
one pass on the GPU, then the consistent set is grown in the reference's candidate order.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import capi
from .graphio import PoseGraph


@dataclass
class Config:
    """The hot-path fields of the reference's struct Config (include/ipc/utils.hpp:22-38)."""
    fast_reject_th: float = 6.251
    fast_reject_iter_base: int = 50
    slow_reject_th: float = 11.345
    slow_reject_iter_base: int = 100
    s_factor: float = 10.0
    canonic_inliers: int = 0


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class IPC:
    def __init__(self, graph: PoseGraph, cfg: Config, device: int = 0):
        self.lib = capi.load()
        self.graph, self.cfg, self.device = graph, cfg, device
        self.dim = graph.dim
        prm = capi.Params(cfg.fast_reject_th, cfg.fast_reject_iter_base, cfg.slow_reject_th,
                          cfg.slow_reject_iter_base, cfg.s_factor)
        om, oi = _d(graph.odom_meas), _d(graph.odom_info)
        h = C.c_void_p()
        capi.check(self.lib.ipc_create(graph.dim, graph.V, _p(om), _p(oi), C.byref(prm), device, C.byref(h)))
        self.h = h
        self.N = 0
        self._max_consensus_set = np.zeros(0, dtype=np.int32)
        if graph.N:
            self.set_candidates(graph.loop_ids, graph.loop_meas, graph.loop_info)

    # ---- candidates ---------------------------------------------------------------------
    def set_candidates(self, ids, meas, info):
        ids = np.ascontiguousarray(ids, dtype=np.int32).reshape(-1, 2)
        meas, info = _d(meas), _d(info)
        capi.check(self.lib.ipc_set_candidates(self.h, ids.shape[0], _p(ids), _p(meas), _p(info)))
        self.N = ids.shape[0]
        self.ids = ids

    def append_candidate(self, ids, meas, info):
        """One more candidate at the end of the list (ipc_append_candidate: the harness hands agreementCheck an edge
        nobody announced, reference src/simulation.cpp:34-47); returns its index.  State and solves in flight stay."""
        ids = np.ascontiguousarray(ids, dtype=np.int32).reshape(2)
        meas, info = _d(meas), _d(info)
        k = C.c_int(-1)
        capi.check(self.lib.ipc_append_candidate(self.h, _p(ids), _p(meas), _p(info), C.byref(k)))
        self.N = k.value + 1
        self.ids = np.vstack([self.ids, ids.reshape(1, 2)]) if self.N > 1 else ids.reshape(1, 2)
        return k.value

    def candidate_order(self):
        order = np.zeros(self.N, dtype=np.int32)
        capi.check(self.lib.ipc_candidate_order(self.h, _p(order)))
        return order

    def initial_poses(self):
        ps = 3 if self.dim == 2 else 12
        out = np.zeros((self.graph.V, ps))
        capi.check(self.lib.ipc_initial_poses(self.h, _p(out)))
        return out

    @property
    def words(self):
        return (self.N + 63) // 64

    # ---- single-GPU batch path ------------------------------------------------------------
    def run(self):
        """Solve the matrix and grow the consensus set; returns (bits [N, words] uint64,
        accepted [N] uint8)."""
        bits = np.zeros((self.N, self.words), dtype=np.uint64)
        acc = np.zeros(self.N, dtype=np.uint8)
        capi.check(self.lib.ipc_run(self.h, _p(bits), _p(acc)))
        order = self.candidate_order()
        self._max_consensus_set = order[acc[order] == 1]
        return bits, acc

    def run_set_only(self):
        """The accepted set of run() without the cells the set-max never reads (diagonal first, then the pairs among the
        candidates whose own cell passed); returns (accepted [N] uint8, cells actually solved)."""
        acc = np.zeros(self.N, dtype=np.uint8)
        n = C.c_int(0)
        capi.check(self.lib.ipc_run_set_only(self.h, _p(acc), C.byref(n)))
        order = self.candidate_order()
        self._max_consensus_set = order[acc[order] == 1]
        return acc, n.value

    def consistency_matrix(self):
        bits, _ = self.run()
        return unpack_bits(bits, self.N)

    def getMaxConsensusSet(self):
        """Candidate indices in acceptance order (reference consensus.hpp:16)."""
        return self._max_consensus_set

    # ---- the reference's own per-candidate interface (faithful incremental mode) -------------
    def reset(self):
        """State right after the constructor (reference src/consensus.cpp:23-27)."""
        capi.check(self.lib.ipc_incremental_reset(self.h))
        self._max_consensus_set = np.zeros(0, dtype=np.int32)

    def agreementCheck(self, k, with_info=False):
        """IPC::agreementCheck (reference src/consensus.cpp:43-75) for candidate k (file index):
        cluster solve from the current state on the GPU; True when every edge passes."""
        ok, info = C.c_int(0), capi.CheckInfo()
        capi.check(self.lib.ipc_agreement_check(self.h, int(k), C.byref(ok), C.byref(info)))
        self._max_consensus_set = self._consensus()
        return (bool(ok.value), info) if with_info else bool(ok.value)

    def _consensus(self):
        n = C.c_int(0)
        capi.check(self.lib.ipc_consensus_size(self.h, C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=np.int32)
        capi.check(self.lib.ipc_consensus_set(self.h, _p(out)))
        return out[:n.value]

    def removeEdgeFromCnS(self, k):
        """IPC::removeEdgeFromCnS (reference src/consensus.cpp:77-96)."""
        r = C.c_int(0)
        capi.check(self.lib.ipc_remove_from_consensus(self.h, int(k), C.byref(r)))
        self._max_consensus_set = self._consensus()
        return bool(r.value)

    def addEdgeToCnS(self, k):
        """IPC::addEdgeToCnS (reference src/consensus.cpp:98-119)."""
        capi.check(self.lib.ipc_add_to_consensus(self.h, int(k)))
        self._max_consensus_set = self._consensus()

    def current_poses(self):
        out = np.zeros((self.graph.V, 3 if self.dim == 2 else 12))
        capi.check(self.lib.ipc_current_poses(self.h, _p(out)))
        return out

    def set_state(self, poses, consensus, resume_position=0):
        """Resume the agreementCheck loop from a saved (current_poses(), getMaxConsensusSet()) pair
        (ipc_incremental_set_state; the two are all the state of reference include/ipc/consensus.hpp:23-32)."""
        poses = _d(poses)
        assert poses.shape == (self.graph.V, 3 if self.dim == 2 else 12)
        cns = np.ascontiguousarray(consensus, dtype=np.int32)
        capi.check(self.lib.ipc_incremental_set_state(self.h, _p(poses), _p(cns), int(cns.shape[0]), int(resume_position)))
        self._max_consensus_set = cns.copy()

    def incremental_counters(self):
        """dict(host_solver_fallbacks, lost_launches, relaunches, literal_band_solves) -- ipc_incremental_counters."""
        c = capi.IncrementalCounters()
        capi.check(self.lib.ipc_incremental_counters(self.h, C.byref(c)))
        return {k: getattr(c, k) for k, _ in capi.IncrementalCounters._fields_}

    def final_optimize(self, accepted, iterations=1000):
        """The harness's final map (reference src/simulation.cpp:50-65): returns (poses [V,3] or
        [V,12] (R row-major, t), CheckInfo with chi2_total)."""
        acc = np.ascontiguousarray(accepted, dtype=np.uint8)
        out = np.zeros((self.graph.V, 3 if self.dim == 2 else 12))
        info = capi.CheckInfo()
        capi.check(self.lib.ipc_final_optimize(self.h, _p(acc), int(iterations), _p(out), C.byref(info)))
        return out, info

    # ---- device-pointer stages (multi-GPU plumbing lives in ipc_amd.dist) -------------------
    def rows_per_rank(self, world):
        return self.lib.ipc_rows_per_rank(self.N, world)

    def solve_rows(self, rank, world, d_upper_ptr, stream=0):
        capi.check(self.lib.ipc_solve_rows(self.h, rank, world, C.c_void_p(d_upper_ptr), C.c_void_p(stream)))

    def assemble_matrix(self, d_gathered_ptr, world, d_bits_ptr, stream=0):
        capi.check(self.lib.ipc_assemble_matrix(self.h, C.c_void_p(d_gathered_ptr), world,
                                                C.c_void_p(d_bits_ptr), C.c_void_p(stream)))

    def set_max(self, d_bits_ptr, d_accepted_ptr, stream=0):
        capi.check(self.lib.ipc_set_max(self.h, C.c_void_p(d_bits_ptr), C.c_void_p(d_accepted_ptr),
                                        C.c_void_p(stream)))

    # ---- diagnostics ----------------------------------------------------------------------
    def cell_info(self):
        n = C.c_int(0)
        capi.check(self.lib.ipc_cell_count(self.h, C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=capi.CELL_DTYPE)
        capi.check(self.lib.ipc_cell_info(self.h, _p(out), n.value))
        return out[:n.value]

    def solve_report(self):
        """Counts over the cells of the last solve: dict(cells, long_cells, failed_cells, capped_cells, nan_cells)."""
        r = capi.SolveReport()
        capi.check(self.lib.ipc_solve_report(self.h, C.byref(r)))
        return {k: getattr(r, k) for k, _ in capi.SolveReport._fields_}

    def solver_time_ms(self):
        ms, nl = C.c_double(0), C.c_int(0)
        capi.check(self.lib.ipc_solver_time_ms(self.h, C.byref(ms), C.byref(nl)))
        return ms.value, nl.value

    def synchronize(self):
        capi.check(self.lib.ipc_synchronize(self.h))

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.ipc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def unpack_bits(bits, N):
    """[N, words] uint64 -> [N, N] uint8."""
    b = np.ascontiguousarray(bits, dtype=np.uint64)
    u8 = b.view(np.uint8).reshape(b.shape[0], -1)
    return np.unpackbits(u8, axis=1, bitorder="little")[:, :N].astype(np.uint8)
