"""ctypes binding of include/ipc_amd.h (libipc_amd.so).

This is the reference-side binding a Python caller uses; the C++ tester binds the same
symbols directly.  The library is the product path: there is NO CPU fallback -- if the HIP
library is missing or cannot be loaded, importing the engine raises.
"""
import ctypes as C
import os

import numpy as np



def export_recommended_environment():
    """The faithful incremental mode keeps up to 16 cluster solves in flight, one persistent launch per HIP stream; streams
    that share a hardware queue run one after the other and the runtime defaults to 4 queues.  GPU_MAX_HW_QUEUES is read
    once, when the HIP runtime initialises -- so it has to be in the environment before the first HIP call of the process
    (torch's included).  That is the HOST PROGRAM's decision; importing the binding no longer edits the environment (round 5).
    __graft_entry__.smoke() and tools/late_state_dump.py call this function; bench.py, tests/conftest.py and most tools set the
    variable themselves (os.environ.setdefault at their top); a Python caller of ipc_amd.consensus.IPC that does neither runs
    the faithful mode on 4 hardware queues -- the library says so once on stderr (spec_ensure)."""
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")


_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IPC_AMD_LIB") or os.path.join(_HERE, "libipc_amd.so")   # (IPC_AMD_LIB: A/B builds, tools/)

# every symbol include/ipc_amd.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "ipc_last_error", "ipc_create", "ipc_destroy", "ipc_set_candidates", "ipc_candidate_order",
    "ipc_initial_poses", "ipc_rows_per_rank", "ipc_solve_rows", "ipc_assemble_matrix", "ipc_set_max",
    "ipc_run", "ipc_cell_count", "ipc_cell_info", "ipc_solve_report", "ipc_solver_time_ms", "ipc_synchronize",
    "ipc_incremental_reset", "ipc_incremental_prepare", "ipc_agreement_check", "ipc_consensus_size", "ipc_consensus_set",
    "ipc_remove_from_consensus", "ipc_add_to_consensus", "ipc_current_poses", "ipc_final_optimize",
    "ipc_debug_dense_solve", "ipc_debug_band_solve", "ipc_debug_band_plan", "ipc_debug_absorbed_edges", "ipc_append_candidate", "ipc_incremental_set_state", "ipc_incremental_counters", "ipc_row_assignment", "ipc_run_sharded", "ipc_run_set_only",
]


class Params(C.Structure):
    _fields_ = [("fast_reject_th", C.c_double), ("fast_reject_iter_base", C.c_int),
                ("slow_reject_th", C.c_double), ("slow_reject_iter_base", C.c_int),
                ("s_factor", C.c_double)]


class CellInfo(C.Structure):
    _fields_ = [("i", C.c_int), ("j", C.c_int), ("lo", C.c_int), ("hi", C.c_int),
                ("max_chi2", C.c_double), ("chi2_total", C.c_double),
                ("iterations", C.c_int), ("tries", C.c_int), ("flags", C.c_int), ("evals", C.c_int)]


class CheckInfo(C.Structure):
    _fields_ = [("lo", C.c_int), ("hi", C.c_int), ("n_cluster_loops", C.c_int), ("iterations", C.c_int),
                ("tries", C.c_int), ("flags", C.c_int), ("max_chi2", C.c_double), ("chi2_total", C.c_double),
                ("chi2_initial", C.c_double)]


class IncrementalCounters(C.Structure):
    _fields_ = [("host_solver_fallbacks", C.c_long), ("lost_launches", C.c_long), ("relaunches", C.c_long),
                ("literal_band_solves", C.c_long)]


class SolveReport(C.Structure):
    _fields_ = [("cells", C.c_int), ("long_cells", C.c_int), ("failed_cells", C.c_int), ("capped_cells", C.c_int),
                ("nan_cells", C.c_int), ("damped_cells", C.c_int), ("literal_cells", C.c_int)]


CELL_DTYPE = np.dtype([("i", "<i4"), ("j", "<i4"), ("lo", "<i4"), ("hi", "<i4"), ("max_chi2", "<f8"),
                       ("chi2_total", "<f8"), ("iterations", "<i4"), ("tries", "<i4"), ("flags", "<i4"),
                       ("evals", "<i4")])

_lib = None


class IpcError(RuntimeError):
    pass


def load():
    """Loads libipc_amd.so; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IpcError("libipc_amd.so is not built (%s); run `python -c 'import __graft_entry__ as g; "
                       "g.build()'` -- there is no CPU fallback" % LIB_PATH)
    # One HIP runtime per process: PyTorch ships its own libamdhip64; with this library loaded
    # first the process binds /opt/rocm's copy and a later `import torch` finds no GPU.  Loading
    # torch's runtime first (when torch is there) makes both share it.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    lib.ipc_last_error.restype = C.c_char_p
    vp, ip, dp = C.c_void_p, C.c_int, C.c_double
    lib.ipc_create.argtypes = [ip, ip, vp, vp, C.POINTER(Params), ip, C.POINTER(vp)]
    lib.ipc_destroy.argtypes = [vp]
    lib.ipc_set_candidates.argtypes = [vp, ip, vp, vp, vp]
    lib.ipc_append_candidate.argtypes = [vp, vp, vp, vp, C.POINTER(ip)]
    lib.ipc_candidate_order.argtypes = [vp, vp]
    lib.ipc_initial_poses.argtypes = [vp, vp]
    lib.ipc_rows_per_rank.argtypes = [ip, ip]
    lib.ipc_row_assignment.argtypes = [ip, vp, ip, ip, vp]
    lib.ipc_solve_rows.argtypes = [vp, ip, ip, vp, vp]
    lib.ipc_assemble_matrix.argtypes = [vp, vp, ip, vp, vp]
    lib.ipc_set_max.argtypes = [vp, vp, vp, vp]
    lib.ipc_run.argtypes = [vp, vp, vp]
    lib.ipc_run_sharded.argtypes = [C.POINTER(vp), ip, vp, vp]
    lib.ipc_run_set_only.argtypes = [vp, vp, C.POINTER(ip)]
    lib.ipc_cell_count.argtypes = [vp, C.POINTER(ip)]
    lib.ipc_cell_info.argtypes = [vp, vp, ip]
    lib.ipc_solve_report.argtypes = [vp, C.POINTER(SolveReport)]
    lib.ipc_solver_time_ms.argtypes = [vp, C.POINTER(dp), C.POINTER(ip)]
    lib.ipc_synchronize.argtypes = [vp]
    lib.ipc_incremental_reset.argtypes = [vp]
    lib.ipc_incremental_prepare.argtypes = [vp]
    lib.ipc_agreement_check.argtypes = [vp, ip, C.POINTER(ip), C.POINTER(CheckInfo)]
    lib.ipc_consensus_size.argtypes = [vp, C.POINTER(ip)]
    lib.ipc_consensus_set.argtypes = [vp, vp]
    lib.ipc_remove_from_consensus.argtypes = [vp, ip, C.POINTER(ip)]
    lib.ipc_add_to_consensus.argtypes = [vp, ip]
    lib.ipc_current_poses.argtypes = [vp, vp]
    lib.ipc_incremental_set_state.argtypes = [vp, vp, vp, ip, ip]
    lib.ipc_incremental_counters.argtypes = [vp, C.POINTER(IncrementalCounters)]
    lib.ipc_final_optimize.argtypes = [vp, vp, ip, vp, C.POINTER(CheckInfo)]
    lib.ipc_debug_dense_solve.argtypes = [ip, vp, ip, ip, vp, C.POINTER(ip)]
    lib.ipc_debug_band_solve.argtypes = [ip, ip, ip, vp, ip, vp, C.POINTER(ip)]
    lib.ipc_debug_absorbed_edges.argtypes = [ip, ip, ip, vp, vp, ip, vp, C.POINTER(ip), C.POINTER(ip), C.POINTER(ip)]
    lib.ipc_debug_band_plan.argtypes = [ip, ip, vp, vp, ip, C.POINTER(ip), C.POINTER(ip), C.POINTER(ip), vp]
    assert C.sizeof(CellInfo) == CELL_DTYPE.itemsize
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise IpcError("ipc_amd error %d: %s" % (rc, load().ipc_last_error().decode()))
