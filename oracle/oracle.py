"""ctypes wrapper of oracle/liboracle.so -- CPU ORACLE, TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; nothing under ipc_amd/ does.  See the header of oracle/ipc_oracle.c for what is
restated and why parity is "unpinned" at the g2o boundary.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "ipc_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


class Stats(C.Structure):
    _fields_ = [("iterations", C.c_int), ("tries_total", C.c_int), ("terminated", C.c_int),
                ("chi2_initial", C.c_double), ("chi2_final", C.c_double)]


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.oracle_normalize_theta.restype = C.c_double
        _LIB.oracle_normalize_theta.argtypes = [C.c_double]
        _LIB.oracle_solve_cell.restype = C.c_double
        _LIB.oracle_pair_cell.restype = C.c_double
        _LIB.oracle_ipc_create.restype = C.c_void_p
        assert _LIB.oracle_stats_size() == C.sizeof(Stats)
    return _LIB


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def pose_size(dim):
    return 3 if dim == 2 else 12


def tan_dim(dim):
    return 3 if dim == 2 else 6


def meas_size(dim):
    return 3 if dim == 2 else 7


def info_size(dim):
    return 6 if dim == 2 else 21


def set_wide_dots(on):
    """Envelope factorisation with eight partial sums per dot product (rounding-level different, several times faster);
    off by default -- every committed fixture except the late-state ones was made with the serial sums."""
    lib().oracle_set_wide_dots(int(bool(on)))


def normalize_theta(t):
    return lib().oracle_normalize_theta(float(t))


def meas_to_pose(dim, m):
    m = _d(m)
    X = np.zeros(pose_size(dim))
    lib().oracle_meas_to_pose(dim, _p(m), _p(X))
    return X


def pose_mul(dim, a, b):
    a, b = _d(a), _d(b)
    r = np.zeros(pose_size(dim))
    lib().oracle_pose_mul(dim, _p(a), _p(b), _p(r))
    return r


def pose_inv(dim, a):
    a = _d(a)
    r = np.zeros(pose_size(dim))
    lib().oracle_pose_inv(dim, _p(a), _p(r))
    return r


def pose_oplus(dim, X, delta):
    X = _d(X).copy()
    delta = _d(delta)
    lib().oracle_pose_oplus(dim, _p(X), _p(delta))
    return X


def edge_error(dim, Z, Xi, Xj):
    Zinv = pose_inv(dim, Z)
    Xi, Xj = _d(Xi), _d(Xj)
    e = np.zeros(tan_dim(dim))
    lib().oracle_edge_error(dim, _p(Zinv), _p(Xi), _p(Xj), _p(e))
    return e


def edge_jacobians(dim, Z, Xi, Xj):
    Z = _d(Z)
    Zinv = pose_inv(dim, Z)
    Xi, Xj = _d(Xi), _d(Xj)
    d = tan_dim(dim)
    A = np.zeros((d, d))
    B = np.zeros((d, d))
    lib().oracle_edge_jacobians(dim, _p(Z), _p(Zinv), _p(Xi), _p(Xj), _p(A), _p(B))
    return A, B


def propagate(dim, odom_meas):
    odom_meas = _d(odom_meas)
    V = odom_meas.shape[0] + 1
    poses = np.zeros((V, pose_size(dim)))
    lib().oracle_propagate(dim, V, _p(odom_meas), _p(poses))
    return poses


def solve_cell(dim, odom_meas, odom_info, s_factor, poses, lo, hi, loop_ids, loop_meas, loop_info,
               iter_base, want_poses=False):
    """isAgreeingWithCurrentState on chain [lo,hi] + the given loop edges.
    Returns dict(max_chi2, chi2 (L+nl), stats, poses?)."""
    odom_meas, odom_info, poses = _d(odom_meas), _d(odom_info), _d(poses)
    loop_ids = _i(loop_ids).reshape(-1, 2)
    nl = loop_ids.shape[0]
    loop_meas = _d(loop_meas).reshape(nl, meas_size(dim))
    loop_info = _d(loop_info).reshape(nl, info_size(dim))
    L = hi - lo
    chi2 = np.zeros(L + nl)
    pout = np.zeros((L + 1, pose_size(dim)))
    st = Stats()
    mx = lib().oracle_solve_cell(dim, _p(odom_meas), _p(odom_info), C.c_double(s_factor), _p(poses),
                                 int(lo), int(hi), int(nl), _p(loop_ids), _p(loop_meas), _p(loop_info),
                                 int(iter_base), _p(chi2), _p(pout), C.byref(st))
    out = dict(max_chi2=mx, chi2=chi2, iterations=st.iterations, tries=st.tries_total,
               terminated=st.terminated, chi2_initial=st.chi2_initial, chi2_final=st.chi2_final)
    if want_poses:
        out["poses"] = pout
    return out


def candidate_order(ids):
    ids = _i(ids).reshape(-1, 2)
    order = np.zeros(ids.shape[0], dtype=np.int32)
    lib().oracle_candidate_order(ids.shape[0], _p(ids), _p(order))
    return order


def pair_cell(dim, odom_meas, odom_info, s_factor, poses, ids, meas, info, i, j, fast_iter, slow_iter):
    """Returns (solved, max_chi2, iterations)."""
    odom_meas, odom_info, poses = _d(odom_meas), _d(odom_info), _d(poses)
    ids, meas, info = _i(ids), _d(meas), _d(info)
    solved = C.c_int(0)
    st = Stats()
    mx = lib().oracle_pair_cell(dim, _p(odom_meas), _p(odom_info), C.c_double(s_factor), _p(poses),
                                _p(ids), _p(meas), _p(info), int(i), int(j), int(fast_iter),
                                int(slow_iter), C.byref(solved), C.byref(st))
    return bool(solved.value), mx, st.iterations


def pair_cells_mt(dim, odom_meas, odom_info, s_factor, poses, ids, meas, info, ci, cj, fast_iter, slow_iter,
                  nthreads):
    """Batch of pair cells over `nthreads` POSIX threads (static partition).  Returns
    (max_chi2 [n] with NaN where the pair does not overlap, iterations [n], threads used)."""
    odom_meas, odom_info, poses = _d(odom_meas), _d(odom_info), _d(poses)
    ids, meas, info = _i(ids), _d(meas), _d(info)
    ci, cj = _i(ci), _i(cj)
    n = ci.shape[0]
    mx = np.zeros(n)
    its = np.zeros(n, dtype=np.int32)
    used = lib().oracle_pair_cells_mt(dim, _p(odom_meas), _p(odom_info), C.c_double(s_factor), _p(poses), _p(ids),
                                      _p(meas), _p(info), int(fast_iter), int(slow_iter), n, _p(ci), _p(cj),
                                      int(nthreads), _p(mx), _p(its))
    return mx, its, used


def consistency_matrix(dim, odom_meas, odom_info, s_factor, ids, meas, info, fast_th, fast_iter,
                       slow_th, slow_iter):
    """Returns (okmat uint8 N x N, maxchi2 N x N with NaN on non-overlapping cells)."""
    odom_meas, odom_info = _d(odom_meas), _d(odom_info)
    ids, meas, info = _i(ids).reshape(-1, 2), _d(meas), _d(info)
    N = ids.shape[0]
    V = odom_meas.shape[0] + 1
    mx = np.zeros((N, N))
    ok = np.zeros((N, N), dtype=np.uint8)
    lib().oracle_consistency_matrix(dim, V, _p(odom_meas), _p(odom_info), C.c_double(s_factor), N,
                                    _p(ids), _p(meas), _p(info), C.c_double(fast_th), int(fast_iter),
                                    C.c_double(slow_th), int(slow_iter), _p(mx), _p(ok))
    return ok, mx


def set_max(okmat, order):
    okmat = np.ascontiguousarray(okmat, dtype=np.uint8)
    order = _i(order)
    N = okmat.shape[0]
    acc = np.zeros(N, dtype=np.uint8)
    lib().oracle_set_max(N, _p(okmat), _p(order), _p(acc))
    return acc


class IncrementalIPC:
    """Faithful incremental IPC (reference src/consensus.cpp) on the CPU oracle."""

    def __init__(self, dim, odom_meas, odom_info, s_factor, fast_th, fast_iter, slow_th, slow_iter,
                 ids, meas, info):
        self.dim = dim
        odom_meas, odom_info = _d(odom_meas), _d(odom_info)
        ids, meas, info = _i(ids).reshape(-1, 2), _d(meas), _d(info)
        self.V = odom_meas.shape[0] + 1
        self.N = ids.shape[0]
        self.ids = ids
        self.h = C.c_void_p(lib().oracle_ipc_create(
            dim, self.V, _p(odom_meas), _p(odom_info), C.c_double(s_factor), C.c_double(fast_th),
            int(fast_iter), C.c_double(slow_th), int(slow_iter), self.N, _p(ids), _p(meas), _p(info)))

    def agreement_check(self, k):
        info = np.zeros(4, dtype=np.int32)
        mx = C.c_double(0)
        r = lib().oracle_ipc_agreement_check(self.h, int(k), _p(info), C.byref(mx))
        return bool(r), dict(lo=int(info[0]), hi=int(info[1]), cluster=int(info[2]),
                             iterations=int(info[3]), max_chi2=mx.value)

    def set_state(self, poses, consensus):
        """Continue from a saved state: vertex estimates [V, 3 | 12] + consensus set (candidate indices in set order)."""
        poses = _d(poses)
        assert poses.shape == (self.V, pose_size(self.dim))
        cns = _i(consensus)
        lib().oracle_ipc_set_state(self.h, _p(poses), _p(cns), int(cns.shape[0]))

    def consensus(self):
        n = lib().oracle_ipc_consensus_size(self.h)
        out = np.zeros(max(n, 1), dtype=np.int32)
        lib().oracle_ipc_consensus(self.h, _p(out))
        return out[:n]

    def poses(self):
        out = np.zeros((self.V, pose_size(self.dim)))
        lib().oracle_ipc_poses(self.h, _p(out))
        return out

    def run(self):
        order = candidate_order(self.ids)
        acc = np.zeros(self.N, dtype=np.uint8)
        for k in order:
            ok, _ = self.agreement_check(k)
            acc[k] = ok
        return acc

    def __del__(self):
        try:
            lib().oracle_ipc_destroy(self.h)
        except Exception:
            pass
