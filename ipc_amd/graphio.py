"""g2o text pose-graph IO and the odometry / loop-candidate split.

Mirrors what the reference does between the file and the hot path:
  * g2o text format tags VERTEX_SE2 / EDGE_SE2 / VERTEX_SE3:QUAT / EDGE_SE3:QUAT
    (loaded by optimizer.load, reference src/utils.cpp:114);
  * splitProblemConstraints (reference src/utils.cpp:172-189): |id1 - id0| > 1 => loop
    candidate, else odometry, both kept in file order;
  * IPC::IPC (reference src/consensus.cpp:13-15): odometry sorted by vertices()[1] id, with
    the implicit contract that odometry edge j joins j -> j+1 and vertex ids are 0..V-1.

Arrays use the file's own layout: measurements (x y theta) / (x y z qx qy qz qw), information
as the upper triangle in row order (6 / 21 numbers).
"""
from dataclasses import dataclass, field

import numpy as np


def meas_size(dim):
    return 3 if dim == 2 else 7


def info_size(dim):
    return 6 if dim == 2 else 21


@dataclass
class PoseGraph:
    dim: int                       # 2 (SE2) or 3 (SE3)
    vertices: np.ndarray           # [V, 3] (x y th) or [V, 7] (x y z qx qy qz qw), file estimates
    odom_meas: np.ndarray          # [V-1, ms]  edge j joins j -> j+1
    odom_info: np.ndarray          # [V-1, is]  un-scaled (s_factor is applied by the engine)
    loop_ids: np.ndarray           # [N, 2] int32 (from, to) in file order
    loop_meas: np.ndarray          # [N, ms]
    loop_info: np.ndarray          # [N, is]
    meta: dict = field(default_factory=dict)

    @property
    def V(self):
        return self.odom_meas.shape[0] + 1

    @property
    def N(self):
        return self.loop_ids.shape[0]

    def subset(self, idx):
        idx = np.asarray(idx)
        return PoseGraph(self.dim, self.vertices, self.odom_meas, self.odom_info,
                         self.loop_ids[idx].copy(), self.loop_meas[idx].copy(),
                         self.loop_info[idx].copy(), dict(self.meta))


def _fmt(x):
    return repr(float(x))


def write_g2o(path, g: PoseGraph):
    """Vertices, then odometry edges, then loop edges (all in array order)."""
    vt = "VERTEX_SE2" if g.dim == 2 else "VERTEX_SE3:QUAT"
    et = "EDGE_SE2" if g.dim == 2 else "EDGE_SE3:QUAT"
    with open(path, "w") as f:
        for i, v in enumerate(g.vertices):
            f.write(vt + " " + str(i) + " " + " ".join(_fmt(x) for x in v) + "\n")
        for j in range(g.V - 1):
            f.write(et + " %d %d " % (j, j + 1) + " ".join(_fmt(x) for x in g.odom_meas[j]) + " "
                    + " ".join(_fmt(x) for x in g.odom_info[j]) + "\n")
        for k in range(g.N):
            f.write(et + " %d %d " % (g.loop_ids[k, 0], g.loop_ids[k, 1])
                    + " ".join(_fmt(x) for x in g.loop_meas[k]) + " "
                    + " ".join(_fmt(x) for x in g.loop_info[k]) + "\n")


def read_g2o(path) -> PoseGraph:
    dim = None
    verts = {}
    odom = {}
    loops = []
    with open(path) as f:
        for line in f:
            tok = line.split()
            if not tok:
                continue
            tag = tok[0]
            if tag in ("VERTEX_SE2", "VERTEX_SE3:QUAT"):
                d = 2 if tag == "VERTEX_SE2" else 3
                if dim is None:
                    dim = d
                if d != dim:
                    raise ValueError("mixed 2D/3D vertices in " + path)
                verts[int(tok[1])] = [float(x) for x in tok[2:2 + meas_size(d)]]
            elif tag in ("EDGE_SE2", "EDGE_SE3:QUAT"):
                d = 2 if tag == "EDGE_SE2" else 3
                if dim is None:
                    dim = d
                if d != dim:
                    raise ValueError("mixed 2D/3D edges in " + path)
                ms, isz = meas_size(d), info_size(d)
                a, b = int(tok[1]), int(tok[2])
                vals = [float(x) for x in tok[3:3 + ms + isz]]
                if len(vals) != ms + isz:
                    raise ValueError("short edge line: " + line)
                if abs(b - a) > 1:                       # utils.cpp:184-186
                    loops.append((a, b, vals[:ms], vals[ms:]))
                else:
                    if b != a + 1:
                        raise ValueError("odometry edge %d->%d is not oriented i -> i+1 "
                                         "(out of the reference's contract)" % (a, b))
                    if b in odom:
                        raise ValueError("duplicate odometry edge into vertex %d" % b)
                    odom[b] = (vals[:ms], vals[ms:])     # keyed by vertices()[1] id (cmpEdgesID)
    if dim is None:
        raise ValueError("no SE2/SE3 vertices or edges in " + path)
    V = len(verts)
    if sorted(verts) != list(range(V)):
        raise ValueError("vertex ids are not exactly 0..V-1 (out of the reference's contract)")
    if sorted(odom) != list(range(1, V)):
        raise ValueError("odometry chain is not exactly one edge per consecutive vertex pair")
    ms, isz = meas_size(dim), info_size(dim)
    g = PoseGraph(
        dim=dim,
        vertices=np.array([verts[i] for i in range(V)], dtype=np.float64).reshape(V, ms),
        odom_meas=np.array([odom[i][0] for i in range(1, V)], dtype=np.float64).reshape(V - 1, ms),
        odom_info=np.array([odom[i][1] for i in range(1, V)], dtype=np.float64).reshape(V - 1, isz),
        loop_ids=np.array([[a, b] for a, b, _, _ in loops], dtype=np.int32).reshape(len(loops), 2),
        loop_meas=np.array([m for _, _, m, _ in loops], dtype=np.float64).reshape(len(loops), ms),
        loop_info=np.array([i for _, _, _, i in loops], dtype=np.float64).reshape(len(loops), isz),
    )
    return g


def candidate_order(loop_ids):
    """cmpTime order (reference src/utils.cpp:379-390, used at src/simulation.cpp:26): ascending
    max(id0, id1).  std::sort leaves ties unspecified; this build fixes (max id, file index)."""
    ids = np.asarray(loop_ids).reshape(-1, 2)
    key = ids.max(axis=1)
    return np.lexsort((np.arange(ids.shape[0]), key)).astype(np.int32)
