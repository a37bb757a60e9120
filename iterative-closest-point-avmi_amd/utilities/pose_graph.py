"""utilities.pose_graph — the reference's 2-D pose-graph optimiser
(/root/reference/utilities/pose_graph.py) with the same names and behaviour; the
Gauss-Newton optimisation runs on the MI355X in one launch (csrc/posegraph.hip).

Nodes are poses [x, y, theta]; edges are relative-pose measurements with 3x3
information matrices.  After ``optimize()`` read ``nodes`` or
``get_poses_as_matrices()``.  ``last_info`` holds what the reference only
prints: iterations run, status and the norm of the last step.  The private
per-edge helper ``_error_and_jacobians`` (pose_graph.py:138-182) has no host
counterpart here: edges are linearised inside the kernels.
"""
import ctypes as C

import numpy as np
import torch

from icpmi import _lib
from icpmi import batch as _b

VERBOSE = True      # the reference prints one line per optimisation

NOTHING_TO_DO, CONVERGED, MAX_ITERATIONS, SINGULAR = 0, 1, 2, 3


def normalize_angle(a):
    """pose_graph.py:15-17: wrap to [-pi, pi)."""
    return (a + np.pi) % (2 * np.pi) - np.pi


def pose_matrix_to_vec(T):
    """pose_graph.py:20-22: 3x3 homogeneous matrix -> [x, y, theta]."""
    return np.array([T[0, 2], T[1, 2], np.arctan2(T[1, 0], T[0, 0])])


def pose_vec_to_matrix(v):
    """pose_graph.py:25-31."""
    x, y, theta = v
    c, s = np.cos(theta), np.sin(theta)
    return np.array([[c, -s, x], [s, c, y], [0, 0, 1]])


def relative_transform_vec(T_i, T_j):
    """pose_graph.py:34-37: z_ij = T_i^-1 T_j as [dx, dy, dtheta]."""
    return pose_matrix_to_vec(np.linalg.inv(T_i) @ T_j)


class PoseGraph2D:
    """pose_graph.py:42-194.  ``nodes`` is a list of (3,) arrays and ``edges`` a list of
    ``(i, j, z_ij, omega)`` tuples, as in the reference; both may be edited freely between calls."""

    def __init__(self):
        self.nodes = []
        self.edges = []
        self.last_info = {}

    def add_node(self, pose_vec):
        self.nodes.append(np.asarray(pose_vec, dtype=float).copy())
        return len(self.nodes) - 1

    def add_edge(self, i, j, measurement, information=None):
        z = np.asarray(measurement, dtype=float).copy()
        omega = np.eye(3) if information is None else np.asarray(information, dtype=float).copy()
        self.edges.append((i, j, z, omega))

    # ── device staging ────────────────────────────────────────────────────
    def _edge_arrays(self):
        m, n = len(self.edges), len(self.nodes)
        ij = np.empty((m, 2), dtype=np.int32)
        z = np.empty((m, 3))
        om = np.empty((m, 3, 3))
        for q, (i, j, zz, o) in enumerate(self.edges):
            i, j = int(i), int(j)
            if i < 0:                    # Python list indexing semantics of the reference
                i += n
            if j < 0:
                j += n
            if not (0 <= i < n and 0 <= j < n):
                raise IndexError("list index out of range")
            ij[q] = (i, j)
            z[q] = zz
            om[q] = o
        return ij, z, om

    def optimize(self, n_iterations=20, fix_node=0, convergence_eps=1e-6):
        """pose_graph.py:83-134: Gauss-Newton with the pose at ``fix_node`` held constant."""
        n = len(self.nodes)
        if n < 2 or len(self.edges) == 0:
            return
        _b.require_gpu()
        L = _lib.lib()
        dev = torch.device("cuda", torch.cuda.current_device())
        ij, z, om = self._edge_arrays()
        m = len(ij)
        fix = fix_node + n if fix_node < 0 else fix_node
        d_nodes = torch.from_numpy(np.ascontiguousarray(np.array(self.nodes, dtype=np.float64).reshape(n, 3))).to(dev)
        d_z, d_om = torch.from_numpy(z).to(dev), torch.from_numpy(om).to(dev)
        info = torch.zeros(10, dtype=torch.float64, device=dev)
        ijp = ij.ctypes.data_as(C.c_void_p)
        need = L.icpmi_pose_graph_workspace_bytes(ijp, n, m)
        ws = torch.empty(max(int(need), 256), dtype=torch.uint8, device=dev)
        _lib.check(L.icpmi_pose_graph_optimize(_b._ptr(d_nodes), ijp, _b._ptr(d_z), _b._ptr(d_om), n, m,
                                               int(n_iterations), int(fix), float(convergence_eps), _b._ptr(info),
                                               _b._ptr(ws), ws.numel(), _b._stream()), "PoseGraph2D.optimize")
        out = d_nodes.cpu().numpy()
        info = info.cpu().numpy()
        iters, status, step = int(info[0]), int(info[1]), info[2]
        for k in range(n):                                   # in place, like pose_graph.py:122-125
            self.nodes[k][:] = out[k]
        self.last_info = dict(iterations=iters, status=status, step_norm=float(step),
                              phase_us=dict(zip(("assemble", "chain_lu", "chain_sweeps", "closure_system",
                                                 "closure_solve", "dx", "apply"), info[3:10].tolist())))
        if VERBOSE:
            if status == SINGULAR:
                print(f"  PoseGraph: singular H at iter {iters}, stopping")
            elif status == CONVERGED:
                print(f"  PoseGraph converged: iter={iters - 1}, ||Δx||={step:.2e}")
            elif status == MAX_ITERATIONS:
                print(f"  PoseGraph max iterations: iter={n_iterations}, ||Δx||={step:.2e}")

    def get_poses_as_matrices(self):
        return [pose_vec_to_matrix(v) for v in self.nodes]

    def total_error(self):
        """pose_graph.py:189-194: sum of e^T Omega e over the edges."""
        if not self.edges:
            return 0.0
        _b.require_gpu()
        dev = torch.device("cuda", torch.cuda.current_device())
        n = len(self.nodes)
        ij, z, om = self._edge_arrays()
        m = len(ij)
        d_nodes = torch.from_numpy(np.ascontiguousarray(np.array(self.nodes, dtype=np.float64).reshape(n, 3))).to(dev)
        d_i = torch.from_numpy(np.ascontiguousarray(ij[:, 0])).to(dev)
        d_j = torch.from_numpy(np.ascontiguousarray(ij[:, 1])).to(dev)
        d_z, d_om = torch.from_numpy(z).to(dev), torch.from_numpy(om).to(dev)
        scratch = torch.empty(m + 1, dtype=torch.float64, device=dev)
        _lib.check(_lib.lib().icpmi_pose_graph_error(_b._ptr(d_nodes), _b._ptr(d_i), _b._ptr(d_j), _b._ptr(d_z), _b._ptr(d_om),
                                                     m, _b._ptr(scratch), _b._ptr(scratch[m:]), _b._stream()),
                   "PoseGraph2D.total_error")
        return float(scratch[m].item())
