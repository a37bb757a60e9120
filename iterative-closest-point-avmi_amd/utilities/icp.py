"""utilities.icp — same function signatures as the reference module
(/root/reference/utilities/icp.py), computed on the MI355X.

NumPy arrays in, NumPy arrays out, inputs never modified; all arithmetic runs
in hand-written HIP kernels behind the C ABI of include/icpmi.h.  There is no
CPU path: without a GPU or without libicpmi.so every function raises.
"""
import numpy as np
import torch

from icpmi import _lib
from icpmi import batch as _b

# Iteration count, status and last delta of the most recent ICP() call.  The
# reference only prints them (icp.py:218,222); the signature stays unchanged and
# callers that need "iterations per second" read them here.
last_icp_info = {}
VERBOSE = True      # the reference prints one line per ICP call


def _as_points(a, name):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim != 2 or a.shape[1] not in (2, 3):
        raise ValueError(f"{name} must have shape (n, 2) or (n, 3), got {a.shape}")
    return a


def center_of_mass(points):
    """icp.py:32-33."""
    return np.array(np.mean(points, axis=0))


def _nn(source, target):
    source, target = _as_points(source, "source"), _as_points(target, "target")
    if len(target) == 0:
        raise ValueError("target is empty")
    cs = _b.CloudSet.from_numpy([source, target])
    dist, idx = _b.nn_set(cs, [0], [1])
    n = len(source)
    return dist[0, :n].cpu().numpy(), idx[0, :n].cpu().numpy().astype(np.int64)


def find_nearest_neighbors(source, target, tree=None):
    """icp.py:35-39.  ``tree`` (a prebuilt KDTree in the reference) is accepted and ignored."""
    _, idx = _nn(source, target)
    return np.asarray(target)[idx]


def find_nearest_neighbor_indices(source, target, tree=None):
    """icp.py:41-46: index of the nearest target point for every source point."""
    return _nn(source, target)[1]


def nearest_neighbors(source, target):
    """(distances, indices) like ``KDTree(target).query(source)`` (icp.py:179)."""
    return _nn(source, target)


def estimate_normals_2d(points, k=10):
    """icp.py:51-76: unit normals from the PCA of the k nearest neighbours."""
    points = _as_points(points, "points")
    if points.shape[1] != 2:
        raise ValueError("estimate_normals_2d needs (n, 2) points")
    if len(points) == 0:
        return np.zeros_like(points)
    cs = _b.CloudSet.from_numpy([points])
    return _b.normals_set(cs, k)[:len(points)].cpu().numpy()


def _point_to_line_solve_2d(source_pts, target_pts, target_normals, nn_indices):
    """icp.py:79-115: one linearised point-to-line step -> (R 2x2, t 2)."""
    _b.require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device())
    s = torch.from_numpy(np.ascontiguousarray(source_pts, dtype=np.float64)).to(dev)
    t = torch.from_numpy(np.ascontiguousarray(target_pts, dtype=np.float64)).to(dev)
    n = torch.from_numpy(np.ascontiguousarray(target_normals, dtype=np.float64)).to(dev)
    i = torch.from_numpy(np.ascontiguousarray(nn_indices, dtype=np.int32)).to(dev)
    if len(i) != len(s):
        raise ValueError("nn_indices and source_pts differ in length")
    if len(i) and (int(i.max()) >= len(t) or int(i.min()) < 0):
        raise IndexError("nn_indices out of range")
    out = torch.empty(6, dtype=torch.float64, device=dev)
    _lib.check(_lib.lib().icpmi_p2l_solve_2d(_b._ptr(s), len(s), _b._ptr(t), _b._ptr(n), _b._ptr(i), _b._ptr(out),
                                             _b._stream()), "_point_to_line_solve_2d")
    o = out.cpu().numpy()
    return o[:4].reshape(2, 2).copy(), o[4:].copy()


def voxel_downsample(points, voxel_size):
    """icp.py:117-129: per-voxel mean, rows ordered by voxel key."""
    points = np.asarray(points, dtype=np.float64)
    if points.ndim != 2:
        raise ValueError("points must be 2-D")
    if points.shape[0] == 0:
        # the reference fails inside np.min on an empty array (icp.py:119)
        raise ValueError("zero-size array to reduction operation minimum which has no identity")
    points = _as_points(points, "points")
    cs = _b.CloudSet.from_numpy([points])
    out = _b.voxel_downsample_set(cs, voxel_size)
    n = int(out.cnt[0].item())
    if n < 0:
        raise OverflowError("voxel key range does not fit in 64 bits")
    return out.pts[:n].cpu().numpy()


def ICP(source, target, error_threshold, max_iterations, voxel_size,
        R_init=None, t_init=None, method="point_to_point", normal_k=10,
        max_corr_dist=None):
    """Iterative Closest Point — icp.py:132-223.  Returns (r_total, t_total, error).

    method: "point_to_point" or "point_to_line" (2-D only; 3-D falls back to
    point-to-point, icp.py:162).  R_init is used only together with t_init
    (icp.py:153).  max_corr_dist=None keeps every correspondence.
    """
    source, target = _as_points(source, "source"), _as_points(target, "target")
    if source.shape[1] != target.shape[1]:
        raise ValueError("source and target differ in dimensionality")
    if len(source) == 0 or len(target) == 0:
        raise ValueError("zero-size array to reduction operation minimum which has no identity")
    R, t, err, info = _b.icp_pair(source, target, error_threshold, max_iterations, voxel_size,
                                  R_init, t_init, method, normal_k, max_corr_dist)
    return R[0], t[0], _report(err[0], info, 0, max_iterations)


def _report(err, info, i, max_iterations):
    """The outcome line of icp.py:218,222 for pair i of a batch result; fills ``last_icp_info``; returns the error."""
    global last_icp_info
    status, iters, delta = int(info["status"][i]), int(info["iters"][i]), float(info["delta"][i])
    last_icp_info = dict(iterations=iters, status=status, delta=delta)
    error = float(err) if np.isinf(err) else np.float64(err)
    if VERBOSE:
        if status == _lib.ST_CONVERGED:
            print(f"  ICP converged: iter={iters - 1}, error={error:.8f}, delta={delta:.2e}")
        else:
            print(f"  ICP max iterations reached: iter={max_iterations}, error={error:.8f}")
    return error


def run_icp(scan_stream, num_scans=None, error_threshold=1e-5, max_iterations=100, voxel_size=0.5):
    """Legacy odometry loop of icp.py:225-250: chains ICP over consecutive scans into 4x4 poses."""
    pose = np.eye(4)
    trajectory, prev, done = [], None, 0
    for _, points in scan_stream:
        if prev is None:
            prev = points
            continue
        r, t, error = ICP(prev, points, error_threshold=error_threshold, max_iterations=max_iterations,
                          voxel_size=voxel_size)
        d = r.shape[0]
        Rn = pose[:d, :d] @ r.T
        pose[:d, :d] = Rn
        pose[:d, 3] = pose[:d, 3] - Rn @ t
        trajectory.append(pose.copy())
        prev = points
        done += 1
        if num_scans is not None and done >= num_scans:
            break
        print("Scan: ", done, "Error: ", error)
    return pose, trajectory
