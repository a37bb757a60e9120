"""utilities.features — the correlative rotation search of the reference
(/root/reference/utilities/features.py:165-242) with the same signature, scored
on the MI355X.

Only ``rotation_search`` (the default pre-alignment, config.yaml:34) is provided:
about half of every non-IMU scan pair of the reference goes into its ~270
nearest-neighbour sweeps.  Here the voxel filter and every sweep are HIP kernels
(one launch per sweep, all angles at once); the angle grids, cos/sin and the
arg-min stay NumPy so that they are the reference's numbers bit for bit.
The RANSAC feature pipeline (``feature_based_alignment``) is outside the
accelerated path (off by default, unseeded in the reference).
"""
import numpy as np
import torch

from icpmi import _lib
from icpmi import batch as _b
from .icp import voxel_downsample

VERBOSE = True      # the reference prints one line per search


def rotation_scores(src_rows, target, angles, shift):
    """Mean squared NN distance of ``src_rows @ R(a).T + shift`` in ``target`` for every angle (radians).

    The scoring function of features.py:213-218 and slam.py:138-143, all angles in one launch."""
    _b.require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device())

    def on_device(x):        # NumPy rows are uploaded; float64 device tensors (a resident submap) are used in place
        if isinstance(x, torch.Tensor):
            return x.to(dev, torch.float64).contiguous()
        return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).to(dev)

    d_s, d_t = on_device(src_rows), on_device(target)
    a = np.ascontiguousarray(angles, dtype=np.float64).ravel()
    if d_s.dim() != 2 or d_s.shape[1] != 2 or d_t.dim() != 2 or d_t.shape[1] != 2 or len(d_s) == 0 or len(d_t) == 0:
        raise ValueError("rotation_scores needs non-empty (n, 2) arrays")
    if len(a) == 0:
        return np.empty(0)
    cs = np.ascontiguousarray(np.stack([np.cos(a), np.sin(a)], axis=1))       # features.py:214
    d_cs = torch.from_numpy(cs).to(dev)
    out = torch.empty(len(a), dtype=torch.float64, device=dev)
    _lib.check(_lib.lib().icpmi_rotation_scores(_b._ptr(d_s), len(d_s), _b._ptr(d_t), len(d_t), _b._ptr(d_cs), len(a),
                                                float(shift[0]), float(shift[1]), _b._ptr(out), _b._stream()),
               "rotation_search")
    return out.cpu().numpy()


def rotation_search(source, target, voxel_size=0.3, angle_step_coarse=2.0, angle_step_fine=0.2):
    """Brute-force rotation search — features.py:165-242.  Returns (R (2,2), t (2,), score)."""
    src = voxel_downsample(source, voxel_size)                                 # features.py:200-201
    tgt = voxel_downsample(target, voxel_size)
    if len(src) < 5 or len(tgt) < 5:                                           # features.py:203-204
        return np.eye(2), np.zeros(2), float("inf")
    mu_s = src.mean(axis=0)
    mu_t = tgt.mean(axis=0)
    src_c = src - mu_s
    angles_coarse = np.deg2rad(np.arange(-180, 180, angle_step_coarse))        # features.py:221
    scores_coarse = rotation_scores(src_c, tgt, angles_coarse, mu_t)
    best_angle = angles_coarse[int(np.argmin(scores_coarse))]
    lo = best_angle - np.deg2rad(angle_step_coarse)                            # features.py:227-229
    hi = best_angle + np.deg2rad(angle_step_coarse)
    angles_fine = np.arange(lo, hi, np.deg2rad(angle_step_fine))
    scores_fine = rotation_scores(src_c, tgt, angles_fine, mu_t)
    best_f = int(np.argmin(scores_fine))
    best_angle = angles_fine[best_f]
    best_score = scores_fine[best_f]
    ca, sa = np.cos(best_angle), np.sin(best_angle)
    R = np.array([[ca, -sa], [sa, ca]])
    t = mu_t - R @ mu_s
    if VERBOSE:
        print(f"  Rotation search: best angle {np.degrees(best_angle):.1f}°, "
              f"score {best_score:.4f}")
    return R, t, best_score


def feature_based_alignment(*args, **kwargs):
    raise NotImplementedError("feature_based_alignment (features.py:247-315) is outside the accelerated path: "
                              "it is off by default in the reference (config.yaml:34) and uses unseeded RANSAC")
