"""utilities.features — the correlative rotation search of the reference
(/root/reference/utilities/features.py:165-242) with the same signature, scored
on the MI355X.

Only ``rotation_search`` (the default pre-alignment, config.yaml:34) is provided:
about half of every non-IMU scan pair of the reference goes into its ~270
nearest-neighbour sweeps.  Here the whole search is one chain of launches
(voxel filters, means, coarse sweep, arg-min, fine sweep, arg-min) behind
``icpmi_rotation_search``; the angle grids and their cos/sin are computed with
the reference's own NumPy expressions (cached on the device), so the chosen angle,
R and t are the reference's numbers bit for bit.
The RANSAC feature pipeline (``feature_based_alignment``) is outside the
accelerated path (off by default, unseeded in the reference).
"""
import numpy as np
import torch

from icpmi import _lib
from icpmi import batch as _b
from icpmi.prealign import AngleTables, arange_rows, rotation_search_batch, run_icp_pair_batch  # noqa: F401

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


class _SearchContext:
    """Device buffers and angle tables of the rotation searches, kept between calls (one per device): the scans of a
    SLAM loop have the same size every time, and allocating, uploading offsets and reading sizes back per call cost
    several times the kernels themselves."""
    _per_device = {}

    @classmethod
    def get(cls):
        _b.require_gpu()
        dev = torch.device("cuda", torch.cuda.current_device())
        if dev not in cls._per_device:
            cls._per_device[dev] = cls(dev)
        return cls._per_device[dev]

    def __init__(self, dev):
        self.dev = dev
        self.cap = 0
        self.tables = {}
        self.rec = torch.zeros(12, dtype=torch.float64, device=dev)
        self.rec_host = torch.zeros(12, dtype=torch.float64).pin_memory()
        self.ws = None

    def upload(self, src, tgt):
        """source rows then target rows in one pinned staging buffer -> one asynchronous copy."""
        n = len(src) + len(tgt)
        if n > self.cap:
            self.cap = max(2 * n, 8192)
            self.stage = torch.empty((self.cap, 2), dtype=torch.float64).pin_memory()
            self.pts = torch.empty((self.cap, 2), dtype=torch.float64, device=self.dev)
        h = self.stage.numpy()
        h[:len(src)] = src
        h[len(src):n] = tgt
        self.pts[:n].copy_(self.stage[:n], non_blocking=True)
        return self.pts

    def workspace(self, n_src, n_tgt, n_coarse, max_fine):
        need = _lib.lib().icpmi_rotation_search_workspace_bytes(n_src, n_tgt, n_coarse, max_fine)
        if self.ws is None or self.ws.numel() < need:
            self.ws = torch.empty(2 * need, dtype=torch.uint8, device=self.dev)
        return self.ws

    def device_table(self, coarse, fine, fine_n):
        """cos / sin of the coarse angles and of every fine grid (features.py:214 uses np.cos / np.sin) on the device."""
        cs = np.ascontiguousarray(np.stack([np.cos(coarse), np.sin(coarse)], axis=1))
        fcs = np.ascontiguousarray(np.stack([np.cos(fine), np.sin(fine)], axis=2)) if fine.size else np.zeros((len(coarse), 0, 2))
        return (torch.from_numpy(cs).to(self.dev), torch.from_numpy(fcs).to(self.dev),
                torch.from_numpy(np.ascontiguousarray(fine_n, dtype=np.int32)).to(self.dev))

    def run(self, src, tgt, voxel_size, coarse, fine, fine_n, dtab, centred, shift):
        """-> the 12-double record on the host (one synchronisation), see include/icpmi.h icpmi_rotation_search."""
        d_cs, d_fcs, d_fn = dtab
        pts = self.upload(src, tgt)
        max_fine = int(fine.shape[1]) if fine.ndim == 2 else 0
        ws = self.workspace(len(src), len(tgt), len(coarse), max_fine)
        _lib.check(_lib.lib().icpmi_rotation_search(_b._ptr(pts), len(src), len(tgt), float(voxel_size), _b._ptr(d_cs), len(coarse),
                                                    _b._ptr(d_fcs) if max_fine else None, _b._ptr(d_fn) if max_fine else None,
                                                    max_fine, 1 if centred else 0, float(shift[0]), float(shift[1]),
                                                    _b._ptr(self.rec), _b._ptr(ws), ws.numel(), _b._stream()), "rotation_search")
        self.rec_host.copy_(self.rec, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return self.rec_host.numpy().copy()

    def filtered_clouds(self, n_src, n_tgt, rec):
        """Views of the voxel-filtered source and target the last run left in the workspace (device, no copy)."""
        v = self.ws[256:256 + (n_src + n_tgt) * 16].view(torch.float64).reshape(-1, 2)
        return v[:int(rec[0])], v[n_src:n_src + int(rec[1])]


def _as_rows(a, name):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim != 2 or a.shape[1] != 2:
        raise ValueError(f"{name} must have shape (n, 2), got {a.shape}")
    if a.shape[0] == 0:
        # the reference fails inside np.min of voxel_downsample on an empty array (icp.py:119)
        raise ValueError("zero-size array to reduction operation minimum which has no identity")
    return a


def rotation_search(source, target, voxel_size=0.3, angle_step_coarse=2.0, angle_step_fine=0.2):
    """Brute-force rotation search — features.py:165-242.  Returns (R (2,2), t (2,), score).

    One chain of launches on the device (voxel filters, means, coarse sweep, arg-min, fine sweep, arg-min) and one
    12-double read-back; the angle grids, their cos / sin and the final R, t are the reference's NumPy expressions."""
    src, tgt = _as_rows(source, "source"), _as_rows(target, "target")
    ctx = _SearchContext.get()
    tab = AngleTables.get(ctx.dev, angle_step_coarse, angle_step_fine)            # features.py:221, 227-229 for every possible winner
    angles_coarse, fine, fine_n, dtab = tab.coarse, tab.fine, tab.fine_n, tab.device_table
    rec = ctx.run(src, tgt, voxel_size, angles_coarse, fine, fine_n, dtab, True, (0.0, 0.0))
    if rec[0] < 5 or rec[1] < 5:                                               # features.py:203-204
        return np.eye(2), np.zeros(2), float("inf")
    k = int(rec[6])
    if int(rec[8]) <= 0:
        raise ValueError("attempt to get argmin of an empty sequence")         # np.argmin(scores_fine) on an empty grid
    best_angle = fine[k, int(rec[9])]
    best_score = np.float64(rec[10])
    mu_s, mu_t = rec[2:4].copy(), rec[4:6].copy()
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
