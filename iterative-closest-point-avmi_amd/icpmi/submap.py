"""Device-resident rolling submap.

The reference keeps the last ``submap_size`` global-frame scans in a Python list
(``slam.py:559-562``: append, pop the oldest beyond the window), and on EVERY
scan stacks them and voxel-filters the stack (``_build_submap``,
``slam.py:103-108`` — ~82 k points in, 63 ms of NumPy) to get the target of the
scan-to-submap ICP (``slam.py:217-225``).  Here the scans stay in HBM once
pushed; ``build()`` concatenates them on the device in buffer order and runs the
voxel filter kernels (bit-identical rows and order to ``np.vstack`` +
``voxel_downsample``), and ``icp()`` registers a scan against the result
without the submap ever visiting the host.

``submap_rotation_search`` / ``attempt_submap_icp`` are the reference's
``_submap_rotation_search`` / ``_attempt_submap_icp`` (``slam.py:111-225``) on
the same kernels; both take the submap as a NumPy array or as the device tensor
``RollingSubmap.build()`` returns.
"""
import numpy as np
import torch

from . import batch as _b

VERBOSE = True      # the reference prints a line for corrections above one degree


class RollingSubmap:
    def __init__(self, window=40, voxel_size=0.04):
        _b.require_gpu()
        if window < 1:
            raise ValueError("window must be at least 1")
        self.window = int(window)
        self.voxel_size = float(voxel_size)
        self._dev = torch.device("cuda", torch.cuda.current_device())
        self._scans = []                # device tensors (n_i, 2) float64, oldest first
        self._built = None              # cached CloudSet of the filtered submap
        self._ws = None

    def __len__(self):
        return len(self._scans)

    def _to_dev(self, pts):
        if isinstance(pts, torch.Tensor):
            t = pts.to(self._dev, torch.float64)
        else:
            t = torch.from_numpy(np.ascontiguousarray(pts, dtype=np.float64)).to(self._dev)
        if t.dim() != 2 or t.shape[1] != 2:
            raise ValueError("scans must have shape (n, 2)")
        return t.contiguous()

    def push(self, global_points):
        """slam.py:559-562: append the scan (global frame); drop the oldest beyond the window."""
        self._scans.append(self._to_dev(global_points))
        if len(self._scans) > self.window:
            self._scans.pop(0)
        self._built = None

    def reset(self, scans=()):
        """slam.py:612-615: rebuild the buffer from (the tail of) a list of global-frame scans."""
        self._scans = [self._to_dev(s) for s in list(scans)[-self.window:]]
        self._built = None

    @property
    def points_in(self):
        return int(sum(s.shape[0] for s in self._scans))

    def build(self):
        """_build_submap (slam.py:103-108) -> (points (m, 2) device tensor view, CloudSet). Empty buffer -> (0, 2)."""
        if self._built is None:
            if not self._scans or self.points_in == 0:
                return torch.empty((0, 2), dtype=torch.float64, device=self._dev), None
            stacked = torch.cat(self._scans, dim=0)                     # np.vstack order: oldest scan first
            cs = _b.CloudSet(stacked, np.array([0, stacked.shape[0]], dtype=np.int32))
            need = _b._lib.lib().icpmi_voxel_workspace_bytes(cs.max_n)
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(need, dtype=torch.uint8, device=self._dev)
            out = _b.voxel_downsample_set(cs, self.voxel_size, workspace=self._ws)
            self._built = out
        m = int(self._built.cnt[0].item())
        return self._built.pts[:m], self._built

    def build_numpy(self):
        pts, _ = self.build()
        return pts.cpu().numpy() if pts.numel() else np.empty((0, 2))

    def icp(self, source_local, error_threshold, max_iterations, voxel_size, R_init=None, t_init=None,
            method="point_to_point", normal_k=10, max_corr_dist=None):
        """ICP(source_local, submap, ...) with the call shape of slam.py:217-225; the submap stays on the device.

        Returns (R, t, error, info) like ``icpmi.batch.icp_batch`` for one pair."""
        sub, _ = self.build()
        if sub.shape[0] == 0:
            raise ValueError("the submap is empty")
        src = self._to_dev(source_local)
        pts = torch.cat([src, sub], dim=0)
        cs = _b.CloudSet(pts, np.array([0, src.shape[0], src.shape[0] + sub.shape[0]], dtype=np.int32))
        b = _b.IcpBatch(cs, [0], [1], error_threshold, max_iterations, voxel_size, R_init, t_init, method, normal_k,
                        max_corr_dist)
        b.run()
        R, t, err, info = b.unpack()
        return R[0], t[0], err[0], {k: v[0] for k, v in info.items()}

    def rotation_search(self, source_local, predicted_pose, **kw):
        """submap_rotation_search against the resident submap."""
        sub, _ = self.build()
        return submap_rotation_search(source_local, sub, predicted_pose, **kw)

    def attempt_icp(self, source_local, predicted_pose, imu_yaw, imu_narrow, sub_rot_range, sub_rot_step, sub_rot_fine,
                    sub_rot_voxel, icp_cfg, sub_corr_dist):
        """attempt_submap_icp against the resident submap (the call of slam.py:505-510)."""
        sub, _ = self.build()
        return attempt_submap_icp(source_local, sub, predicted_pose, imu_yaw, imu_narrow, sub_rot_range, sub_rot_step,
                                  sub_rot_fine, sub_rot_voxel, icp_cfg, sub_corr_dist)


# ── slam.py:111-225 ──────────────────────────────────────────────────────────
def _device_rows(points):
    dev = torch.device("cuda", torch.cuda.current_device())
    if isinstance(points, torch.Tensor):
        t = points.to(dev, torch.float64)
    else:
        t = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float64)).to(dev)
    if t.dim() != 2 or t.shape[1] != 2:
        raise ValueError("points must have shape (n, 2)")
    return t.contiguous()


def _voxel_rows(points_dev, voxel_size):
    """voxel_downsample (icp.py:117-129) of a device (n, 2) tensor -> device (m, 2) tensor."""
    n = points_dev.shape[0]
    if n == 0:
        raise ValueError("zero-size array to reduction operation minimum which has no identity")   # np.min, icp.py:119
    out = _b.voxel_downsample_set(_b.CloudSet(points_dev, np.array([0, n], dtype=np.int32)), voxel_size)
    return out.pts[: int(out.cnt[0].item())]


def submap_rotation_search(source_local, submap_global, predicted_pose, angle_range=60.0, angle_step=2.0,
                           fine_step=0.5, voxel_size=0.3):
    """slam.py:111-183 -> (R (2,2), t (2,)).

    Sweeps the rotation of the scan about the predicted pose (coarse grid, then a fine grid around the winner) as
    ONE chain of launches (``icpmi_rotation_search``: voxel filters, both sweeps and both arg-mins on the device,
    nothing read back in between), then refines the translation with one nearest-neighbour step over the closest 80 %
    of the matches — on the device too (``icpmi_rotation_refine``: NumPy's own arithmetic for the rotated rows, the
    percentile and the mean; scans of more than 2 048 raw rows refine on the host from the filtered clouds) — and reads
    back 16 doubles.  The angle grids are the reference's NumPy expressions (every fine grid that can follow a coarse
    winner is tabulated up front); R and t are the reference's bit for bit.
    """
    from utilities.features import _SearchContext, arange_rows
    _b.require_gpu()
    predicted_pose = np.asarray(predicted_pose, dtype=np.float64)
    src_in, tgt_in = _device_rows(source_local), _device_rows(submap_global)
    if src_in.shape[0] == 0 or tgt_in.shape[0] == 0:
        raise ValueError("zero-size array to reduction operation minimum which has no identity")   # np.min, icp.py:119
    pred_t = predicted_pose[:2, 2]
    pred_theta = np.arctan2(predicted_pose[1, 0], predicted_pose[0, 0])
    offsets = np.deg2rad(np.arange(-angle_range, angle_range + angle_step, angle_step))    # slam.py:146-151
    angles = pred_theta + offsets
    fine, fine_n = arange_rows(angles - np.deg2rad(angle_step), angles + np.deg2rad(angle_step),
                               np.deg2rad(fine_step))                            # slam.py:154-156, for every possible winner
    ctx = _SearchContext.get()
    ns, nt = int(src_in.shape[0]), int(tgt_in.shape[0])
    if ns + nt > ctx.cap:
        ctx.cap = max(2 * (ns + nt), 8192)
        ctx.stage = torch.empty((ctx.cap, 2), dtype=torch.float64).pin_memory()
        ctx.pts = torch.empty((ctx.cap, 2), dtype=torch.float64, device=ctx.dev)
    ctx.pts[:ns].copy_(src_in)                                                    # device to device: the submap never visits the host
    ctx.pts[ns:ns + nt].copy_(tgt_in)
    dtab = ctx.device_table(angles, fine, fine_n)
    max_fine = int(fine.shape[1])
    ws = ctx.workspace(ns, nt, len(angles), max_fine)
    _b._lib.check(_b._lib.lib().icpmi_rotation_search(_b._ptr(ctx.pts), ns, nt, float(voxel_size), _b._ptr(dtab[0]), len(angles),
                                                      _b._ptr(dtab[1]) if max_fine else None, _b._ptr(dtab[2]) if max_fine else None,
                                                      max_fine, 0, float(pred_t[0]), float(pred_t[1]), _b._ptr(ctx.rec), _b._ptr(ws),
                                                      ws.numel(), _b._stream()), "submap_rotation_search")
    L = _b._lib.lib()
    on_device = ns <= 2048                                                    # the refinement's finishing workgroup holds the rows in LDS
    if on_device:
        # translation: one nearest-neighbour centroid step over the closest 80 %, slam.py:161-181, still on the device
        need = L.icpmi_rotation_refine_workspace_bytes(ns)
        if getattr(ctx, "ref_ws", None) is None or ctx.ref_ws.numel() < need:
            ctx.ref_ws = torch.empty(2 * need, dtype=torch.uint8, device=ctx.dev)
            ctx.ref_out = torch.zeros(4, dtype=torch.float64, device=ctx.dev)
        _b._lib.check(L.icpmi_rotation_refine(_b._ptr(ws), ns, nt, _b._ptr(ctx.rec), _b._ptr(dtab[0]),
                                              _b._ptr(dtab[1]) if max_fine else None, max_fine, float(pred_t[0]), float(pred_t[1]),
                                              _b._ptr(ctx.ref_out), _b._ptr(ctx.ref_ws), ctx.ref_ws.numel(), _b._stream()),
                      "submap_rotation_search (refinement)")
        both = torch.cat([ctx.rec, ctx.ref_out]).cpu().numpy()                # one read-back
        rec, ref = both[:12], both[12:]
    else:
        rec = ctx.rec.cpu().numpy()
    if rec[0] < 5 or rec[1] < 5:                                              # slam.py:128-129
        return predicted_pose[:2, :2], predicted_pose[:2, 2]
    if on_device:
        k = int(rec[6])
        best_angle = angles[k]
        if int(rec[8]) > 0:                                                   # slam.py:157-159
            best_angle = fine[k, int(rec[9])]
        correction = np.degrees(best_angle - pred_theta)
        if abs(correction) > 1.0 and VERBOSE:
            print(f"  Submap rotation correction: {correction:+.1f}°")
        ca, sa = np.cos(best_angle), np.sin(best_angle)
        return np.array([[ca, -sa], [sa, ca]]), ref[:2].copy()
    src_d, tgt_d = ctx.filtered_clouds(ns, nt, rec)
    src = src_d.cpu().numpy()
    k = int(rec[6])
    best_angle = angles[k]
    if int(rec[8]) > 0:                                                       # slam.py:157-159
        best_angle = fine[k, int(rec[9])]
    correction = np.degrees(best_angle - pred_theta)
    if abs(correction) > 1.0 and VERBOSE:
        print(f"  Submap rotation correction: {correction:+.1f}°")
    ca, sa = np.cos(best_angle), np.sin(best_angle)
    R_best = np.array([[ca, -sa], [sa, ca]])
    # translation: one nearest-neighbour centroid step, slam.py:168-181
    rotated_src = src @ R_best.T
    placed = rotated_src + pred_t
    both = torch.cat([torch.from_numpy(np.ascontiguousarray(placed)).to(tgt_d.device), tgt_d], dim=0)
    cs = _b.CloudSet(both, np.array([0, len(placed), len(placed) + len(tgt_d)], dtype=np.int32))
    dist, idx = _b.nn_set(cs, [0], [1])
    nn_dists = dist[0, : len(placed)].cpu().numpy()
    nn_idx = idx[0, : len(placed)].cpu().numpy().astype(np.int64)
    nn_dists_sq = nn_dists ** 2
    dist_thresh = np.percentile(nn_dists_sq, 80)
    inlier_mask = nn_dists_sq <= dist_thresh
    if inlier_mask.sum() >= 5:
        matched = tgt_d[torch.from_numpy(nn_idx).to(tgt_d.device)].cpu().numpy()
        refined_t = np.mean(matched[inlier_mask] - rotated_src[inlier_mask], axis=0)
    else:
        refined_t = pred_t
    return R_best, refined_t


def attempt_submap_icp(source, submap, predicted, imu_yaw, imu_narrow, sub_rot_range, sub_rot_step, sub_rot_fine,
                       sub_rot_voxel, icp_cfg, sub_corr_dist):
    """slam.py:186-225: rotation search (narrow about the IMU yaw when one is given) + point-to-point ICP
    against the submap -> (r, t, error) like ``ICP()``."""
    from utilities import icp as _uicp
    pred = np.array(predicted, dtype=np.float64, copy=True)
    if imu_yaw is not None:                                                   # slam.py:200-204
        ca, sa = np.cos(imu_yaw), np.sin(imu_yaw)
        pred[:2, :2] = np.array([[ca, -sa], [sa, ca]])
        angle_range, angle_step = imu_narrow, 0.5
    else:
        angle_range, angle_step = sub_rot_range, sub_rot_step
    R_init, t_init = submap_rotation_search(source, submap, pred, angle_range=angle_range, angle_step=angle_step,
                                            fine_step=sub_rot_fine, voxel_size=sub_rot_voxel)
    src_d, sub_d = _device_rows(source), _device_rows(submap)
    pts = torch.cat([src_d, sub_d], dim=0)
    cs = _b.CloudSet(pts, np.array([0, len(src_d), len(src_d) + len(sub_d)], dtype=np.int32))
    b = _b.IcpBatch(cs, [0], [1], icp_cfg.get("error_threshold", 1e-7), icp_cfg.get("max_iterations", 100),
                    icp_cfg.get("voxel_size", 0.06), R_init, t_init, "point_to_point", 10, sub_corr_dist)
    b.run()
    R, t, err, info = b.unpack()
    return R[0], t[0], _uicp._report(err[0], info, 0, icp_cfg.get("max_iterations", 100))
