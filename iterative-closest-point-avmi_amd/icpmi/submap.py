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
"""
import numpy as np
import torch

from . import batch as _b


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
