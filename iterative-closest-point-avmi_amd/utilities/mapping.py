"""utilities.mapping — ``OccupancyGrid2D`` with the reference's interface
(/root/reference/utilities/mapping.py), the grid living in MI355X HBM.

``update_scan`` runs the Bresenham ray-cast and the log-odds update as two HIP
kernels (integer hit/miss counts per cell, then an exact replay of the
reference's float32(float64 + l) adds and the per-scan clip), so cell indices
and cell values are bit-identical to the NumPy implementation.
"""
import ctypes as C

import math

import numpy as np
import torch

from icpmi import _lib
from icpmi import batch as _b

UNEXPLORED = 0.0   # log-odds 0 = probability 0.5 (mapping.py:10)


class OccupancyGrid2D:
    """2-D log-odds occupancy grid; layout ``log_odds[iy, ix]``, shape (ny, nx), float32."""

    def __init__(self, min_x, max_x, min_y, max_y, resolution=0.1, p_hit=0.7, p_miss=0.4,
                 log_odds_min=-5.0, log_odds_max=5.0):
        _b.require_gpu()
        self.min_x, self.max_x = float(min_x), float(max_x)
        self.min_y, self.max_y = float(min_y), float(max_y)
        self.resolution = float(resolution)
        self.nx = int(np.ceil((self.max_x - self.min_x) / self.resolution))     # mapping.py:44-45
        self.ny = int(np.ceil((self.max_y - self.min_y) / self.resolution))
        if self.nx <= 0 or self.ny <= 0:
            raise ValueError("empty grid")
        self.l_hit = np.log(p_hit / (1.0 - p_hit))                                # np.float64, mapping.py:49-50
        self.l_miss = np.log(p_miss / (1.0 - p_miss))
        self.log_odds_min = float(log_odds_min)
        self.log_odds_max = float(log_odds_max)
        self._dev = torch.device("cuda", torch.cuda.current_device())
        self._grid = torch.zeros((self.ny, self.nx), dtype=torch.float32, device=self._dev)
        self._ws = None                    # counter workspace (four grids of uint32): allocated by the first update
        self._seq = 0                      # non-empty scans applied so far (selects counter grid / box slot)
        self._host = None                  # cached host copy of the grid
        self._full_clip = not (self.log_odds_min <= 0.0 <= self.log_odds_max)
        self._stages = None                # pinned staging rows of update_scan (NumPy in)
        self._call = None                  # converted constant arguments of the update call (see _apply)
        self.cell_updates = 0              # not tracked on the device; see update_scans()

    # ── the grid as the reference exposes it ─────────────────────────────────
    @property
    def log_odds(self):
        """READ-ONLY host copy (ny, nx) float32 of the device grid.

        The reference's attribute is the live array; here the live grid is in HBM, so an in-place edit of
        this copy (``grid.log_odds[...] = x``, ``np.clip(..., out=grid.log_odds)``) could not reach it and
        raises instead of being silently lost.  ASSIGN an array (``grid.log_odds = a``) to upload one."""
        if self._host is None:
            self._host = self._grid.cpu().numpy()
            self._host.setflags(write=False)
        return self._host

    @log_odds.setter
    def log_odds(self, value):
        v = np.ascontiguousarray(value, dtype=np.float32)
        if v.shape != (self.ny, self.nx):
            raise ValueError(f"log_odds must have shape {(self.ny, self.nx)}")
        self._grid.copy_(torch.from_numpy(v))
        self._host = None
        self._full_clip = True             # arbitrary values: the next scan clips every cell like np.clip does

    @property
    def device_log_odds(self):
        """The float32 (ny, nx) torch tensor in HBM (no copy)."""
        return self._grid

    # ── coordinate helpers, mapping.py:57-63,94-98 ───────────────────────────
    def _world_to_grid(self, wx, wy):
        ix = int(np.floor((wx - self.min_x) / self.resolution))
        iy = int(np.floor((wy - self.min_y) / self.resolution))
        return ix, iy

    def _in_bounds(self, ix, iy):
        return 0 <= ix < self.nx and 0 <= iy < self.ny

    def _world_to_grid_batch(self, wx, wy):
        L = _lib.lib()
        out = []
        for w, mn in ((wx, self.min_x), (wy, self.min_y)):
            w = torch.from_numpy(np.ascontiguousarray(w, dtype=np.float64)).to(self._dev)
            o = torch.empty(max(w.numel(), 1), dtype=torch.int64, device=self._dev)
            _lib.check(L.icpmi_world_to_grid(_b._ptr(w), w.numel(), mn, self.resolution, _b._ptr(o), _b._stream()),
                       "_world_to_grid_batch")
            out.append(o[:w.numel()].cpu().numpy())
        return out[0], out[1]

    @staticmethod
    def _bresenham(x0, y0, x1, y1):
        """Cells from (x0,y0) towards (x1,y1), end point excluded — mapping.py:68-89 — as a list of tuples."""
        cells = bresenham_cells(np.array([[x0, y0, x1, y1]], dtype=np.int64))[0]
        return [(int(x), int(y)) for x, y in cells]

    # ── update, mapping.py:103-141 ───────────────────────────────────────────
    def update_scan(self, origin_xy, hit_points):
        """Trace a ray from ``origin_xy`` (2,) to every row of ``hit_points`` (N, 2), world frame."""
        if isinstance(hit_points, torch.Tensor):
            if hit_points.numel() == 0:
                return
            self.update_scans(torch.as_tensor(origin_xy, dtype=torch.float64).reshape(1, 2), [hit_points])
            return
        hit_points = np.asarray(hit_points, dtype=np.float64)
        if hit_points.size == 0:                                   # mapping.py:113-114
            return
        if hit_points.ndim != 2 or hit_points.shape[1] != 2:
            raise ValueError("hit_points must have shape (N, 2)")
        # the call slam.py:557 makes once per scan: origin and hits go up in ONE asynchronous copy from a pinned buffer
        # (two buffers in turn: the previous call's copy may still be in flight), the cell box comes from the host rows
        n = hit_points.shape[0]
        st = self._stage(n + 1)
        h = st[0].numpy()
        h[0] = np.asarray(origin_xy, dtype=np.float64).reshape(2)
        h[1:n + 1] = hit_points
        st[1][:n + 1].copy_(st[0][:n + 1], non_blocking=True)
        st[2].record()
        rows = h[:n + 1]
        self._off1[1] = n
        self._apply(st[1][:1], st[1][1:n + 1], self._off1, None, self._box_of(*_minmax_rows(rows)))

    def _stage(self, rows):
        """(pinned host rows, device rows, event of the last copy out of the host rows) — two sets used in turn."""
        if self._stages is None or self._stages[0][0].shape[0] < rows:
            cap = max(4096, 2 * rows)
            self._stages = [(torch.empty((cap, 2), dtype=torch.float64).pin_memory(),
                             torch.empty((cap, 2), dtype=torch.float64, device=self._dev), torch.cuda.Event()) for _ in range(2)]
            self._stage_i = 0
            self._off1 = np.zeros(2, dtype=np.int32)
        self._stage_i ^= 1
        st = self._stages[self._stage_i]
        st[2].synchronize()                                        # its previous upload has left the pinned rows
        return st

    def update_scans(self, origins, hits, rows=None):
        """Apply several scans in order (the replay of slam.py:271-277) without returning to the host.

        origins: (S, 2); hits: list of S arrays (N_s, 2) or one packed (sum N_s, 2)
        torch tensor with ``hit_offsets``-style list semantics.
        rows=(begin, end): write only that band of grid rows — one rank's share of a sharded
        replay (``icpmi.dist.replay_scans_sharded``); every other row is left untouched.
        """
        L = _lib.lib()
        S = len(hits)
        if S == 0:
            return
        if isinstance(origins, torch.Tensor):
            org = origins.to(self._dev, torch.float64).contiguous()
        else:
            org = torch.from_numpy(np.ascontiguousarray(origins, dtype=np.float64).reshape(S, 2)).to(self._dev)
        sizes = [int(h.shape[0]) for h in hits]
        off = np.zeros(S + 1, dtype=np.int32)
        np.cumsum(sizes, out=off[1:])
        box = None
        if all(isinstance(h, torch.Tensor) for h in hits):
            packed = torch.cat([h.to(self._dev, torch.float64).reshape(-1, 2) for h in hits]) if off[-1] else None
        else:
            host = np.concatenate([np.asarray(h, dtype=np.float64).reshape(-1, 2) for h in hits]) if off[-1] else None
            packed = torch.from_numpy(np.ascontiguousarray(host)).to(self._dev) if host is not None else None
            if host is not None and not isinstance(origins, torch.Tensor):       # everything is on the host: no read-back
                o_host = np.asarray(origins, dtype=np.float64).reshape(S, 2)
                both = np.vstack([o_host, host])
                box = self._box_of(*_minmax_rows(both))
                # A long trajectory in one call: the box of all scans is far larger than any scan's reach, and the tile
                # pass of the library enumerates the tiles of the box per scan.  Consecutive scans lie close together, so
                # the replay goes down in pieces of _REPLAY_PIECE scans, each with its own box (same order, same result).
                if box is not None and S > self._REPLAY_PIECE and self._box_tiles(box) > 64:
                    for c0 in range(0, S, self._REPLAY_PIECE):
                        c1 = min(S, c0 + self._REPLAY_PIECE)
                        if off[c1] == off[c0]:
                            continue
                        piece = np.vstack([o_host[c0:c1], host[off[c0]:off[c1]]])
                        self._apply(org[c0:c1], packed[off[c0]:off[c1]], off[c0:c1 + 1] - off[c0], rows,
                                    self._box_of(*_minmax_rows(piece)))
                    return
        self._apply(org, packed, off, rows, box)

    _REPLAY_PIECE = 64

    @staticmethod
    def _box_tiles(box):
        """Tiles of 64 x 64 cells a cell box spans."""
        return ((int(box[2]) - int(box[0])) // 64 + 1) * ((int(box[3]) - int(box[1])) // 64 + 1)

    def _cell_box(self, org, packed):
        """Inclusive cell bounds {x0, y0, x1, y1} of the origins and hits (host int32[4]): every ray stays inside
        (Bresenham never leaves the rectangle of its end points).  Device tensors cost one small read-back."""
        if packed is None or packed.numel() == 0:
            return None
        pts = torch.cat([org.reshape(-1, 2), packed.reshape(-1, 2)])
        lo_hi = torch.stack([pts.amin(dim=0), pts.amax(dim=0)]).cpu().numpy()
        return self._box_of(lo_hi[0], lo_hi[1])

    def _box_of(self, lo, hi):
        # four scalars: plain float arithmetic (the same IEEE operations as the array expression, a fifth of its time —
        # this runs once per live scan)
        x0, y0, x1, y1 = float(lo[0]), float(lo[1]), float(hi[0]), float(hi[1])
        if not (math.isfinite(x0) and math.isfinite(y0) and math.isfinite(x1) and math.isfinite(y1)):
            return None                                         # a NaN / inf coordinate somewhere: no promise
        res, lim = self.resolution, 2.0 ** 29
        # same expression as the cell index, monotone in the coordinate
        c = [min(max(math.floor((v - m) / res), -lim), lim) for v, m in ((x0, self.min_x), (y0, self.min_y), (x1, self.min_x), (y1, self.min_y))]
        return np.array([c[0] - 1, c[1] - 1, c[2] + 1, c[3] + 1], dtype=np.int32)

    def _apply(self, org, packed, off, rows=None, box=None):
        """org (S,2) and packed hits (sum N,2) are float64 device tensors; off is a host int32 array.
        box: int32[4] cell bounds of all rays (``_cell_box``), computed here when not given.

        This is the per-scan call of a SLAM loop, so the host side is kept short: the function pointer and the grid's and
        the workspace's addresses are converted once, the raw stream handle is read without
        building a torch.cuda.Stream."""
        S = len(off) - 1
        r0, r1 = (0, self.ny) if rows is None else (int(rows[0]), int(rows[1]))
        if not 0 <= r0 <= r1 <= self.ny:
            raise ValueError("rows must satisfy 0 <= begin <= end <= ny")
        c = self._call
        if c is None:
            L = _lib.lib()
            if self._ws is None:
                self._ws = torch.zeros(L.icpmi_grid_workspace_bytes(self.ny, self.nx), dtype=torch.uint8, device=self._dev)
            c = self._call = (L.icpmi_grid_update_scans_box, C.c_void_p(self._grid.data_ptr()), C.c_void_p(self._ws.data_ptr()),
                              self._dev.index)
        if box is None:
            box = self._cell_box(org, packed)
        # (the scalars are the object's public attributes, as in the reference: read at every call)
        code = c[0](c[1], c[2], self.ny, self.nx, self.min_x, self.min_y, self.resolution, org.data_ptr(),
                    packed.data_ptr() if packed is not None else None, off.ctypes.data, S, float(self.l_hit), float(self.l_miss),
                    self.log_odds_min, self.log_odds_max, self._seq, 1 if self._full_clip else 0, r0, r1,
                    box.ctypes.data if box is not None else None, torch._C._cuda_getCurrentRawStream(c[3]))
        if code != 0:
            _lib.check(code, "update_scan")
        applied = (1 if off[1] > off[0] else 0) if S == 1 else int(np.count_nonzero(np.diff(off)))
        if r1 > r0:
            self._seq += applied
        if applied and (r0, r1) == (0, self.ny):
            self._full_clip = False        # every cell is inside [min, max] after a clipped scan
        self._host = None

    def reset(self):
        """Zero every cell (mapping.py:143-145)."""
        self._grid.zero_()
        self._host = None
        self._full_clip = not (self.log_odds_min <= 0.0 <= self.log_odds_max)

    # ── probability / display, mapping.py:150-166 (NumPy on the host copy) ───
    def to_probability(self):
        return 1.0 / (1.0 + np.exp(-self.log_odds))

    def to_display(self):
        lo = self.log_odds
        display = 1.0 - self.to_probability()
        display[lo == 0.0] = 1.0      # unexplored -> white
        display[lo < 0.0] = 0.85      # free -> light grey
        return display

    def _flat_cell_data(self):
        return self.to_display().ravel(order="C")

    # ── PyVista helpers, mapping.py:168-178 (display only; pyvista optional) ─
    def create_pyvista_grid(self):
        import pyvista as pv
        grid = pv.ImageData(dimensions=(self.nx + 1, self.ny + 1, 1),
                            spacing=(self.resolution, self.resolution, 1e-6),
                            origin=(self.min_x, self.min_y, 0.0))
        grid.cell_data["occ"] = self._flat_cell_data()
        return grid

    def update_pyvista_grid(self, grid):
        grid.cell_data["occ"] = self._flat_cell_data()

    # ── export, mapping.py:183-187 ───────────────────────────────────────────
    def save_csv(self, file_path):
        np.savetxt(file_path, self.to_probability(), delimiter=",")

    def save_npy(self, file_path):
        np.save(file_path, self.to_probability())


def _minmax_rows(rows):
    """(min, max) over the rows of an (n, 2) array — on a transposed copy: NumPy reduces axis 0 of a two-column array
    with an inner loop of length two (90 us for a 2 048-beam scan), and the contiguous axis of the copy in 5."""
    t = np.ascontiguousarray(rows.T)
    return t.min(axis=1), t.max(axis=1)


def bresenham_cells(segments):
    """Cells of ``_bresenham`` for every row ``[x0, y0, x1, y1]`` of ``segments`` -> list of (n_s, 2) int32 arrays."""
    _b.require_gpu()
    L = _lib.lib()
    seg = np.ascontiguousarray(segments, dtype=np.int64).reshape(-1, 4)
    if len(seg) and np.abs(seg).max() >= 2 ** 29:
        raise OverflowError("cell coordinates beyond +-2^29 are not supported")
    n = np.maximum(np.abs(seg[:, 2] - seg[:, 0]), np.abs(seg[:, 3] - seg[:, 1]))
    off = np.zeros(len(seg) + 1, dtype=np.int64)
    np.cumsum(n, out=off[1:])
    dev = torch.device("cuda", torch.cuda.current_device())
    d_seg = torch.from_numpy(seg.astype(np.int32)).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    d_out = torch.empty((max(int(off[-1]), 1), 2), dtype=torch.int32, device=dev)
    _lib.check(L.icpmi_bresenham_cells(_b._ptr(d_seg), _b._ptr(d_off), len(seg), _b._ptr(d_out), _b._stream()),
               "_bresenham")
    out = d_out.cpu().numpy()
    return [out[off[i]:off[i + 1]].copy() for i in range(len(seg))]
