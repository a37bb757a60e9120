"""Batched ``_run_icp_pair``: correlative rotation search + ICP for many scan pairs, nothing returning to the host
in between.

The reference matches its loop-closure candidates one after another (slam.py:575-579), and each match is
``_run_icp_pair`` (slam.py:53-98): ``rotation_search`` (utilities/features.py:165-242, the default pre-alignment,
config.yaml:34) and then ``ICP`` started from its result.  Here the searches of a whole batch are one chain of launches
behind ``icpmi_rotation_search_batch`` — voxel filter of every cloud at the search's own voxel size, their means, the
search order of the targets, one workgroup per pair for both sweeps — which leaves R_init / t_init of every pair in
device memory, where the batched ICP (``icpmi.batch.IcpBatch``) reads them.

The angle grids and their cos / sin are computed on the host with the reference's own NumPy expressions (cached on the
device), so the chosen angle, R and t are the reference's numbers bit for bit.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check
from .batch import CloudSet, IcpBatch, _ptr, _stream, require_gpu, unpack_results

REC_DOUBLES = 16
ST_OK, ST_FEW, ST_CAPACITY, ST_NO_FINE = 0, 1, 2, 3
RSB_MAX_ANGLES = 1024          # angles per sweep the batched kernel tabulates (csrc/rotsearch.hip)


def arange_rows(lo, hi, step):
    """``np.arange(lo[k], hi[k], step)`` for every k at once -> (values [K, L], lengths [K]); rows are padded with
    their last value.  Bit for bit NumPy's own numbers: arange takes ceil((stop - start) / step) elements and fills
    them as start + i * delta with delta = (start + step) - start."""
    lo = np.asarray(lo, dtype=np.float64).reshape(-1)
    hi = np.asarray(hi, dtype=np.float64).reshape(-1)
    step = np.float64(step)
    n = np.ceil((hi - lo) / step)
    n = np.where(np.isfinite(n) & (n > 0), n, 0).astype(np.int64)
    L = int(n.max()) if len(n) else 0
    i = np.arange(L, dtype=np.float64)[None, :]
    delta = ((lo + step) - lo)[:, None]
    vals = lo[:, None] + i * delta
    if L > 1:
        vals[:, 1] = lo + step
    if L > 0:
        vals[:, 0] = lo
    last = np.clip(n - 1, 0, None)
    vals = np.where(np.arange(L)[None, :] < n[:, None], vals, vals[np.arange(len(n)), last][:, None]) if L else vals
    return vals, n


class AngleTables:
    """Coarse angles of features.py:221 and, for every coarse angle that can win, the fine grid of features.py:227-229,
    with cos / sin of all of them on the device (features.py:214 uses np.cos / np.sin).  One per (device, steps)."""
    _cache = {}

    @classmethod
    def get(cls, dev, angle_step_coarse, angle_step_fine):
        key = (dev, float(angle_step_coarse), float(angle_step_fine))
        if key not in cls._cache:
            cls._cache[key] = cls(dev, angle_step_coarse, angle_step_fine)
        return cls._cache[key]

    def __init__(self, dev, angle_step_coarse, angle_step_fine):
        self.coarse = np.deg2rad(np.arange(-180, 180, angle_step_coarse))            # features.py:221
        lo = self.coarse - np.deg2rad(angle_step_coarse)                             # features.py:227-229, for every possible winner
        hi = self.coarse + np.deg2rad(angle_step_coarse)
        self.fine, self.fine_n = arange_rows(lo, hi, np.deg2rad(angle_step_fine))
        self.max_fine = int(self.fine.shape[1]) if self.fine.ndim == 2 else 0
        cs = np.ascontiguousarray(np.stack([np.cos(self.coarse), np.sin(self.coarse)], axis=1))
        fcs = (np.ascontiguousarray(np.stack([np.cos(self.fine), np.sin(self.fine)], axis=2)) if self.fine.size
               else np.zeros((len(self.coarse), 0, 2)))
        self.d_cs = torch.from_numpy(cs).to(dev)
        self.d_fcs = torch.from_numpy(fcs).to(dev)
        self.d_fn = torch.from_numpy(np.ascontiguousarray(self.fine_n, dtype=np.int32)).to(dev)

    @property
    def device_table(self):
        return self.d_cs, self.d_fcs, self.d_fn


class RotationSearchBatch:
    """rotation_search (features.py:165-242) of every pair of a cloud set, resident on the device.

    ``run()`` enqueues the chain on the current stream and returns the (B, 16) record tensor; ``init`` ([B, 6]: R row
    major, t) then holds R_init / t_init of every pair for ``icpmi_icp_batch``.  ``results()`` reads the records back
    and forms (R, t, score) with the reference's NumPy expressions (features.py:235-237)."""

    def __init__(self, clouds, pair_src, pair_tgt, voxel_size=0.3, angle_step_coarse=2.0, angle_step_fine=0.2,
                 init=None, max_rows_hint=0):
        require_gpu()
        L = _lib.lib()
        self.raw = clouds if isinstance(clouds, CloudSet) else CloudSet.from_numpy(clouds)
        if self.raw.dim != 2:
            raise ValueError("rotation_search is 2-D")
        dev = self.raw.pts.device
        self.pair_src_host = np.ascontiguousarray(pair_src, dtype=np.int32)
        self.pair_tgt_host = np.ascontiguousarray(pair_tgt, dtype=np.int32)
        self.B = len(self.pair_src_host)
        if len(self.pair_tgt_host) != self.B:
            raise ValueError("pair_src and pair_tgt differ in length")
        self.pair_src = torch.from_numpy(self.pair_src_host).to(dev)
        self.pair_tgt = torch.from_numpy(self.pair_tgt_host).to(dev)
        self.tgt_ids = torch.from_numpy(np.unique(self.pair_tgt_host).astype(np.int32)).to(dev)
        self.voxel_size = float(voxel_size)
        if not self.voxel_size > 0:
            raise ValueError("voxel_size must be positive")
        self.steps = (angle_step_coarse, angle_step_fine)
        self.tables = AngleTables.get(dev, angle_step_coarse, angle_step_fine)
        self.max_rows_hint = int(max_rows_hint)
        self.too_many_angles = len(self.tables.coarse) > RSB_MAX_ANGLES or self.tables.max_fine > RSB_MAX_ANGLES
        self.records = torch.zeros((max(self.B, 1), REC_DOUBLES), dtype=torch.float64, device=dev)
        self.init = init if init is not None else torch.zeros((max(self.B, 1), 6), dtype=torch.float64, device=dev)
        need = L.icpmi_rotation_search_batch_workspace_bytes(self.raw.total_rows, self.raw.n_clouds, self.raw.max_n)
        self.ws = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)

    def run(self):
        t = self.tables
        mf = t.max_fine
        if self.too_many_angles:                   # icpmi_rotation_search_batch would answer ICPMI_ERR_UNSUPPORTED
            self.records.zero_()
            self.records[:, 11] = ST_CAPACITY      # results() searches such pairs one by one: same numbers
            return self.records
        check(_lib.lib().icpmi_rotation_search_batch(
            _ptr(self.raw.pts), _ptr(self.raw.off), self.raw.off_host.ctypes.data_as(C.c_void_p), self.raw.n_clouds,
            _ptr(self.tgt_ids), len(self.tgt_ids), _ptr(self.pair_src), _ptr(self.pair_tgt), self.B, self.voxel_size,
            _ptr(t.d_cs), len(t.coarse), _ptr(t.d_fcs) if mf else None, _ptr(t.d_fn) if mf else None, mf,
            self.max_rows_hint, _ptr(self.records), _ptr(self.init), _ptr(self.ws), self.ws.numel(), _stream()),
            "rotation_search (batch)")
        return self.records

    def results(self, records=None):
        """-> (R [B,2,2], t [B,2], score [B], records [B,16]) on the host (synchronises).  Pairs the on-chip search could
        not hold (status 2: a filtered cloud above the capacity hint) are searched one by one through the single-pair
        entry — same numbers."""
        rec = (self.records if records is None else records).cpu().numpy()[:self.B]
        t = self.tables
        R = np.tile(np.eye(2), (self.B, 1, 1))
        tt = np.zeros((self.B, 2))
        score = np.full(self.B, np.inf)
        status = rec[:, 11].astype(np.int64)
        if (status == ST_NO_FINE).any():
            raise ValueError("attempt to get argmin of an empty sequence")          # np.argmin(scores_fine) on an empty grid
        ok = np.flatnonzero(status == ST_OK)
        if len(ok):
            k = rec[ok, 6].astype(np.int64)
            j = rec[ok, 9].astype(np.int64)
            ang = t.fine[k, j]
            ca, sa = np.cos(ang), np.sin(ang)
            for q, i in enumerate(ok):                                              # features.py:235-237, NumPy's own matmul
                Ri = np.array([[ca[q], -sa[q]], [sa[q], ca[q]]])
                R[i] = Ri
                tt[i] = rec[i, 4:6] - Ri @ rec[i, 2:4]
            score[ok] = rec[ok, 10]
        over = np.flatnonzero(status == ST_CAPACITY)
        if len(over):
            from utilities import features
            clouds = self.raw.to_numpy()
            keep = features.VERBOSE
            features.VERBOSE = False
            try:
                for i in over:
                    R[i], tt[i], score[i] = features.rotation_search(
                        clouds[self.pair_src_host[i]], clouds[self.pair_tgt_host[i]], self.voxel_size, *self.steps)
            finally:
                features.VERBOSE = keep
        return R, tt, score, rec

def rotation_search_batch(sources, targets, voxel_size=0.3, angle_step_coarse=2.0, angle_step_fine=0.2):
    """rotation_search(sources[i], targets[i]) for every i in one chain of launches -> (R [B,2,2], t [B,2], score [B]).
    ``sources`` may be one array shared by every pair (the loop-closure shape, slam.py:576-579)."""
    clouds, ps, pt = _pair_lists(sources, targets)
    b = RotationSearchBatch(clouds, ps, pt, voxel_size, angle_step_coarse, angle_step_fine)
    b.run()
    R, t, score, _ = b.results()
    return R, t, score


def _pair_lists(sources, targets):
    targets = list(targets)
    B = len(targets)
    if isinstance(sources, np.ndarray) and sources.ndim == 2:
        return [sources] + targets, np.zeros(B, dtype=np.int32), np.arange(1, B + 1, dtype=np.int32)
    sources = list(sources)
    if len(sources) != B:
        raise ValueError("sources and targets differ in length")
    return sources + targets, np.arange(B, dtype=np.int32), np.arange(B, 2 * B, dtype=np.int32)


class RunIcpPairBatch:
    """``_run_icp_pair`` (slam.py:53-98, alignment_method "rotation_search") for a batch of pairs resident in HBM:
    ``run()`` = rotation search of every pair, then ICP of every pair from its own R_init / t_init — one stream, no host
    round trip; returns the (B, 16) ICP result tensor (icpmi.batch.unpack_results)."""

    def __init__(self, clouds, pair_src, pair_tgt, error_threshold=1e-7, max_iterations=100, voxel_size=0.06,
                 method="point_to_line", normal_k=10, rotation_voxel_size=0.3, angle_step_coarse=2.0, angle_step_fine=0.2,
                 max_corr_dist=None, max_rows_hint=0):
        raw = clouds if isinstance(clouds, CloudSet) else CloudSet.from_numpy(clouds)
        B = len(pair_src)
        self.icp = IcpBatch(raw, pair_src, pair_tgt, error_threshold, max_iterations, voxel_size,
                            np.tile(np.eye(2), (B, 1, 1)), np.zeros((B, 2)), method, normal_k, max_corr_dist)
        self.search = RotationSearchBatch(raw, pair_src, pair_tgt, rotation_voxel_size, angle_step_coarse, angle_step_fine,
                                          init=self.icp.init, max_rows_hint=max_rows_hint)
        self.B = B

    def run(self, events=None):
        if self.search.too_many_angles:
            # more angles than the batched kernel tabulates (a step below ~0.36 degrees): every pair is searched by the
            # single-pair entry, as pairs beyond the capacity hint are — same numbers
            self.search.records.zero_()
            self.search.records[:, 11] = ST_CAPACITY
            if events is not None:
                events[0].record(); events[1].record()
            return self.icp.results
        self.search.run()
        return self.icp.run(events=events)

    def unpack(self):
        """(R, t, err, info) of the ICPs; pairs whose search fell outside the on-chip capacity (status 2) are redone
        with the single-pair search's result as their start."""
        rec = self.search.records.cpu().numpy()[:self.B]
        res = self.icp.results.cpu().numpy()[:self.B].copy()
        if (rec[:, 11].astype(np.int64) == ST_NO_FINE).any():
            raise ValueError("attempt to get argmin of an empty sequence")          # features.py:231: np.argmin of an empty fine grid
        over = np.flatnonzero(rec[:, 11].astype(np.int64) == ST_CAPACITY)
        if len(over):
            from .batch import icp_pair
            R0, t0, _, _ = self.search.results()
            clouds = self.search.raw.to_numpy()
            p = self.icp.params
            for i in over:
                Ri, ti, ei, info = icp_pair(clouds[self.icp.pair_src_host[i]], clouds[self.icp.pair_tgt_host[i]],
                                            p.error_threshold, p.max_iterations, self.icp.voxel_size, R0[i], t0[i],
                                            "point_to_line" if self.icp.use_p2l else "point_to_point", self.icp.normal_k,
                                            None if p.max_corr_dist < 0 else p.max_corr_dist)
                res[i, :] = 0.0
                res[i, 0:4] = Ri[0].reshape(4); res[i, 9:11] = ti[0]; res[i, 12] = ei[0]
                res[i, 13] = info["delta"][0]; res[i, 14] = info["iters"][0]; res[i, 15] = info["status"][0]
        return unpack_results(res, 2)


def run_icp_pair_batch(sources, targets, icp_cfg=None, feat_cfg=None):
    """``_run_icp_pair(sources[i], targets[i], icp_cfg, feat_cfg, "rotation_search")`` for every i (slam.py:53-98, same
    configuration keys and defaults) -> (R [B,2,2], t [B,2], err [B], info)."""
    icp_cfg, feat_cfg = icp_cfg or {}, feat_cfg or {}
    clouds, ps, pt = _pair_lists(sources, targets)
    b = RunIcpPairBatch(clouds, ps, pt,
                        error_threshold=icp_cfg.get("error_threshold", 1e-7), max_iterations=icp_cfg.get("max_iterations", 100),
                        voxel_size=icp_cfg.get("voxel_size", 0.06), method=icp_cfg.get("method", "point_to_line"),
                        normal_k=icp_cfg.get("normal_k", 10), rotation_voxel_size=feat_cfg.get("rotation_voxel_size", 0.3),
                        angle_step_coarse=feat_cfg.get("angle_step_coarse", 2.0), angle_step_fine=feat_cfg.get("angle_step_fine", 0.2))
    b.run()
    return b.unpack()
