"""Batched scan-pair registration on one MI355X.

The reference runs its scan pairs one after another (loop-closure candidates,
slam.py:575-597; scan-to-submap, slam.py:505-510).  Here a batch of independent
pairs is one pipeline of three launches on the current HIP stream — voxel
filter of every cloud, normals of every target cloud, one fused ICP workgroup
per pair — with device-side cloud sizes in between, so nothing returns to the
host until the results are read.  ``utilities.icp.ICP`` is this with one pair.

torch is used for device memory and streams only.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import IcpmiError, IcpParams, check


PREP_MAX_POINTS = 4096      # clouds up to this many rows take the sorted-sweep kernels (prep.hip, icp2.hip)


def require_gpu():
    if not torch.cuda.is_available():
        raise IcpmiError("libicpmi needs an AMD GPU (gfx950); torch.cuda.is_available() is False "
                         "and there is no CPU fallback")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class CloudSet:
    """Point clouds packed row-major in one float64 device buffer.

    off_host/off: first row of each cloud (C+1 entries, capacity based);
    cnt: valid rows per cloud on the device (None = full capacity).
    """

    def __init__(self, pts, off_host, cnt=None, off=None):
        self.pts = pts
        self.off_host = np.ascontiguousarray(off_host, dtype=np.int32)
        self.off = off if off is not None else torch.from_numpy(self.off_host).to(pts.device)
        self.cnt = cnt
        self.dim = int(pts.shape[1])

    @property
    def n_clouds(self):
        return len(self.off_host) - 1

    @property
    def total_rows(self):
        return int(self.off_host[-1])

    @property
    def max_n(self):
        return int(np.diff(self.off_host).max()) if self.n_clouds else 0

    @classmethod
    def from_numpy(cls, clouds, device=None):
        require_gpu()
        device = device or torch.device("cuda", torch.cuda.current_device())
        arrs = [np.ascontiguousarray(c, dtype=np.float64) for c in clouds]
        dim = arrs[0].shape[1] if arrs else 2
        if dim not in (2, 3) or any(a.ndim != 2 or a.shape[1] != dim for a in arrs):
            raise ValueError("clouds must be (n, 2) or (n, 3) arrays of one dimensionality")
        off = np.zeros(len(arrs) + 1, dtype=np.int64)
        np.cumsum([len(a) for a in arrs], out=off[1:])
        if off[-1] >= 2 ** 31:
            raise ValueError("cloud set too large for 32-bit row offsets")
        host = np.concatenate(arrs, axis=0) if arrs and off[-1] > 0 else np.empty((0, dim))
        pts = torch.from_numpy(host).to(device)
        if pts.shape[0] == 0:
            pts = torch.empty((1, dim), dtype=torch.float64, device=device)   # never hand out a null pointer
        return cls(pts, off.astype(np.int32))

    def counts_host(self):
        if self.cnt is None:
            return np.diff(self.off_host)
        return self.cnt.cpu().numpy()

    def to_numpy(self):
        """List of (cnt_c, dim) arrays (synchronises)."""
        host = self.pts.cpu().numpy()
        cnt = self.counts_host()
        return [host[self.off_host[c]:self.off_host[c] + cnt[c]].copy() for c in range(self.n_clouds)]


def voxel_downsample_set(cs, voxel_size, out=None, workspace=None):
    """voxel_downsample (reference icp.py:117-129) of every cloud of the set."""
    L = _lib.lib()
    if not voxel_size > 0:
        raise ValueError("voxel_size must be positive")
    if out is None:
        out = CloudSet(torch.empty_like(cs.pts), cs.off_host,
                       cnt=torch.empty(max(cs.n_clouds, 1), dtype=torch.int32, device=cs.pts.device), off=cs.off)
    need = L.icpmi_voxel_workspace_bytes(cs.max_n)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=cs.pts.device)
    check(L.icpmi_voxel_downsample_batch(_ptr(cs.pts), _ptr(cs.off), cs.off_host.ctypes.data_as(C.c_void_p),
                                         cs.n_clouds, cs.dim, float(voxel_size), _ptr(out.pts), _ptr(out.cnt),
                                         _ptr(workspace), workspace.numel(), _stream()), "voxel_downsample")
    return out


def normals_set(cs, k, cloud_ids=None, out=None, workspace=None):
    """estimate_normals_2d (reference icp.py:51-76) for the selected clouds of a 2-D set."""
    L = _lib.lib()
    if cs.dim != 2:
        raise ValueError("normals are defined for 2-D clouds only")
    if out is None:
        out = torch.zeros((max(cs.total_rows, 1), 2), dtype=torch.float64, device=cs.pts.device)
    if cloud_ids is None:
        n_sel, ids_t, max_n = cs.n_clouds, None, cs.max_n
    else:
        ids = np.ascontiguousarray(cloud_ids, dtype=np.int32)
        n_sel = len(ids)
        max_n = int(np.diff(cs.off_host)[ids].max()) if n_sel else 0
        ids_t = torch.from_numpy(ids).to(cs.pts.device)
    if max_n <= PREP_MAX_POINTS or k > 31:
        # sweep search on the axis-sorted copy (prep.hip; clouds above 4096 rows through global memory); the sorted copy
        # is a by-product.  Any k (the exhaustive kernel below holds its lists in registers: k <= 31).
        need = L.icpmi_prepared_bytes(cs.total_rows, cs.n_clouds, max_n)
        if workspace is None or workspace.numel() < need:
            workspace = torch.empty(need, dtype=torch.uint8, device=cs.pts.device)
        ids_host = None if cloud_ids is None else np.ascontiguousarray(cloud_ids, dtype=np.int32)
        check(L.icpmi_prepare_targets_ex(_ptr(cs.pts), _ptr(cs.off), cs.off_host.ctypes.data_as(C.c_void_p), _ptr(cs.cnt),
                                         _ptr(ids_t), None if ids_host is None else ids_host.ctypes.data_as(C.c_void_p), n_sel,
                                         cs.n_clouds, cs.total_rows, max_n, int(k), _ptr(out), _ptr(workspace),
                                         workspace.numel(), 1, _stream()), "estimate_normals_2d")
        return out
    need = L.icpmi_normals_workspace_bytes(cs.total_rows, max_n)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=cs.pts.device)
    check(L.icpmi_normals_2d_batch(_ptr(cs.pts), _ptr(cs.off), _ptr(cs.cnt), _ptr(ids_t), n_sel, cs.total_rows,
                                   max_n, int(k), _ptr(out), _ptr(workspace), workspace.numel(), _stream()),
          "estimate_normals_2d")
    return out


def nn_set(cs, pair_src, pair_tgt):
    """1-NN of every row of cloud pair_src[b] in cloud pair_tgt[b] -> (dist, idx) device tensors [B, stride]."""
    L = _lib.lib()
    ps = torch.as_tensor(np.ascontiguousarray(pair_src, dtype=np.int32), device=cs.pts.device)
    pt = torch.as_tensor(np.ascontiguousarray(pair_tgt, dtype=np.int32), device=cs.pts.device)
    B = len(pair_src)
    stride = max(int(np.diff(cs.off_host)[np.asarray(pair_src)].max()) if B else 0, 1)
    idx = torch.empty((max(B, 1), stride), dtype=torch.int32, device=cs.pts.device)
    dist = torch.empty((max(B, 1), stride), dtype=torch.float64, device=cs.pts.device)
    check(L.icpmi_nn_batch(_ptr(cs.pts), _ptr(cs.off), _ptr(cs.cnt), _ptr(ps), _ptr(pt), B, stride, cs.dim,
                           _ptr(idx), _ptr(dist), stride, _stream()), "nn")
    return dist, idx


def nn_set_sweep(cs, pair_src, pair_tgt, return_second=False):
    """Same result as ``nn_set`` by the sorted-sweep search (2-D, target clouds of at most 4096 rows):
    prepares the target clouds (axis choice + sort), then binary search + outward sweep per query."""
    L = _lib.lib()
    if cs.dim != 2:
        raise ValueError("the sweep search is 2-D only")
    dev = cs.pts.device
    pair_src = np.ascontiguousarray(pair_src, dtype=np.int32)
    pair_tgt = np.ascontiguousarray(pair_tgt, dtype=np.int32)
    B = len(pair_src)
    sizes = np.diff(cs.off_host)
    tgt_ids = np.unique(pair_tgt)
    max_tgt = int(sizes[tgt_ids].max()) if B else 0
    if max_tgt > PREP_MAX_POINTS:
        raise ValueError(f"target clouds above {PREP_MAX_POINTS} rows need the exhaustive search (nn_set)")
    stride = max(int(sizes[pair_src].max()) if B else 0, 1)
    prepared = torch.empty(L.icpmi_prepared_bytes(cs.total_rows, cs.n_clouds, max_tgt), dtype=torch.uint8, device=dev)
    ids = torch.from_numpy(tgt_ids.astype(np.int32)).to(dev)
    check(L.icpmi_prepare_targets(_ptr(cs.pts), _ptr(cs.off), None, _ptr(cs.cnt), _ptr(ids), None, len(tgt_ids),
                                  cs.n_clouds, cs.total_rows, max_tgt, -1, None, _ptr(prepared), prepared.numel(),
                                  _stream()), "prepare_targets")
    ps, pt = torch.from_numpy(pair_src).to(dev), torch.from_numpy(pair_tgt).to(dev)
    idx = torch.empty((max(B, 1), stride), dtype=torch.int32, device=dev)
    dist = torch.empty((max(B, 1), stride), dtype=torch.float64, device=dev)
    second = torch.empty((max(B, 1), stride), dtype=torch.float64, device=dev) if return_second else None
    check(L.icpmi_nn_prepared_batch(_ptr(cs.pts), _ptr(cs.off), _ptr(cs.cnt), _ptr(prepared), _ptr(ps), _ptr(pt), B,
                                    stride, max_tgt, cs.total_rows, _ptr(idx), _ptr(dist), _ptr(second), stride,
                                    _stream()), "nn (sweep)")
    return (dist, idx, second) if return_second else (dist, idx)


class IcpBatch:
    """A batch of scan pairs resident in HBM, ready to be registered repeatedly.

    clouds: list of arrays; pair_src/pair_tgt: cloud indices per pair (a source
    shared by many pairs is stored once).  ``run()`` performs exactly what the
    reference's ``ICP()`` does per pair — voxel filter of source and target,
    target normals for point_to_line, the ICP loop — and leaves a
    (B, 16) float64 result tensor on the device.
    """

    def __init__(self, clouds, pair_src, pair_tgt, error_threshold, max_iterations, voxel_size,
                 R_init=None, t_init=None, method="point_to_point", normal_k=10, max_corr_dist=None,
                 force_exhaustive=False):
        require_gpu()
        L = _lib.lib()
        self.raw = clouds if isinstance(clouds, CloudSet) else CloudSet.from_numpy(clouds)
        dev = self.raw.pts.device
        self.dim = self.raw.dim
        self.pair_src_host = np.ascontiguousarray(pair_src, dtype=np.int32)
        self.pair_tgt_host = np.ascontiguousarray(pair_tgt, dtype=np.int32)
        self.B = len(self.pair_src_host)
        if len(self.pair_tgt_host) != self.B:
            raise ValueError("pair_src and pair_tgt differ in length")
        self.pair_src = torch.from_numpy(self.pair_src_host).to(dev)
        self.pair_tgt = torch.from_numpy(self.pair_tgt_host).to(dev)
        self.voxel_size = float(voxel_size)
        self.normal_k = int(normal_k)
        use_p2l = method == "point_to_line" and self.dim == 2          # icp.py:162
        if method not in ("point_to_point", "point_to_line"):
            use_p2l = False                                             # any other string: icp.py:196 else-branch
        have_init = R_init is not None and t_init is not None          # icp.py:153
        self.params = IcpParams(float(error_threshold), -1.0 if max_corr_dist is None else float(max_corr_dist),
                                int(max_iterations), _lib.POINT_TO_LINE if use_p2l else _lib.POINT_TO_POINT,
                                1 if have_init else 0, self.dim)
        self.use_p2l = use_p2l
        self.init = None
        if have_init:
            d = self.dim
            R = np.broadcast_to(np.asarray(R_init, dtype=np.float64), (self.B, d, d)).reshape(self.B, d * d)
            t = np.broadcast_to(np.asarray(t_init, dtype=np.float64), (self.B, d))
            self.init = torch.from_numpy(np.ascontiguousarray(np.concatenate([R, t], axis=1))).to(dev)
        sizes = np.diff(self.raw.off_host)
        self.max_src_n = int(sizes[self.pair_src_host].max()) if self.B else 0
        self.tgt_ids = np.unique(self.pair_tgt_host)
        # persistent device buffers: nothing is allocated inside run()
        self.vox = CloudSet(torch.empty_like(self.raw.pts), self.raw.off_host,
                            cnt=torch.zeros(max(self.raw.n_clouds, 1), dtype=torch.int32, device=dev), off=self.raw.off)
        self.vox_ws = torch.empty(L.icpmi_voxel_workspace_bytes(self.raw.max_n), dtype=torch.uint8, device=dev)
        self.normals = None
        self.tgt_ids_dev = torch.from_numpy(self.tgt_ids.astype(np.int32)).to(dev)
        self.max_tgt_n = int(sizes[self.tgt_ids].max()) if len(self.tgt_ids) else 0
        # fast path: 2-D, every cloud small enough for the on-chip kernels
        # (targets above PREP_MAX_POINTS rows — a rolling submap — are sorted through global memory and searched via L2)
        self.fast = self.dim == 2 and self.max_src_n <= PREP_MAX_POINTS and not force_exhaustive
        self.prepared = None
        self.icp_ws = None
        if self.fast:
            self.prepared = torch.empty(L.icpmi_prepared_bytes(self.raw.total_rows, self.raw.n_clouds, self.max_tgt_n),
                                        dtype=torch.uint8, device=dev)
            self.tgt_ids_host = np.ascontiguousarray(self.tgt_ids, dtype=np.int32)
            # parked pairs (csrc/icp2.hip): a large batch runs in two stages — the pairs still running after 12
            # iterations are continued together by a second launch — and pairs that start metres from their target
            # are continued by the kernel for far queries
            self.icp_ws = torch.empty(L.icpmi_icp_workspace_bytes(self.B, self.max_src_n, self.dim),
                                      dtype=torch.uint8, device=dev)
        else:
            if use_p2l:
                self.normals = torch.zeros((max(self.raw.total_rows, 1), 2), dtype=torch.float64, device=dev)
                self.nrm_ws = torch.empty(L.icpmi_normals_workspace_bytes(self.raw.total_rows, self.max_tgt_n),
                                          dtype=torch.uint8, device=dev)
            self.icp_ws = torch.empty(L.icpmi_icp_workspace_bytes(self.B, self.max_src_n, self.dim),
                                      dtype=torch.uint8, device=dev)
        self.results = torch.zeros((max(self.B, 1), _lib.RES_DOUBLES), dtype=torch.float64, device=dev)

    def run(self, events=None):
        """Enqueue voxel filter -> normals -> fused ICP on the current stream; returns the device result tensor.

        events: optional (start, end) torch.cuda.Event pair recorded around the fused ICP launch only."""
        L = _lib.lib()
        st = _stream()
        voxel_downsample_set(self.raw, self.voxel_size, out=self.vox, workspace=self.vox_ws)
        if self.fast:
            # allow_polar: the sort order (a projection, or the bearing about the frame origin) is the library's choice
            check(L.icpmi_prepare_targets_ex(_ptr(self.vox.pts), _ptr(self.vox.off),
                                             self.raw.off_host.ctypes.data_as(C.c_void_p), _ptr(self.vox.cnt),
                                             _ptr(self.tgt_ids_dev), self.tgt_ids_host.ctypes.data_as(C.c_void_p),
                                             len(self.tgt_ids), self.raw.n_clouds, self.raw.total_rows, self.max_tgt_n,
                                             self.normal_k if self.use_p2l else -1, None, _ptr(self.prepared),
                                             self.prepared.numel(), 1, st), "prepare_targets")
        elif self.use_p2l and self.normal_k > 31:
            normals_set(self.vox, self.normal_k, cloud_ids=self.tgt_ids, out=self.normals)      # any k: the prepare path
        elif self.use_p2l:
            check(L.icpmi_normals_2d_batch(_ptr(self.vox.pts), _ptr(self.vox.off), _ptr(self.vox.cnt),
                                           _ptr(self.tgt_ids_dev), len(self.tgt_ids), self.raw.total_rows,
                                           self.max_tgt_n, self.normal_k, _ptr(self.normals), _ptr(self.nrm_ws),
                                           self.nrm_ws.numel(), st), "estimate_normals_2d")
        if events is not None:
            events[0].record()
        check(L.icpmi_icp_batch(_ptr(self.vox.pts), _ptr(self.vox.off), _ptr(self.vox.cnt), _ptr(self.normals),
                                _ptr(self.prepared), _ptr(self.pair_src), _ptr(self.pair_tgt), self.B,
                                self.max_src_n, self.max_tgt_n, self.raw.total_rows,
                                C.byref(self.params), _ptr(self.init), _ptr(self.results), _ptr(self.icp_ws),
                                self.icp_ws.numel() if self.icp_ws is not None else 0, st), "ICP")
        if events is not None:
            events[1].record()
        return self.results

    def unpack(self, results=None):
        """(R [B,d,d], t [B,d], err [B], info) on the host (synchronises)."""
        res = (self.results if results is None else results).cpu().numpy()[:self.B]
        return unpack_results(res, self.dim)


def unpack_results(res, dim):
    d = dim
    R = res[:, _lib.RES_R:_lib.RES_R + d * d].reshape(-1, d, d).copy()
    t = res[:, _lib.RES_T:_lib.RES_T + d].copy()
    err = res[:, _lib.RES_ERR].copy()
    info = dict(iters=res[:, _lib.RES_ITERS].astype(np.int64), status=res[:, _lib.RES_STATUS].astype(np.int64),
                delta=res[:, _lib.RES_DELTA].copy())
    return R, t, err, info


class PairContext:
    """One scan pair at a time through buffers that stay allocated (one context per device).

    ``utilities.icp.ICP`` is called once per scan by a SLAM loop (slam.py:90, 217, 471); building an ``IcpBatch`` per
    call — a dozen allocations, uploads of offsets and pair lists — cost more than its kernels.  Here the device
    buffers are sized for a capacity (grown when a larger cloud arrives) and a call is: one pinned upload of the
    two clouds and their offsets, the three launches of ``IcpBatch.run``, one 128-byte read-back.  2-D clouds of at
    most 4096 rows each (the on-chip kernels); anything else builds a fresh ``IcpBatch``."""
    _per_device = {}

    @classmethod
    def get(cls):
        require_gpu()
        dev = torch.device("cuda", torch.cuda.current_device())
        if dev not in cls._per_device:
            cls._per_device[dev] = cls(dev)
        return cls._per_device[dev]

    def __init__(self, dev):
        self.dev = dev
        self.cap_s = self.cap_t = 0
        self.batch = None

    def _grow(self, ns, nt):
        # source and target capacities grow independently, each at most PREP_MAX_POINTS rows, so the batch object always
        # takes the on-chip kernels (a 2 100-row scan against a 3 000-row target: 5 100 rows in all, both halves fit)
        self.cap_s = min(PREP_MAX_POINTS, max(2048, self.cap_s, 2 * ns))
        self.cap_t = min(PREP_MAX_POINTS, max(2048, self.cap_t, 2 * nt))
        cap = self.cap_s + self.cap_t
        dummy = CloudSet(torch.zeros((cap, 2), dtype=torch.float64, device=self.dev),
                         np.array([0, self.cap_s, cap], dtype=np.int32))
        self.batch = IcpBatch(dummy, [0], [1], 1e-6, 1, 1.0, np.eye(2), np.zeros(2), "point_to_line", 1, None)
        assert self.batch.fast, "PairContext must stay on the sorted-sweep kernels"
        self.stage = torch.empty((cap, 2), dtype=torch.float64).pin_memory()
        self.off_stage = torch.zeros(3, dtype=torch.int32).pin_memory()
        self.init_stage = torch.zeros((1, 6), dtype=torch.float64).pin_memory()
        self.res_host = torch.zeros((1, _lib.RES_DOUBLES), dtype=torch.float64).pin_memory()

    def solve(self, source, target, error_threshold, max_iterations, voxel_size, R_init, t_init, method, normal_k,
              max_corr_dist):
        """-> one (RES_DOUBLES,) float64 result record on the host."""
        ns, nt = len(source), len(target)
        if ns > self.cap_s or nt > self.cap_t:
            self._grow(ns, nt)
        b = self.batch
        h = self.stage.numpy()
        h[:ns] = source
        h[ns:ns + nt] = target
        b.raw.pts[:ns + nt].copy_(self.stage[:ns + nt], non_blocking=True)
        b.raw.off_host[:] = (0, ns, ns + nt)
        self.off_stage.numpy()[:] = b.raw.off_host
        b.raw.off.copy_(self.off_stage, non_blocking=True)
        b.max_src_n, b.max_tgt_n = ns, nt
        b.voxel_size, b.normal_k = float(voxel_size), int(normal_k)
        use_p2l = method == "point_to_line"
        have_init = R_init is not None and t_init is not None                    # icp.py:153
        b.use_p2l = use_p2l
        b.params = IcpParams(float(error_threshold), -1.0 if max_corr_dist is None else float(max_corr_dist),
                             int(max_iterations), _lib.POINT_TO_LINE if use_p2l else _lib.POINT_TO_POINT,
                             1 if have_init else 0, 2)
        if have_init:
            self.init_stage.numpy()[0, :4] = np.asarray(R_init, dtype=np.float64).reshape(4)
            self.init_stage.numpy()[0, 4:] = np.asarray(t_init, dtype=np.float64).reshape(2)
            b.init.copy_(self.init_stage, non_blocking=True)
        res = b.run()
        self.res_host.copy_(res[:1], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return self.res_host.numpy()[0].copy()


def icp_pair(source, target, error_threshold, max_iterations, voxel_size, R_init=None, t_init=None,
             method="point_to_point", normal_k=10, max_corr_dist=None):
    """One registration with the semantics of the reference ``ICP`` -> (R, t, err, info) like ``icp_batch``."""
    d = source.shape[1]
    if d == 2 and len(source) <= PREP_MAX_POINTS and len(target) <= PREP_MAX_POINTS and method in ("point_to_point", "point_to_line"):
        res = PairContext.get().solve(source, target, error_threshold, max_iterations, voxel_size, R_init, t_init, method,
                                      normal_k, max_corr_dist)
        return unpack_results(res[None, :], 2)
    return icp_batch([source], [target], error_threshold, max_iterations, voxel_size, R_init, t_init, method, normal_k,
                     max_corr_dist)


def icp_batch(sources, targets, error_threshold, max_iterations, voxel_size, R_init=None, t_init=None,
              method="point_to_point", normal_k=10, max_corr_dist=None, force_exhaustive=False):
    """Register sources[i] onto targets[i] for every i; same per-pair semantics as the reference ``ICP``.

    ``sources`` may be one array shared by every pair (the loop-closure shape,
    slam.py:576-579).  Returns (R [B,d,d], t [B,d], err [B], info).
    """
    targets = list(targets)
    B = len(targets)
    if isinstance(sources, np.ndarray) and sources.ndim == 2:
        clouds = [sources] + targets
        ps = np.zeros(B, dtype=np.int32)
        pt = np.arange(1, B + 1, dtype=np.int32)
    else:
        sources = list(sources)
        if len(sources) != B:
            raise ValueError("sources and targets differ in length")
        clouds = sources + targets
        ps = np.arange(B, dtype=np.int32)
        pt = np.arange(B, 2 * B, dtype=np.int32)
    batch = IcpBatch(clouds, ps, pt, error_threshold, max_iterations, voxel_size, R_init, t_init,
                     method, normal_k, max_corr_dist, force_exhaustive)
    batch.run()
    return batch.unpack()
