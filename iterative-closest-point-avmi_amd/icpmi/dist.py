"""Sharding a batch of independent scan pairs over the GPUs of one node.

The reference matches loop-closure candidates one after another
(slam.py:575-597); the pairs share no state, so pair i goes to rank
``i mod world`` (interleaved: iteration counts vary several-fold between pairs),
every rank registers its share with the fused ICP kernel, and the only exchange
is one all_gather of fixed 128-byte result records (RCCL over xGMI when the
process group is ``nccl``; ``gloo`` on CPU tensors in the tests).  One process
per GPU, launched by ``torch.distributed.run``.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import _lib


def _all_gather_into(out, inp, group=None):
    """all_gather_into_tensor; device tensors under a gloo group (a rehearsal of the multi-rank path on
    one GPU, or a CPU-only cluster fabric) are staged through host memory, RCCL takes them as they are."""
    if inp.is_cuda and dist.get_backend(group) == "gloo":
        o, i = torch.empty(out.shape, dtype=out.dtype), inp.cpu()
        dist.all_gather_into_tensor(o, i, group=group)
        out.copy_(o)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)


def shard(n_pairs, rank, world):
    """Indices of the pairs rank ``rank`` owns (interleaved)."""
    return np.arange(rank, n_pairs, world, dtype=np.int64)


def slots_per_rank(n_pairs, world):
    return (n_pairs + world - 1) // world


def gather_results(local, n_pairs, rank, world, group=None, force_collective=False):
    """all_gather the per-rank result records and put them back in pair order.

    force_collective: run the collective even with one rank (a world of one still goes through RCCL; the GPU test
    uses it to load the backend on the one GPU a test box has).

    local: [len(shard(...)), RES_DOUBLES] float64 tensor (device for nccl, CPU
    for gloo).  Returns an [n_pairs, RES_DOUBLES] tensor on the same device,
    identical on every rank.
    """
    per = slots_per_rank(n_pairs, world)
    width = local.shape[1] if local.dim() == 2 else _lib.RES_DOUBLES
    padded = torch.zeros((per, width), dtype=torch.float64, device=local.device)
    k = len(shard(n_pairs, rank, world))
    if k:
        padded[:k] = local[:k]
    if world == 1 and not force_collective:
        return padded[:n_pairs]
    out = torch.empty((world * per, width), dtype=torch.float64, device=local.device)
    _all_gather_into(out, padded, group)
    # record of pair i sits at rank i % world, slot i // world
    i = torch.arange(n_pairs, device=local.device)
    return out[(i % world) * per + (i // world)]


def first_accepted(results, error_accept):
    """slam.py:582-597: candidates are tried in order and the first with error below the gate wins."""
    err = results[:, _lib.RES_ERR]
    ok = torch.nonzero(err < error_accept)
    return int(ok[0]) if len(ok) else -1


def icp_batch_sharded(sources, targets, error_threshold, max_iterations, voxel_size, R_init=None, t_init=None,
                      method="point_to_point", normal_k=10, max_corr_dist=None, group=None, solver=None):
    """Register sources[i] onto targets[i] across all ranks of the process group.

    Every rank passes the same lists (or at least its own shard's entries); the
    result [n_pairs, RES_DOUBLES] is returned on every rank in pair order.
    ``solver(src_list, tgt_list, init_R, init_t) -> [k, RES_DOUBLES] tensor`` replaces the local
    GPU batch (the CPU tests inject the oracle there).
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = len(targets)
    mine = shard(n, rank, world)
    shared = isinstance(sources, np.ndarray) and sources.ndim == 2
    src = [sources if shared else sources[i] for i in mine]
    tgt = [targets[i] for i in mine]
    Ri = ti = None
    if R_init is not None and t_init is not None:
        Rb = np.asarray(R_init, dtype=np.float64)
        tb = np.asarray(t_init, dtype=np.float64)
        Ri = Rb[mine] if Rb.ndim == 3 else Rb
        ti = tb[mine] if tb.ndim == 2 else tb
    if solver is not None:
        local = solver(src, tgt, Ri, ti)
    elif len(mine) == 0:
        local = torch.zeros((0, _lib.RES_DOUBLES), dtype=torch.float64, device=torch.device("cuda", torch.cuda.current_device()))
    else:
        from .batch import IcpBatch
        k = len(mine)
        if shared:
            clouds, ps, pt = [sources] + tgt, np.zeros(k, dtype=np.int32), np.arange(1, k + 1, dtype=np.int32)
        else:
            clouds, ps, pt = src + tgt, np.arange(k, dtype=np.int32), np.arange(k, 2 * k, dtype=np.int32)
        b = IcpBatch(clouds, ps, pt, error_threshold, max_iterations, voxel_size, Ri, ti, method, normal_k, max_corr_dist)
        local = b.run()[:k]
    return gather_results(local, n, rank, world, group)


class RunIcpPairSharded:
    """The loop-closure matching of slam.py:575-597 across the ranks of the process group: candidate i of ONE current scan
    goes to rank ``i mod world``; every rank runs ``_run_icp_pair`` (slam.py:53-98: rotation search, then ICP from its
    result — ``icpmi.prealign.RunIcpPairBatch``, one chain of launches) on its share, one all_gather of the 128-byte result
    records follows, and ``first_accepted`` takes the FIRST candidate (in candidate order, as the reference's loop does)
    whose error is below the gate.  The candidates stay resident: ``run()`` may be repeated.

    ``solver(source, target_list) -> [k, RES_DOUBLES] tensor`` replaces the local GPU batch (the CPU tests inject the
    oracle there).  Slot 8 of a gathered record (unused by 2-D results) carries the status of the pair's rotation search, so
    that every rank knows which candidates fell outside the on-chip search's capacity: their owners redo them through the
    single-pair entries (same numbers) and the records are gathered once more — all ranks take part, none has to be told."""
    SEARCH_STATUS = 8

    def __init__(self, source, targets, icp_cfg=None, feat_cfg=None, group=None, max_rows_hint=0, solver=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n = len(targets)
        self.mine = shard(self.n, self.rank, self.world)
        self.source = source
        self.targets = [targets[i] for i in self.mine]
        self.icp_cfg, self.feat_cfg = dict(icp_cfg or {}), dict(feat_cfg or {})
        self.solver = solver
        self.batch = None
        self.gathered = None
        if solver is None and len(self.mine):
            from .prealign import RunIcpPairBatch
            k = len(self.mine)
            c, f = self.icp_cfg, self.feat_cfg
            self.batch = RunIcpPairBatch([source] + self.targets, np.zeros(k, dtype=np.int32), np.arange(1, k + 1, dtype=np.int32),
                                         error_threshold=c.get("error_threshold", 1e-7), max_iterations=c.get("max_iterations", 100),
                                         voxel_size=c.get("voxel_size", 0.06), method=c.get("method", "point_to_line"),
                                         normal_k=c.get("normal_k", 10), rotation_voxel_size=f.get("rotation_voxel_size", 0.3),
                                         angle_step_coarse=f.get("angle_step_coarse", 2.0), angle_step_fine=f.get("angle_step_fine", 0.2),
                                         max_rows_hint=max_rows_hint)

    def _local(self, events=None):
        k = len(self.mine)
        if self.solver is not None:
            return self.solver(self.source, self.targets)
        if k == 0:
            return torch.zeros((0, _lib.RES_DOUBLES), dtype=torch.float64, device=torch.device("cuda", torch.cuda.current_device()))
        if events is not None:
            events[0].record()
        self.batch.search.run()
        res = self.batch.icp.run(events=None if events is None else (events[1], events[2]))[:k]
        res[:, self.SEARCH_STATUS] = self.batch.search.records[:k, 11]          # device-side copy: rides on the gather
        return res

    def run(self, events=None, force_collective=False):
        """Enqueue the local chain and the gather; returns the [n, RES_DOUBLES] records in candidate order (device tensor
        under nccl, no host round trip).  events: optional 3 torch events — before the search, between search and ICP half,
        after the ICP half."""
        self.gathered = gather_results(self._local(events), self.n, self.rank, self.world, self.group, force_collective)
        return self.gathered

    def results(self):
        """(R [n,2,2], t [n,2], err [n], info) of the last run on the host (synchronises); candidates whose rotation search
        fell outside the on-chip capacity are redone by their owners and gathered again."""
        from .batch import unpack_results
        res = self.gathered
        over = torch.nonzero(res[:, self.SEARCH_STATUS] == 2.0).reshape(-1).cpu().numpy() if self.solver is None else np.empty(0, dtype=np.int64)
        if len(over):
            k = len(self.mine)
            if self.batch is not None:
                fixed = torch.from_numpy(np.ascontiguousarray(_pack_results(*self.batch.unpack()))).to(res.device)
            else:
                fixed = torch.zeros((0, _lib.RES_DOUBLES), dtype=torch.float64, device=res.device)
            res = gather_results(fixed[:k], self.n, self.rank, self.world, self.group)
            self.gathered = res
        host = res.cpu().numpy().copy()
        host[:, self.SEARCH_STATUS] = 0.0
        return unpack_results(host, 2)

    def first_accepted(self, error_accept):
        """slam.py:582-597 on the gathered records: index of the first candidate with error < error_accept, or -1."""
        return first_accepted(self.gathered, error_accept)


def _pack_results(R, t, err, info):
    res = np.zeros((len(err), _lib.RES_DOUBLES))
    d = R.shape[1]
    res[:, _lib.RES_R:_lib.RES_R + d * d] = R.reshape(len(err), d * d)
    res[:, _lib.RES_T:_lib.RES_T + d] = t
    res[:, _lib.RES_ERR] = err
    res[:, _lib.RES_DELTA] = info["delta"]
    res[:, _lib.RES_ITERS] = info["iters"]
    res[:, _lib.RES_STATUS] = info["status"]
    return res


def run_icp_pair_batch_sharded(source, targets, icp_cfg=None, feat_cfg=None, error_accept=None, group=None, solver=None):
    """``_run_icp_pair(source, targets[i], icp_cfg, feat_cfg, "rotation_search")`` for every candidate i (slam.py:575-579),
    the candidates sharded over the ranks of the process group -> (R [n,2,2], t [n,2], err [n], info) on every rank, with
    ``info["first_accepted"]`` = the candidate slam.py:582-597 would accept (first with err < error_accept; -1: none) when
    a gate is given.  Same configuration keys and defaults as the reference."""
    job = RunIcpPairSharded(source, targets, icp_cfg, feat_cfg, group=group, solver=solver)
    job.run()
    R, t, err, info = job.results()
    if error_accept is not None:
        ok = np.flatnonzero(err < error_accept)
        info["first_accepted"] = int(ok[0]) if len(ok) else -1
    return R, t, err, info


# ── sharded map replay: every rank replays all scans into its own band of rows ──────────────
def _as_tensor(a):
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))


def row_costs(ny, min_y, resolution, origins, hits):
    """Ray cells per grid row, estimated from the end points only (integer arithmetic, so every rank gets
    the same numbers): a beam that spans r rows and max(|dx|, |dy|) cells puts cells/r on each of its rows.
    The x extent is taken in the same units (cells of `resolution`), without the grid's x origin."""
    sizes = [int(h.shape[0]) for h in hits]
    if sum(sizes) == 0:
        return torch.zeros(ny, dtype=torch.int64)
    h = torch.cat([_as_tensor(x).reshape(-1, 2) for x in hits])
    dev = h.device
    org = _as_tensor(origins).reshape(-1, 2).to(dev)
    o = torch.repeat_interleave(org[: len(sizes)], torch.tensor(sizes, device=dev), dim=0)
    y0 = torch.floor((o[:, 1] - min_y) / resolution).clamp(-2.0**29, 2.0**29).to(torch.int64)
    y1 = torch.floor((h[:, 1] - min_y) / resolution).clamp(-2.0**29, 2.0**29).to(torch.int64)
    dx = torch.floor(torch.abs(h[:, 0] - o[:, 0]) / resolution).clamp(0, 2.0**29).to(torch.int64)
    lo, hi = torch.minimum(y1, y0), torch.maximum(y1, y0)
    cells = torch.maximum(hi - lo, dx) + 1
    w = (cells * 256) // (hi - lo + 1)                           # per-row share, fixed point
    lo_c, hi_c = lo.clamp(0, ny), (hi + 1).clamp(0, ny)
    keep = hi_c > lo_c
    c = torch.zeros(ny + 1, dtype=torch.int64, device=dev)
    c.index_add_(0, lo_c[keep], w[keep])
    c.index_add_(0, hi_c[keep], -w[keep])
    return torch.cumsum(c, 0)[:ny].cpu()


def row_bands(ny, world, cost=None):
    """world + 1 row boundaries 0 = b[0] <= ... <= b[world] = ny; with `cost` (per-row work) the bands carry
    equal work, otherwise equal rows."""
    if cost is None or int(cost.sum()) == 0:
        return [ny * r // world for r in range(world + 1)]
    cum = torch.cumsum(cost.to(torch.int64), 0)
    total = int(cum[-1])
    b = [0]
    for r in range(1, world):
        cut = int(torch.searchsorted(cum, torch.tensor((total * r + world - 1) // world, dtype=torch.int64)))
        b.append(min(ny, max(b[-1], cut + 1)))
    b.append(ny)
    return b


def gather_bands(band, bands, rank, world, group=None, force_collective=False):
    """all_gather the ranks' bands (rows bands[r]:bands[r+1]) into the full (ny, nx) grid, on band's device.
    force_collective: as in gather_results."""
    nx = band.shape[1]
    ny = bands[-1]
    if world == 1 and not force_collective:
        return band
    rows = max(bands[r + 1] - bands[r] for r in range(world))
    padded = torch.zeros((rows, nx), dtype=band.dtype, device=band.device)
    padded[: band.shape[0]] = band
    out = torch.empty((world * rows, nx), dtype=band.dtype, device=band.device)
    _all_gather_into(out, padded, group)
    full = torch.empty((ny, nx), dtype=band.dtype, device=band.device)
    for r in range(world):
        full[bands[r]:bands[r + 1]] = out[r * rows: r * rows + bands[r + 1] - bands[r]]
    return full


def replay_scans_sharded(grid, origins, hits, group=None, replay=None, balance=True, bands=None):
    """The map rebuild of slam.py:271-277 (reset + update_scan over the whole history) across the ranks of
    the process group: rank r replays EVERY scan but writes only rows bands[r]:bands[r+1] of the grid (rays
    are clipped to the band, so the work is shared too); one all_gather of the bands then gives every rank
    the complete grid, bit-identical to a single-GPU replay (cells are independent and keep their scan order).

    grid: utilities.mapping.OccupancyGrid2D (same geometry and starting contents on every rank).
    ``replay(row_begin, row_end) -> (rows, nx) float32 tensor`` replaces the local GPU replay in the CPU tests.
    bands: row boundaries planned earlier (``row_bands``), e.g. once for many replays of the same history.
    Returns (row boundaries, the complete grid as a tensor).
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if bands is None:                  # identical on every rank: integer arithmetic on the same inputs
        cost = row_costs(grid.ny, grid.min_y, grid.resolution, origins, hits) if balance and world > 1 else None
        bands = row_bands(grid.ny, world, cost)
    if len(bands) != world + 1 or bands[0] != 0 or bands[-1] != grid.ny:
        raise ValueError("bands must hold world + 1 row boundaries from 0 to ny")
    r0, r1 = bands[rank], bands[rank + 1]
    if replay is not None:
        band = replay(r0, r1)
    else:
        grid.update_scans(origins, hits, rows=(r0, r1))
        band = grid.device_log_odds[r0:r1]
    full = gather_bands(band, bands, rank, world, group)
    if replay is not None:
        return bands, full
    if world > 1:
        grid.device_log_odds.copy_(full)
        grid._host = None
    grid._full_clip = False
    return bands, grid.device_log_odds
