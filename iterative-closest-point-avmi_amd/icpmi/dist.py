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


def shard(n_pairs, rank, world):
    """Indices of the pairs rank ``rank`` owns (interleaved)."""
    return np.arange(rank, n_pairs, world, dtype=np.int64)


def slots_per_rank(n_pairs, world):
    return (n_pairs + world - 1) // world


def gather_results(local, n_pairs, rank, world, group=None):
    """all_gather the per-rank result records and put them back in pair order.

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
    if world == 1:
        return padded[:n_pairs]
    out = torch.empty((world * per, width), dtype=torch.float64, device=local.device)
    dist.all_gather_into_tensor(out, padded, group=group)
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
