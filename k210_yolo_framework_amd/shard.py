"""Image-sharded multi-GPU inference: one process per GPU, image i -> rank i mod G, weights
replicated, NO data-path collective — detections are returned per shard and merged on the host by image id.
torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used only for the barrier / timing reduction and
for the optional host-side gather of the (tiny) detection lists."""
from __future__ import annotations

from typing import Callable, List, Sequence

import numpy as np


def shard_indices(n_images: int, rank: int, world: int) -> np.ndarray:
    """Indices of the images rank `rank` processes (round-robin keeps per-rank batches equal to within 1)."""
    if not (0 <= rank < world):
        raise ValueError(f'rank {rank} outside world {world}')
    return np.arange(rank, n_images, world)


def merge_by_image(n_images: int, per_rank_indices: Sequence[np.ndarray], per_rank_results: Sequence[Sequence]) -> List:
    """Inverse of shard_indices: results back in image order; every image must appear exactly once."""
    out = [None] * n_images
    seen = np.zeros(n_images, bool)
    for idx, res in zip(per_rank_indices, per_rank_results):
        if len(idx) != len(res):
            raise ValueError('shard / result length mismatch')
        for i, r in zip(idx, res):
            if seen[i]:
                raise ValueError(f'image {i} produced twice')
            seen[i] = True
            out[i] = r
    if not seen.all():
        raise ValueError(f'images {np.nonzero(~seen)[0].tolist()} missing')
    return out


def run_sharded(n_images: int, detect_fn: Callable[[np.ndarray], Sequence], dist=None) -> List:
    """detect_fn(indices) -> one result per index, computed on this rank's GPU.  Returns the merged list on
    every rank when `dist` (an initialised torch.distributed) is given, else runs single-process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        idx = shard_indices(n_images, 0, 1)
        return merge_by_image(n_images, [idx], [detect_fn(idx)])
    rank, world = dist.get_rank(), dist.get_world_size()
    idx = shard_indices(n_images, rank, world)
    mine = list(detect_fn(idx))
    gathered = [None] * world
    dist.all_gather_object(gathered, (idx, mine))       # host-side, a few KB of detections
    return merge_by_image(n_images, [g[0] for g in gathered], [g[1] for g in gathered])


def max_over_ranks(seconds: float, dist=None, device=None) -> float:
    """The bench contract: a step takes as long as its slowest rank."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allreduce_gradients(flat_grad, dist=None, group=None):
    """Training-step exchange: ONE sum all-reduce over the whole flat gradient bucket (a few MB for the
    MobileNet detectors — a single large ring pass is what per-link-bound xGMI wants).  Each rank's gradient already
    carries the 1/global_batch factor, so the sum is the global-batch gradient.  In place; returns the tensor."""
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return flat_grad
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return flat_grad
