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


# ---- host placement: one process per GPU, its feeding threads and pinned buffers on the GPU's own NUMA node ------------------------------
# "expected scaling limit = host feeding".  Eight ranks each copy ~7 MB of u8 frames per batch out of pinned host memory; with
# the default placement a rank's pinned ring can sit on the other socket and every H2D copy crosses the inter-socket link.  Linux places
# pages on the node of the thread that first touches them, so binding the PROCESS to the CPUs of the GPU's node before it allocates anything
# puts its pinned buffers, its producer threads and its submit loop next to the GPU's PCIe root.
def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' -> [0,1,2,3,8,10,11] (the format of /sys/devices/system/node/nodeN/cpulist)."""
    cpus: List[int] = []
    for part in text.strip().split(','):
        if not part:
            continue
        if '-' in part:
            a, b = part.split('-')
            cpus.extend(range(int(a), int(b) + 1))
        else:
            cpus.append(int(part))
    return cpus


def gpu_numa_node(pci_bus_id: str, sysfs: str = '/sys') -> int:
    """NUMA node of the PCI device 'dddd:bb:dd.f' (-1: unknown / single-node box)."""
    import os
    path = os.path.join(sysfs, 'bus', 'pci', 'devices', pci_bus_id.lower(), 'numa_node')
    try:
        with open(path) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return -1


def bind_to_gpu_numa(device_index: int, sysfs: str = '/sys', pci_bus_id: str = None, apply: bool = True) -> dict:
    """Bind this process (and every thread it starts later) to the CPUs of the NUMA node `device_index`'s GPU hangs off.  Call it first
    thing in a rank, before pinned buffers are allocated.  Returns {'pci', 'node', 'cpus'} (node -1 / cpus None: nothing was changed - a
    single-node host, a container without sysfs, or a cpuset that excludes the node's CPUs)."""
    import os
    if pci_bus_id is None:
        import torch
        pr = torch.cuda.get_device_properties(device_index)
        pci_bus_id = f'{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0'
    node = gpu_numa_node(pci_bus_id, sysfs)
    info = {'pci': pci_bus_id, 'node': node, 'cpus': None}
    if node < 0:
        return info
    try:
        with open(os.path.join(sysfs, 'devices', 'system', 'node', f'node{node}', 'cpulist')) as f:
            cpus = set(parse_cpulist(f.read()))
    except OSError:
        return info
    allowed = cpus & set(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else cpus
    if not allowed:
        return info
    if apply and hasattr(os, 'sched_setaffinity'):
        os.sched_setaffinity(0, allowed)
    info['cpus'] = len(allowed)
    return info
