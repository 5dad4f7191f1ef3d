"""N>1 path on CPU: world_size-2 gloo processes exercise the image sharding, the host-side merge and the
max-over-ranks timing reduction that bench.py uses (the per-rank GPU work is replaced by a stand-in function —
NOT the oracle, NOT a CPU model path: it only labels which rank handled which image)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from k210_yolo_framework_amd import shard


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_images, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        def fake_detect(idx):
            return [(int(i), rank) for i in idx]            # (image id, which rank did it)
        merged = shard.run_sharded(n_images, fake_detect, dist)
        t = shard.max_over_ranks(0.010 * (rank + 1), dist)
        dist.barrier()
        q.put((rank, merged, t))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n_images', [64, 7, 1])
def test_two_rank_sharding_and_merge(n_images):
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, n_images, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, merged, t in res:
        assert [m[0] for m in merged] == list(range(n_images))             # image order restored on every rank
        assert [m[1] for m in merged] == [i % world for i in range(n_images)]  # image i -> rank i mod G
        assert abs(t - 0.020) < 1e-9                                        # slowest rank defines the step


def test_shard_indices_partition_properties():
    for n in (0, 1, 5, 32, 257):
        for w in (1, 2, 3, 8):
            parts = [shard.shard_indices(n, r, w) for r in range(w)]
            allidx = np.sort(np.concatenate(parts)) if n else np.array([], int)
            assert allidx.tolist() == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        shard.shard_indices(4, 2, 2)
    with pytest.raises(ValueError):
        shard.merge_by_image(3, [np.array([0, 1])], [['a', 'b']])
    with pytest.raises(ValueError):
        shard.merge_by_image(2, [np.array([0, 0])], [['a', 'b']])
