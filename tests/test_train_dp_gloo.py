"""Data-parallel training step on CPU (world_size-2 gloo): the exchange train.Trainer.step performs — per-rank
gradients with the GLOBAL batch as divisor, one flat sum all-reduce (shard.allreduce_gradients), the regulariser
gradient added once afterwards, identical Adam update everywhere.  The per-rank gradient comes from
oracle/train_ref.py here (the HIP kernels need a GPU; tests/test_gpu_train.py checks them against the same oracle)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from k210_yolo_framework_amd import netspec as ns, shard
from k210_yolo_framework_amd.helper import Helper, VOC_ANCHORS


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case(seed=3, B=4):
    spec = ns.yolo_mobilev1([32, 64, 3], 3, 20, alpha=0.25)
    w = spec.init_weights(seed)
    h = Helper(None, 20, VOC_ANCHORS, [[32, 64]], [list(x) for x in spec.out_hw()])
    rng = np.random.default_rng(seed)
    ys = [[] for _ in spec.outputs]
    for b in range(B):
        boxes = np.stack([rng.integers(0, 20, 2), rng.uniform(.2, .8, 2), rng.uniform(.2, .8, 2), rng.uniform(.1, .6, 2), rng.uniform(.1, .6, 2)], 1)
        for i, lab in enumerate(h.box_to_label(boxes)):
            ys[i].append(lab)
    return spec, w, h, rng.uniform(0, 1, (B, 32, 64, 3)).astype(np.float32), [np.stack(y).astype(np.float32) for y in ys]


def _flat(spec, g):
    keys = sorted(g)
    return keys, torch.from_numpy(np.concatenate([np.asarray(g[k], np.float64).ravel() for k in keys]))


def _rank_grad(spec, w, h, x, yt, idx, global_batch):
    from oracle import train_ref
    d, r, g, _, _ = train_ref.loss_and_grads(spec, w, x[idx], [y[idx] for y in yt], h.anchors, batch_size=global_batch, with_reg=False)
    return d, g


def _reg_grad(spec, w):
    from oracle import train_ref
    return {l.name + '/kernel': 2 * train_ref.L2_WEIGHT * np.asarray(w[l.name + '/kernel'], np.float64)
            for l in spec.layers if l.kind == 'conv' and train_ref._is_darknet_conv(l.name)}


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        spec, w, h, x, yt = _case()
        idx = shard.shard_indices(len(x), rank, world)
        d, g = _rank_grad(spec, w, h, x, yt, idx, len(x))
        keys, flat = _flat(spec, g)
        shard.allreduce_gradients(flat, dist)
        q.put((rank, d, flat.numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_exchange_equals_serial_sum_and_replicas_stay_identical():
    from oracle import train_ref
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=240) for _ in ps], key=lambda t: t[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    spec, w, h, x, yt = _case()
    parts = [_rank_grad(spec, w, h, x, yt, shard.shard_indices(len(x), r, world), len(x)) for r in range(world)]
    keys, serial = _flat(spec, {k: parts[0][1][k] + parts[1][1][k] for k in parts[0][1]})
    assert np.array_equal(res[0][2], res[1][2])                                  # every replica holds the same bucket
    np.testing.assert_allclose(res[0][2], serial.numpy(), rtol=1e-12, atol=1e-14)
    assert abs(res[0][1] + res[1][1] - (parts[0][0] + parts[1][0])) < 1e-9       # data loss: sum of the shard losses
    # regulariser once, then the same Adam update on every replica
    reg = _reg_grad(spec, w)
    outs = []
    for r in range(world):
        g, o = {}, 0
        for k in keys:
            n = int(np.prod(np.shape(w[k])))
            g[k] = res[r][2][o:o + n].reshape(np.shape(w[k])) + reg.get(k, 0.0)
            o += n
        outs.append(train_ref.AdamRef(5e-4).apply({k: np.asarray(w[k], np.float64) for k in keys}, g))
    for k in keys:
        assert np.array_equal(outs[0][k], outs[1][k])
    assert any(not np.array_equal(outs[0][k], np.asarray(w[k], np.float64)) for k in keys)


def test_allreduce_is_identity_for_a_single_process():
    g = torch.arange(5, dtype=torch.float32)
    assert shard.allreduce_gradients(g.clone(), None).equal(g)
