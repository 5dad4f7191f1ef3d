"""The real train.Trainer on two ranks (training exchange): two PROCESSES, one Trainer each, every one stepping its
half of a global batch through Trainer.step -> shard.allreduce_gradients over a torch.distributed group, graph replay on.

On the one-GPU test box both ranks share cuda:0 and the group is gloo (CUDA tensors, staged through the host by gloo); when two
devices are visible the same test also runs with one rank per device over nccl (= RCCL).

What must hold, and is asserted bit for bit:
  * the replicas' parameters and Adam moments stay identical (they start identical and apply the same summed gradient);
  * they equal a serial restatement in ONE process — two Trainer replicas whose gradient buckets are added by hand and fed to the
    same exchange / update code — so the process group moves exactly the bucket and nothing else;
  * the reported loss is the global-batch loss (sum of the ranks' data terms, which carry 1/global_batch, plus the regulariser once).
BatchNorm statistics are per replica (train.py's docstring; keras' MirroredStrategy default), so a data-parallel step is NOT the step
of one replica on the whole batch — the forward normalises each half with its own mean/variance.  The comparison with a whole-batch
Trainer is therefore stated where it is true: the loss DIVISOR.  With learning rate 0 the weights do not move, and the two ranks'
summed data loss must equal a sum over halves computed by one process with divisor = global batch (checked to fp32 summation
tolerance), while each replica's moving statistics follow its own half."""
import os
import socket

import numpy as np
import pytest

from k210_yolo_framework_amd import netspec as ns
from k210_yolo_framework_amd.helper import Helper, VOC_ANCHORS

pytestmark = pytest.mark.gpu

STEPS, GLOBAL_B, HW = 4, 8, (64, 96)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case(seed=5):
    """One network, STEPS global batches of GLOBAL_B images (fresh pictures and labels every step)."""
    spec = ns.yolo_mobilev1([*HW, 3], 3, 20, alpha=0.5)
    w = spec.init_weights(seed)
    h = Helper(None, 20, VOC_ANCHORS, [list(HW)], [list(x) for x in spec.out_hw()])
    rng = np.random.default_rng(seed)
    batches = []
    for _ in range(STEPS):
        ys = [[] for _ in spec.outputs]
        for b in range(GLOBAL_B):
            n = int(rng.integers(1, 4))
            boxes = np.stack([rng.integers(0, 20, n), rng.uniform(.2, .8, n), rng.uniform(.2, .8, n), rng.uniform(.1, .6, n),
                              rng.uniform(.1, .6, n)], 1)
            for i, lab in enumerate(h.box_to_label(boxes)):
                ys[i].append(lab)
        batches.append((rng.uniform(0, 1, (GLOBAL_B, *HW, 3)).astype(np.float32), [np.stack(y).astype(np.float32) for y in ys]))
    return spec, w, h, batches


def _state(tr):
    import torch
    torch.cuda.synchronize()
    return {k: getattr(tr, k).cpu().numpy().copy() for k in ('P', 'm', 'v')}


def _worker(rank, world, port, backend, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch
    import torch.distributed as dist
    from k210_yolo_framework_amd import shard
    from k210_yolo_framework_amd.train import Trainer
    device = rank if backend == 'nccl' else 0
    torch.cuda.set_device(device)
    kw = dict(device_id=torch.device('cuda', device)) if backend == 'nccl' else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    try:
        spec, w, h, batches = _case()
        tr = Trainer(spec, w, h.anchors, GLOBAL_B // world, lr=1e-3, decay=1e-2, device=device, world_size=world, use_graph=True)
        idx = shard.shard_indices(GLOBAL_B, rank, world)
        losses = []
        for x, yt in batches:
            xd = torch.from_numpy(x[idx]).to(tr.dev)
            losses.append(tr.step(xd, [torch.from_numpy(y[idx]).to(tr.dev) for y in yt]))
        replayed = tr._graph is not None
        mm = tr.export_weights()['conv1_bn/moving_mean']
        q.put((rank, _state(tr), losses, replayed, mm))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run_two_ranks(backend):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, backend, q)) for r in range(world)]
    for p in ps:
        p.start()
    try:
        res = sorted([q.get(timeout=600) for _ in ps], key=lambda t: t[0])
    finally:
        for p in ps:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()                                          # exactly the processes this test started
    assert all(p.exitcode == 0 for p in ps), [p.exitcode for p in ps]
    return res


def _serial_restatement():
    """Two replicas in this process; the all-reduce is a hand-written sum of the two buckets."""
    import torch
    from k210_yolo_framework_amd import shard
    from k210_yolo_framework_amd.train import Trainer
    spec, w, h, batches = _case()
    reps = [Trainer(spec, w, h.anchors, GLOBAL_B // 2, lr=1e-3, decay=1e-2, world_size=2, use_graph=True) for _ in range(2)]
    losses = []
    for x, yt in batches:
        parts = []
        for r, tr in enumerate(reps):
            idx = shard.shard_indices(GLOBAL_B, r, 2)
            parts.append(tr._loss_and_grads_replayed(torch.from_numpy(x[idx]).cuda(), [torch.from_numpy(y[idx]).cuda() for y in yt]))
        total = reps[0].G + reps[1].G
        data = [torch.stack([p[0] for p in part['layers']]).sum() for part in parts]
        for tr in reps:
            tr.exchange(reduce=lambda g: g.copy_(total))
            tr.apply_update()
        losses.append(float((data[0] + data[1]).item()) + float(parts[0]['reg'].item()))
    return [_state(tr) for tr in reps], losses, w


def _check(res):
    (_, s0, l0, g0, mm0), (_, s1, l1, g1, mm1) = res
    assert g0 and g1                                              # both ranks ended on the replayed graph
    for k in ('P', 'm', 'v'):
        assert np.array_equal(s0[k], s1[k]), k                    # replicas bit-identical
    assert [d['loss'] for d in l0] == [d['loss'] for d in l1]     # every rank reports the global loss
    assert not np.array_equal(mm0, mm1)                           # per-replica BatchNorm statistics: each saw its own half
    serial, losses, w = _serial_restatement()
    for k in ('P', 'm', 'v'):
        assert np.array_equal(serial[0][k], serial[1][k])
        assert np.array_equal(s0[k], serial[0][k]), (k, np.abs(s0[k] - serial[0][k]).max())
    assert np.allclose([d['loss'] for d in l0], losses, rtol=1e-6, atol=0)
    assert np.abs(s0['P']).sum() > 0 and np.abs(s0['m']).max() > 0
    assert l0[-1]['loss'] != l0[0]['loss'] and all(np.isfinite(d['loss']) for d in l0)


def test_two_processes_on_one_device_gloo_group():
    _check(_run_two_ranks('gloo'))


def test_two_processes_one_device_each_rccl_group():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('one device visible: the two-rank run over RCCL needs two (the gloo variant above covers the exchange here)')
    _check(_run_two_ranks('nccl'))


def test_loss_divisor_is_the_global_batch():
    """lr = 0: the summed per-rank data losses of a world-2 step equal the halves' losses computed with divisor GLOBAL_B by a
    world-1 Trainer of the half batch scaled by its own divisor, i.e. data(world 2, half) == data(world 1, half) / 2."""
    import torch
    from k210_yolo_framework_amd.train import Trainer
    spec, w, h, batches = _case()
    x, yt = batches[0]
    half = slice(0, GLOBAL_B // 2)
    xd, yd = torch.from_numpy(x[half]).cuda(), [torch.from_numpy(y[half]).cuda() for y in yt]
    one = Trainer(spec, w, h.anchors, GLOBAL_B // 2, lr=0.0, use_graph=False)
    two = Trainer(spec, w, h.anchors, GLOBAL_B // 2, lr=0.0, world_size=2, use_graph=False)
    a = one.step(xd, yd)
    b = two.step(xd, yd, reduce=lambda g: g, reduce_scalar=lambda t: t)
    assert np.isclose(b['data_loss'], a['data_loss'] / 2, rtol=1e-6)
    assert np.isclose(b['reg_loss'], a['reg_loss'], rtol=0, atol=0)
    d1 = one.G.cpu().numpy()
    d2 = two.G.cpu().numpy()
    # gradient: data part halves, the regulariser's part does not; recover the data part through a second, regulariser-free net
    reg = np.zeros_like(d1)
    for l in spec.layers:
        from k210_yolo_framework_amd.train import _is_darknet_conv, L2_WEIGHT
        if l.kind == 'conv' and _is_darknet_conv(l.name):
            off, shp = one.slots[l.name + '/kernel']
            n = int(np.prod(shp))
            reg[off:off + n] = 2 * L2_WEIGHT * one.P[off:off + n].cpu().numpy()
    assert np.allclose(d2 - reg, (d1 - reg) / 2, rtol=1e-4, atol=1e-7 * np.abs(d1).max())
    with pytest.raises(Exception, match='process group'):
        two.step(xd, yd)                                          # a world-2 Trainer without a group says so
