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


def test_numa_binding_reads_sysfs_and_binds_to_the_gpus_node(tmp_path):
    """shard.bind_to_gpu_numa against a fake sysfs tree: the node of the GPU's PCI function, that node's cpulist intersected with the
    CPUs this process may use; unknown node (-1) or missing files change nothing (host feeding bounds the N-GPU curve)."""
    import os
    from k210_yolo_framework_amd import shard
    assert shard.parse_cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11]
    have = sorted(os.sched_getaffinity(0))
    dev = tmp_path / 'bus' / 'pci' / 'devices' / '0000:c1:00.0'
    dev.mkdir(parents=True)
    (dev / 'numa_node').write_text('1\n')
    nd = tmp_path / 'devices' / 'system' / 'node' / 'node1'
    nd.mkdir(parents=True)
    (nd / 'cpulist').write_text(f'{have[0]},900-903\n')
    info = shard.bind_to_gpu_numa(0, sysfs=str(tmp_path), pci_bus_id='0000:C1:00.0', apply=False)
    assert info == {'pci': '0000:C1:00.0', 'node': 1, 'cpus': 1}
    before = os.sched_getaffinity(0)
    try:
        info = shard.bind_to_gpu_numa(0, sysfs=str(tmp_path), pci_bus_id='0000:c1:00.0')
        assert info['cpus'] == 1 and os.sched_getaffinity(0) == {have[0]}
    finally:
        os.sched_setaffinity(0, before)
    (dev / 'numa_node').write_text('-1\n')
    assert shard.bind_to_gpu_numa(0, sysfs=str(tmp_path), pci_bus_id='0000:c1:00.0')['cpus'] is None
    assert shard.bind_to_gpu_numa(0, sysfs=str(tmp_path), pci_bus_id='0000:ff:00.0')['node'] == -1
    assert os.sched_getaffinity(0) == before


def test_bind_to_gpu_numa_on_a_faked_sysfs_tree_eight_devices_over_two_nodes(tmp_path):
    """eight ranks, GPUs 0-3 on node 0 and 4-7 on node 1 - each rank must pick the CPUs of ITS GPU's node (no 8-GPU box here:
    the sysfs tree is faked; apply=False so the test process keeps its own affinity)."""
    import os
    from k210_yolo_framework_amd import shard
    ncpu = max(os.sched_getaffinity(0)) + 1
    half = max(1, ncpu // 2)
    lists = {0: f'0-{half - 1}', 1: f'{half}-{max(half, ncpu - 1)}'}
    for node, cl in lists.items():
        d = tmp_path / 'devices' / 'system' / 'node' / f'node{node}'
        d.mkdir(parents=True)
        (d / 'cpulist').write_text(cl + '\n')
    pcis = [f'0000:{0x10 + 0x10 * g:02x}:00.0' for g in range(8)]
    for g, pci in enumerate(pcis):
        d = tmp_path / 'bus' / 'pci' / 'devices' / pci
        d.mkdir(parents=True)
        (d / 'numa_node').write_text(f'{g // 4}\n')
    allowed = set(os.sched_getaffinity(0))
    for g, pci in enumerate(pcis):
        info = shard.bind_to_gpu_numa(g, sysfs=str(tmp_path), pci_bus_id=pci.upper(), apply=False)
        want = set(shard.parse_cpulist(lists[g // 4])) & allowed
        assert info['pci'] == pci.upper() and info['node'] == g // 4
        assert info['cpus'] == (len(want) if want else None)
    assert os.sched_getaffinity(0) == allowed                             # apply=False changed nothing
    # a device sysfs does not know (container without the PCI tree) and a node whose CPUs the cpuset excludes: nothing to bind, reported as such
    assert shard.bind_to_gpu_numa(0, sysfs=str(tmp_path), pci_bus_id='0000:ff:00.0', apply=False) == {'pci': '0000:ff:00.0', 'node': -1, 'cpus': None}
    d = tmp_path / 'devices' / 'system' / 'node' / 'node1' / 'cpulist'
    d.write_text(f'{ncpu + 64}-{ncpu + 71}\n')
    assert shard.bind_to_gpu_numa(5, sysfs=str(tmp_path), pci_bus_id=pcis[5], apply=False)['cpus'] is None
