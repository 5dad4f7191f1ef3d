"""engine.Pipeline as captured hipGraphs (yk_graph_*): a replayed step gives exactly what the eager step gives, holds nothing but
kernels (the host-to-device copy of the from-host path is issued in front of the replay), and the from-host path delivers the reference's concatenated detections
(keras_inference.py:133-135) to host memory at their live size."""
import numpy as np
import pytest

from k210_yolo_framework_amd import netspec as ns
from k210_yolo_framework_amd.helper import VOC_ANCHORS

pytestmark = pytest.mark.gpu


def _net():
    spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
    return spec, spec.init_weights(seed=1)


@pytest.mark.parametrize('precision', ['f16x2', 'f16'])
def test_replayed_step_equals_eager_step_and_is_kernels_only(precision):
    import torch
    from k210_yolo_framework_amd import engine
    spec, w = _net()
    B = 8
    g = torch.Generator(device='cuda').manual_seed(11)
    frames = [torch.randint(0, 256, (B, 224, 320, 3), dtype=torch.uint8, device='cuda', generator=g) for _ in range(4)]
    eager = engine.Pipeline(spec, w, VOC_ANCHORS, max_batch=B, depth=2, precision=precision, graph=False)
    want = []
    for f in frames:
        d, c, _, i = eager.submit(f, return_index=True)
        eager.wait()
        want.append((d.cpu().numpy().copy(), c.cpu().numpy().copy(), i.cpu().numpy().copy()))
    eager.close()
    pipe = engine.Pipeline(spec, w, VOC_ANCHORS, max_batch=B, depth=2, precision=precision, graph=True)
    for rnd in range(3):                                           # round 0 captures, rounds 1-2 replay (two resident buffers per slot)
        for k in range(0, 4, 2):
            got = [pipe.submit(f, return_index=True) for f in frames[k:k + 2]]
            pipe.wait()
            for (d, c, _, i), (wd, wc, wi) in zip(got, want[k:k + 2]):
                c = c.cpu().numpy()
                assert np.array_equal(c, wc)
                for b in range(B):
                    assert np.array_equal(d[b, :c[b]].cpu().numpy(), wd[b, :wc[b]]), (rnd, k, b)
                    assert np.array_equal(i[b, :c[b]].cpu().numpy(), wi[b, :wc[b]])
    graphs = [g for s in pipe.slots for g in s.graphs.values()]
    assert len(graphs) == 4                                        # 2 slots x 2 resident caller buffers
    nl = len(pipe.plans[0].launches())
    for g in graphs:
        assert g.nodes == g.kernel_nodes, (g.nodes, g.kernel_nodes)   # no fill / copy node in a device-resident step
        assert g.nodes >= nl + 3                                   # every launch of the plan + decode, NMS, compaction
    # a third buffer per slot is copied into the slot's own input and replayed from there: same results
    extra = frames[0].clone()
    d, c, _ = pipe.submit(extra)
    pipe.wait()
    assert np.array_equal(c.cpu().numpy(), want[0][1])
    pipe.close()


def test_from_host_ticket_delivers_the_concatenated_detections():
    import torch
    from k210_yolo_framework_amd import engine
    spec, w = _net()
    B = 6
    rng = np.random.default_rng(5)
    batches = [rng.integers(0, 256, (B, 224, 320, 3), dtype=np.uint8) for _ in range(3)]
    hw = np.array([[240, 320], [224, 320], [480, 640], [100, 300], [224, 320], [375, 500]], np.float32)
    for graph in (False, True):
        pipe = engine.Pipeline(spec, w, VOC_ANCHORS, max_batch=8, depth=2, graph=graph)
        for rnd in range(2):
            for f in batches:
                fd = torch.from_numpy(f).cuda()
                d, c, _, i = pipe.submit(fd, image_hw=hw, return_index=True)
                pipe.wait()
                d, c, i = d.cpu().numpy(), c.cpu().numpy(), i.cpu().numpy()
                t = pipe.submit_host(f, image_hw=hw, return_index=True)
                rows, off, idx = t.result()
                assert off.shape == (B + 1,) and off[0] == 0 and np.array_equal(np.diff(off), c)
                assert rows.shape == (c.sum(), 6)
                for b in range(B):
                    assert np.array_equal(rows[off[b]:off[b + 1]], d[b, :c[b]]), (graph, rnd, b)
                    assert np.array_equal(idx[off[b]:off[b + 1]], i[b, :c[b]])
        if graph:
            host_graphs = [g for s in pipe.slots for k, g in s.graphs.items() if k[2]]
            assert host_graphs and all(g.nodes == g.kernel_nodes for g in host_graphs)       # kernels only: the H2D copy goes in front of the replay
        # frames staged by the caller in the slot's pinned buffer: no host-side copy at all
        i0 = pipe.next_slot()
        pipe.host_input(i0)[:B].copy_(torch.from_numpy(batches[0]))
        rows2, off2 = pipe.submit_host(None, batch=B, image_hw=hw).result()
        d, c, _ = pipe.submit(torch.from_numpy(batches[0]).cuda(), image_hw=hw)
        pipe.wait()
        assert np.array_equal(np.diff(off2), c.cpu().numpy())
        pipe.close()


def test_staged_host_copy_gives_the_same_rows_and_is_dropped_when_frames_are_passed():
    """Pipeline.stage_host starts a slot's host -> device copy ahead of its submit (bench.py's from-host leg does this one submit ahead).
    The rows must equal those of an unstaged submit; frames handed over at submit time replace a copy staged from the buffer's old contents;
    a staged batch size that differs from the submit's is not used."""
    import torch
    from k210_yolo_framework_amd import engine
    spec, w = _net()
    B = 5
    rng = np.random.default_rng(11)
    batches = [rng.integers(0, 256, (B, 224, 320, 3), dtype=np.uint8) for _ in range(4)]
    for graph in (False, True):
        pipe = engine.Pipeline(spec, w, VOC_ANCHORS, max_batch=8, depth=2, graph=graph)
        want = []
        for f in batches:
            rows, off = pipe.submit_host(f).result()
            want.append((rows.copy(), off.copy()))
        assert sum(len(r) for r, _ in want) > 0
        for rnd in range(2):
            # the producer fills the slot AFTER the next one and stages it, then submits the next one: one copy always runs ahead
            tickets, got = [], {}
            i0 = pipe.next_slot()
            pipe.host_input(i0)[:B].copy_(torch.from_numpy(batches[0]))
            pipe.stage_host(i0, batch=B)
            for k in range(len(batches)):
                if k + 1 < len(batches):
                    i1 = pipe.next_slot() + 1
                    if k + 1 >= pipe.depth:
                        got[k + 1 - pipe.depth] = tuple(a.copy() for a in tickets[k + 1 - pipe.depth].result())   # consume the slot's previous batch before refilling it
                    pipe.host_input(i1)[:B].copy_(torch.from_numpy(batches[k + 1]))
                    pipe.stage_host(i1, batch=B)
                tickets.append(pipe.submit_host(None, batch=B))
            for k, t in enumerate(tickets):
                rows, off = got[k] if k in got else t.result()
                assert np.array_equal(off, want[k][1]) and np.array_equal(rows, want[k][0]), (graph, rnd, k)
        # a staged copy of stale contents is dropped when the submit brings its own frames
        i0 = pipe.next_slot()
        pipe.host_input(i0)[:B].copy_(torch.from_numpy(batches[3]))
        pipe.stage_host(i0, batch=B)
        rows, off = pipe.submit_host(batches[1]).result()
        assert np.array_equal(off, want[1][1]) and np.array_equal(rows, want[1][0])
        # a copy staged for another batch size is not used: the submit copies what it needs itself
        i0 = pipe.next_slot()
        pipe.host_input(i0)[:B].copy_(torch.from_numpy(batches[2]))
        pipe.stage_host(i0, batch=2)
        rows, off = pipe.submit_host(None, batch=B).result()
        assert np.array_equal(off, want[2][1]) and np.array_equal(rows, want[2][0])
        pipe.close()


def test_letterboxing_pipeline_equals_letterbox_then_run():
    import torch
    from k210_yolo_framework_amd import engine
    spec, w = _net()
    B = 4
    g = torch.Generator(device='cuda').manual_seed(2)
    cam = torch.randint(0, 256, (B, 240, 320, 3), dtype=torch.uint8, device='cuda', generator=g)
    plan = engine.Plan(spec, w, max_batch=B, schedule='throughput')                  # what a Pipeline of depth 2 builds
    cfg = engine.make_decode_cfg(VOC_ANCHORS, 20, spec.in_hw, spec.out_hw())
    plan.run_u8(engine.letterbox_u8(cam, (224, 320)))
    d0, c0 = engine.decode_py(cfg, plan.outputs(), B, None, 0.7, 0.5)
    torch.cuda.synchronize()
    d0, c0 = d0.cpu().numpy(), c0.cpu().numpy()
    pipe = engine.Pipeline(spec, w, VOC_ANCHORS, max_batch=B, depth=2, src_hw=(240, 320))
    for _ in range(3):
        d, c, _ = pipe.submit(cam)
        pipe.wait()
        assert np.array_equal(c.cpu().numpy(), c0)
        for b in range(B):
            assert np.array_equal(d[b, :c0[b]].cpu().numpy(), d0[b, :c0[b]])
    rows, off = pipe.submit_host(cam.cpu().numpy()).result()
    assert np.array_equal(np.diff(off), c0)
    pipe.close()
    plan.close()


def test_capture_refuses_a_cold_stream():
    """A capture cannot allocate: yk_scratch must say so instead of invalidating the recording silently."""
    import ctypes as C
    import torch
    from k210_yolo_framework_amd import engine
    spec, w = _net()
    plan = engine.Plan(spec, w, max_batch=2)
    cfg = engine.make_decode_cfg(VOC_ANCHORS, 20, spec.in_hw, spec.out_hw())
    st = torch.cuda.Stream()
    frames = torch.zeros((2, 224, 320, 3), dtype=torch.uint8, device='cuda')
    with torch.cuda.stream(st):
        plan.run_u8(frames)                                              # the network has run on this stream, the decode has not
        probe = [torch.empty((2, 600, 6), device='cuda'), torch.empty((2,), dtype=torch.int32, device='cuda')]
        del probe                                                        # torch's own blocks for the decode outputs are cached now
    torch.cuda.synchronize()

    def issue():
        with torch.cuda.stream(st):
            plan.run_u8(frames)
            engine.decode_py(cfg, plan.outputs(), 2, None, 0.7, 0.5)    # first decode on this stream: needs its scratch

    with pytest.raises(engine.YkError, match='capturing'):
        engine.capture(C.c_void_p(st.cuda_stream), issue)
    torch.cuda.synchronize()
    with torch.cuda.stream(st):                                          # the stream is usable again, and warm after one eager step
        plan.run_u8(frames)
        engine.decode_py(cfg, plan.outputs(), 2, None, 0.7, 0.5)
    st.synchronize()
    g = engine.capture(C.c_void_p(st.cuda_stream), issue)
    assert g.nodes == g.kernel_nodes > 10
    g.launch(C.c_void_p(st.cuda_stream))
    st.synchronize()
    g.close()
    plan.close()


def test_batch_sizes_small_max_small_replay_correctly():
    """Captured steps of different batch sizes share the slot's decode scratch: the pipeline sizes it once for max_batch x max_out (a
    warm-up step at the largest shape) so that a larger batch later cannot move the buffer under a graph captured at a smaller one;
    should it move anyway (yk_scratch_generation), the slot's graphs are dropped and captured again."""
    import torch
    from k210_yolo_framework_amd import engine
    spec, w = _net()
    g = torch.Generator(device='cuda').manual_seed(3)
    frames = torch.randint(0, 256, (16, 224, 320, 3), dtype=torch.uint8, device='cuda', generator=g)
    eager = engine.Pipeline(spec, w, VOC_ANCHORS, max_batch=16, depth=1, graph=False)
    want = {}
    for B in (3, 16):
        d, c, _ = eager.submit(None if False else frames[:B].contiguous())
        eager.wait()
        want[B] = (d.cpu().numpy().copy(), c.cpu().numpy().copy())
    eager.close()
    pipe = engine.Pipeline(spec, w, VOC_ANCHORS, max_batch=16, depth=1, graph=True)
    small, big = frames[:3].contiguous(), frames
    for B, f in ((3, small), (16, big), (3, small), (16, big), (3, small)):
        gen0 = pipe.slots[0].scratch_gen
        d, c, _ = pipe.submit(f)
        pipe.wait()
        wd, wc = want[B]
        assert np.array_equal(c.cpu().numpy(), wc), B
        for b in range(B):
            assert np.array_equal(d[b, :wc[b]].cpu().numpy(), wd[b, :wc[b]]), (B, b)
        if gen0:
            assert pipe.slots[0].scratch_gen == gen0                     # the warm slot's scratch never moves again
    assert len(pipe.slots[0].graphs) == 2                                # one capture per batch size, both replayed
    # thresholds are baked into a capture: a sweep captures per value but keeps at most GRAPH_CACHE of them
    for t in np.linspace(0.3, 0.9, engine.Pipeline.GRAPH_CACHE + 3):
        pipe.submit(small, obj_thresh=float(t))
    pipe.wait()
    assert len(pipe.slots[0].graphs) == engine.Pipeline.GRAPH_CACHE
    d, c, _ = pipe.submit(small)                                         # evicted or not, the default step still gives the same rows
    pipe.wait()
    assert np.array_equal(c.cpu().numpy(), want[3][1])
    pipe.close()


def test_device_side_failure_of_a_cluster_launch_is_raised_where_results_are_consumed():
    """The latency schedule's cluster launches give up after a bounded spin and set the plan's sticky error word (mapped host memory):
    Pipeline.wait() / Ticket.result() / Plan.check() raise instead of handing out garbage; the report clears the flag."""
    import ctypes as C
    import torch
    from k210_yolo_framework_amd import engine
    spec, w = _net()
    pipe = engine.Pipeline(spec, w, VOC_ANCHORS, max_batch=2, depth=1)
    assert pipe.schedule == 'latency'
    frames = torch.randint(0, 256, (2, 224, 320, 3), dtype=torch.uint8, device='cuda')
    pipe.submit(frames)
    pipe.wait()                                                           # a healthy run: nothing raised
    e = C.c_uint()
    engine._check(engine.lib().yk_plan_peek_error(pipe.plans[0]._h, 0, C.byref(e)), 'peek')
    assert e.value == 0
    engine._check(engine.lib().yk_plan_debug_set_error(pipe.plans[0]._h, 1), 'set')   # what a timed-out cluster barrier stores
    with pytest.raises(engine.YkError, match='cluster'):
        pipe.wait()
    pipe.wait()                                                           # reported once, then clear
    engine._check(engine.lib().yk_plan_debug_set_error(pipe.plans[0]._h, 1), 'set')
    t = pipe.submit_host(frames.cpu().numpy())
    with pytest.raises(engine.YkError, match='cluster'):
        t.result()
    pipe.plans[0].check()
    pipe.close()
