"""End to end on the GPU: u8 frames -> conv stack -> Python-mode decode + NMS, vs the CPU oracle chain
(normalise -> yolo_net_ref fp32 / fp16-emulating -> decode_ref), at BASELINE's headline config and size.

North-star tolerance (BASELINE.json): class / box indices exact, scores and box coords within 1e-3 of the fp32 path.
  * precision 'f16x2' (fp32 activations, compensated fp16 MFMA operands) is held to exactly that, on all 32 images of K2 with the
    undamped SURVEY 8(d) weights: the same detections in the same order (class sequence and count per image - NMS survivors
    are box identities), every score within 1e-3 (measured ~3e-6), every box corner within 1e-3 of the box size.
  * precision 'f16' (fp16 activation storage, the throughput mode) cannot meet the max bound by construction - 20 layers of
    2^-11 roundings, error budget in DESIGN.md 4 - and is held to that budget: mean score error <= 1e-3, max <= TOL_MAX_F16,
    >= 97 % of the detections reproduced away from the threshold margin."""
import contextlib
import io

import numpy as np
import pytest

import oracle
from oracle import decode_ref as dr
from k210_yolo_framework_amd import netspec as ns
from k210_yolo_framework_amd.helper import VOC_ANCHORS

pytestmark = pytest.mark.gpu
MARGIN = 6e-3          # f16 mode only: detections this close to the obj threshold are not decidable at fp16 storage precision
TOL_MAX = 5e-3         # f16 mode only (TOL_MAX_F16): worst-case score drift allowed by the error budget
NORTH_STAR = 1e-3      # f16x2 mode: BASELINE.json's tolerance, asserted as a MAX over every detection


def _gpu(spec, w, frames, obj, iou, image_hw=None, precision='f16', schedule='latency', anchors=None):
    import torch
    from k210_yolo_framework_amd import engine
    B = frames.shape[0]
    plan = engine.Plan(spec, w, max_batch=B, precision=precision, schedule=schedule)
    plan.run_u8(torch.from_numpy(frames).cuda())
    cfg = engine.make_decode_cfg(VOC_ANCHORS if anchors is None else anchors, spec.class_num, spec.in_hw, spec.out_hw())
    dets, counts, index = engine.decode_py(cfg, plan.outputs(), B, image_hw, obj, iou, return_index=True)
    torch.cuda.synchronize()
    outs = [o[:B].cpu().numpy() for o in plan.outputs()]
    d, c, ix = dets.cpu().numpy(), counts.cpu().numpy(), index.cpu().numpy()
    plan.close()
    _gpu.last_index = [ix[b, :c[b]] for b in range(B)]
    return outs, [d[b, :c[b]] for b in range(B)]


def _match(got, ref_dets, ref_scores_all, obj, hw):
    """Pair detections by (class, nearest box) — positional pairing breaks when two near-equal scores swap
    order.  Returns (#paired, #ref, #got).  A pair needs: same class, score within TOL_MAX, corners within
    2% of the box size.  Threshold-margin detections are set aside first."""
    r = ref_dets[np.abs(ref_dets[:, 4] - obj) > MARGIN]
    g = got[np.abs(got[:, 4] - obj) > MARGIN]
    used = np.zeros(len(g), bool)
    paired, es_all = 0, []
    for d in r:
        cand = np.nonzero((g[:, 5] == d[5]) & ~used)[0]
        if not len(cand):
            continue
        size = max(d[2] - d[0], d[3] - d[1], 1.0)
        eb = np.abs(g[cand, :4] - d[:4]).max(1) / size
        j = cand[np.argmin(eb)]
        if eb.min() <= 2e-2 and abs(g[j, 4] - d[4]) <= TOL_MAX:
            used[j] = True
            paired += 1
            es_all.append(abs(g[j, 4] - d[4]))
    if es_all:
        assert np.mean(es_all) <= 1e-3, np.mean(es_all)          # the north-star figure holds on average
    return paired, len(r), len(g)


def _assert_exact_indices(index, ref, tag):
    """BASELINE north_star "class/box indices bit-exact": per image the SET of (class, box index) pairs the GPU selected equals the
    fp32 oracle's (keras_inference.py:116-131: boolean_mask + non_max_suppression + gather); inside a class the order is by score,
    which is only defined up to the logit error, so the comparison is on sets, and additionally on sequences wherever the scores of
    neighbouring detections differ by more than that error."""
    n = 0
    for b, (gi, (rd, ri)) in enumerate(zip(index, ref)):
        assert len(gi) == len(ri), (tag, b, len(gi), len(ri))
        got = sorted(zip(_gpu.last_dets_cls[b].tolist(), gi.tolist()))
        want = sorted(zip(rd[:, 5].astype(int).tolist(), ri.tolist()))
        assert got == want, (tag, b, [x for x in got if x not in want][:5], [x for x in want if x not in got][:5])
        n += len(ri)
    return n


def _assert_north_star(dets, ref_dets, tag):
    """The same SET of detections (class, box) per image; scores within 1e-3; corners within 1e-3 of the box size (w,h = exp(t)*anchor
    scales the logit error by the box size itself; boxes are in pixels of the original image).  Rows are compared in place; where two
    detections of a class have scores closer than the logit error their order inside the class is not defined (350 random-weight
    detections per image: it happens), so rows that differ in place are paired one-to-one by class and nearest box instead."""
    n = 0
    for b, (g, r) in enumerate(zip(dets, ref_dets)):
        assert len(g) == len(r), (tag, b, len(g), len(r))
        if not len(r):
            continue
        assert np.array_equal(g[:, 5], r[:, 5]), (tag, b)                       # class of every detection, class-major order
        size = np.maximum(np.maximum(r[:, 2] - r[:, 0], r[:, 3] - r[:, 1]), 1.0)
        same = (np.abs(g[:, :4] - r[:, :4]).max(1) / size <= NORTH_STAR) & (np.abs(g[:, 4] - r[:, 4]) <= NORTH_STAR)
        free = list(np.nonzero(~same)[0])
        assert len(free) <= max(2, len(r) // 50), (tag, b, len(free))           # reordering is the exception
        for i in np.nonzero(~same)[0]:
            cand = [j for j in free if g[j, 5] == r[i, 5]]
            assert cand, (tag, b, i)
            eb = [np.abs(g[j, :4] - r[i, :4]).max() / size[i] for j in cand]
            j = cand[int(np.argmin(eb))]
            assert min(eb) <= NORTH_STAR and abs(g[j, 4] - r[i, 4]) <= NORTH_STAR, (tag, b, i, min(eb), abs(g[j, 4] - r[i, 4]))
            free.remove(j)
        n += len(r)
    return n


_K2_REF = {}


def _k2_reference():
    """configs[1] at full size: spec, weights, the 32 seeded frames, the fp32 oracle's logits and detections (computed once per session)."""
    if not _K2_REF:
        spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
        w = spec.init_weights(seed=1)                                           # undamped SURVEY 8(d) initialisation
        frames = np.random.default_rng(0).integers(0, 256, (32, 224, 320, 3), dtype=np.uint8)
        ref32 = oracle.net_forward(spec.compile_plan(w), oracle.normalise_u8(frames), emulate_f16=False, out_ids=spec.outputs)
        rd = dr.decode_batch([r.reshape(32, r.shape[1], r.shape[2], 3, 25) for r in ref32], VOC_ANCHORS, (224, 320), (224, 320), 0.7, 0.5)
        _K2_REF.update(spec=spec, w=w, frames=frames, ref32=ref32, rd=rd)
    return _K2_REF


@pytest.mark.parametrize('schedule', ['latency', 'throughput'])
def test_north_star_tolerance_headline_config_all_32_images(schedule):
    """BASELINE configs[1] at full size in the f16x2 mode vs the FP32 oracle: indices exact, scores / coords within 1e-3 (max) - for BOTH
    launch schedules of the plan (keras_inference.py:113-135 is what the detections are held to)."""
    k = _k2_reference()
    spec, w, frames = k['spec'], k['w'], k['frames']
    outs, dets = _gpu(spec, w, frames, 0.7, 0.5, precision='f16x2', schedule=schedule)
    ref32, rd = k['ref32'], k['rd']
    for g, r in zip(outs, ref32):
        assert np.abs(g - r).max() <= 1e-4 * np.abs(r).max()                    # logits: measured 2e-6 relative
    n = _assert_north_star(dets, [x[0] for x in rd], f'K2 f16x2 {schedule}')
    assert n > 1000, n                                                          # ~350 detections per image on these weights
    _gpu.last_dets_cls = [d[:, 5].astype(int) for d in dets]
    assert _assert_exact_indices(_gpu.last_index, rd, f'K2 f16x2 {schedule}') == n   # the box INDEX of every detection, exact


@pytest.mark.parametrize('from_host', [False, True])
def test_north_star_tolerance_on_the_benched_pipeline_replayed_graphs(from_host):
    """The plan behind bench.py's `value` / `value_from_host`, exactly as bench.Harness drives it: engine.Pipeline(depth=4, graph=True)
    -> throughput schedule, every slot's step a REPLAYED hipGraph, B=32, four batches in flight - and every slot's detections of a
    replayed step are held to the north-star bar (indices exact, scores / coords within 1e-3) against the fp32 oracle."""
    import torch
    from k210_yolo_framework_amd import engine
    k = _k2_reference()
    spec, w, frames, rd = k['spec'], k['w'], k['frames'], k['rd']
    pipe = engine.Pipeline(spec, w, VOC_ANCHORS, max_batch=32, depth=4, precision='f16x2', graph=True)
    assert pipe.schedule == 'throughput' and all(p.schedule == 'throughput' for p in pipe.plans)
    d_frames = torch.from_numpy(frames).cuda()
    if from_host:
        for i in range(4):
            pipe.host_input(i).copy_(torch.from_numpy(frames))
    res = []
    for it in range(12):                                                        # slot i: eager warm-up + capture on its first turn, replays after
        if from_host:
            res.append(pipe.submit_host(None, return_index=True))
        else:
            res.append(pipe.submit(d_frames, sync_input=False, return_index=True))
            if it >= 8:                                                         # the last round: keep each slot's results before the slot is reused
                dets, counts, st, index = res[-1]
                st.synchronize()
                res[-1] = (dets.cpu().numpy(), counts.cpu().numpy(), index.cpu().numpy())
    # replayed, not eager: one captured step per slot - two from host, where a slot's batches alternate between its two device input buffers
    assert all(len(sl.graphs) == (2 if from_host else 1) and all(g.kernel_nodes >= 20 for g in sl.graphs.values()) for sl in pipe.slots)
    for slot, r in enumerate(res[8:]):
        if from_host:
            rows, off, ix = r.result()
            dets = [rows[off[b]:off[b + 1]] for b in range(32)]
            index = [ix[off[b]:off[b + 1]] for b in range(32)]
        else:
            d, c, ix = r
            dets = [d[b, :c[b]] for b in range(32)]
            index = [ix[b, :c[b]] for b in range(32)]
        n = _assert_north_star(dets, [x[0] for x in rd], f'pipeline slot {slot}')
        assert n > 1000
        _gpu.last_dets_cls = [d[:, 5].astype(int) for d in dets]
        assert _assert_exact_indices(index, rd, f'pipeline slot {slot}') == n
    pipe.wait()
    pipe.close()


@pytest.mark.parametrize('schedule', ['latency', 'throughput'])
@pytest.mark.parametrize('name,shape,alpha,B', [('yolo_mobilev2', (224, 320, 3), 1.0, 4), ('tiny_yolo', (416, 416, 3), 1.0, 2),
                                                ('yolo', (416, 416, 3), 1.0, 4)])      # configs[4]: Darknet-53 416x416, three scales (yolonet.py:161-191)
def test_north_star_tolerance_other_networks(name, shape, alpha, B, schedule):
    if name == 'yolo' and schedule == 'latency':
        pytest.skip('Darknet-53 has no cluster launches: both schedules build the same plan')
    spec = ns.NETWORKS[name](shape, 3, 20, alpha=alpha)
    w = spec.init_weights(seed=1)                                               # undamped, also for MobileNet-v2
    frames = np.random.default_rng(5).integers(0, 256, (B, *shape), dtype=np.uint8)
    anchors = VOC_ANCHORS
    if name == 'yolo':
        # three scales -> a third anchor layer; and 75 undamped He-normal layers put the logits at ~1e5, where every sigmoid is 0 or 1 and
        # exp(wh) overflows: the three OUTPUT convs' kernels are scaled so that the logits are O(1) (the stack in front of them stays undamped)
        anchors = np.concatenate([VOC_ANCHORS, VOC_ANCHORS[:1] * 0.5])
        r0 = oracle.net_forward(spec.compile_plan(w), oracle.normalise_u8(frames[:1]), emulate_f16=False, out_ids=spec.outputs)
        out_layers = [op['layer'] for op in spec.ops if op['type'] == ns.OP_CONV and op['flags'] & ns.FLAG_NET_OUTPUT]
        assert len(out_layers) == 3
        for lname, r in zip(out_layers, r0):
            w[lname + '/kernel'] = (w[lname + '/kernel'] * (8.0 / float(np.percentile(np.abs(r), 99)))).astype(np.float32)   # ~280 detections per image
    outs, dets = _gpu(spec, w, frames, 0.7, 0.5, precision='f16x2', schedule=schedule, anchors=anchors)
    ref32 = oracle.net_forward(spec.compile_plan(w), oracle.normalise_u8(frames), emulate_f16=False, out_ids=spec.outputs)
    rd = dr.decode_batch([r.reshape(B, r.shape[1], r.shape[2], 3, 25) for r in ref32], anchors, shape[:2], shape[:2], 0.7, 0.5)
    assert _assert_north_star(dets, [x[0] for x in rd], name) > 50
    _gpu.last_dets_cls = [d[:, 5].astype(int) for d in dets]
    assert _assert_exact_indices(_gpu.last_index, rd, name) > 50


def test_make_inference_cli_prints_the_reference_table(tmp_path, capsys):
    """BASELINE configs[0] (`make inference MODEL=yolo_mobilev1 DEPTHMUL=0.75 CKPT=... IMG=...`): the CLI end to end - Keras .h5
    checkpoint in, `[top left bottom right score class]` rows out (keras_inference.py:146,154) - against the oracle chain
    Helper._read_img -> letterbox -> img/max -> fp32 conv stack -> decode_ref on the same file."""
    from k210_yolo_framework_amd import inference, keras_io
    from k210_yolo_framework_amd.helper import Helper
    spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
    w = spec.init_weights(seed=1)
    ck = tmp_path / 'yolo_model.h5'
    keras_io.save_keras_weights(spec, w, ck)
    img_path = 'data/synthetic_320x224.jpg'
    argv = [str(ck), img_path, '--model_def', 'yolo_mobilev1', '--depth_multiplier', '0.75', '--obj_thresh', '0.7', '--iou_thresh', '0.5',
            '--image_size', '224', '320', '--output_size', '7', '10', '14', '20', '--train_set', 'voc', '--class_num', '20']
    got = inference.cli(argv)
    out = capsys.readouterr().out
    assert f'[ INFO  ]  Load CKPT {ck}' in out
    h = Helper(None, 20, VOC_ANCHORS, [[224, 320]], [[7, 10], [14, 20]])
    orig = h._read_img(img_path)
    x, _ = h._process_img(orig, None, is_training=False, is_resize=True)
    ref32 = oracle.net_forward(spec.compile_plan(w), x[None].astype(np.float32), emulate_f16=False, out_ids=spec.outputs)
    rd = dr.decode_batch([r.reshape(1, r.shape[1], r.shape[2], 3, 25) for r in ref32], VOC_ANCHORS, (224, 320), orig.shape[:2], 0.7, 0.5)[0][0]
    assert len(rd) > 0 and _assert_north_star([got], [rd], 'cli') == len(rd)
    lines = [l for l in out.splitlines() if l.startswith('[') and not l.startswith('[ ')]
    assert lines[0] == '[top\tleft\tbottom\tright\tscore\tclass]'
    want = [f'[{t:.1f}\t{l:.1f}\t{b:.1f}\t{r:.1f}\t{s:.2f}\t{int(c):2d}]' for t, l, b, r, s, c in rd]
    assert len(lines) == 1 + len(want)
    # the printed rows themselves: same format, same class column; a printed digit may differ where the value sits on a rounding
    # boundary of the one-decimal format (the synthetic weights produce boxes thousands of pixels wide: 1e-3 of the box size > 0.05)
    for a, b in zip(lines[1:], want):
        fa, fb = a.strip('[]').split('\t'), b.strip('[]').split('\t')
        assert len(fa) == 6 and fa[5] == fb[5], (a, b)
        va, vb = np.array(fa[:5], float), np.array(fb[:5], float)
        size = max(vb[2] - vb[0], vb[3] - vb[1], 1.0)
        assert np.abs(va[:4] - vb[:4]).max() <= 0.1 + NORTH_STAR * size and abs(va[4] - vb[4]) <= 0.011, (a, b)
    same = sum(a == b for a, b in zip(lines[1:], want))
    assert same >= 0.8 * len(want), (same, len(want))
    # no detections -> the reference's note (keras_inference.py:176)
    argv2 = list(argv)
    argv2[argv2.index('--obj_thresh') + 1] = '0.99999999'
    assert len(inference.cli(argv2)) == 0
    assert 'no boxes detected' in capsys.readouterr().out


@pytest.mark.parametrize('B', [4, 32])
def test_headline_config_end_to_end(B):
    spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
    w = spec.init_weights(seed=1)
    frames = np.random.default_rng(0).integers(0, 256, (B, 224, 320, 3), dtype=np.uint8)
    outs, dets = _gpu(spec, w, frames, 0.7, 0.5)
    nchk = min(B, 8)                                      # oracle forward is ~0.1 s/img
    x = oracle.normalise_u8(frames[:nchk])
    plan = spec.compile_plan(w)
    ref32 = oracle.net_forward(plan, x, emulate_f16=False, out_ids=spec.outputs)
    ref16 = oracle.net_forward(plan, x, emulate_f16=True, out_ids=spec.outputs)
    for name, ref in (('fp16-emulating', ref16), ('fp32', ref32)):
        rd = dr.decode_batch([r.reshape(nchk, r.shape[1], r.shape[2], 3, 25) for r in ref], VOC_ANCHORS, (224, 320),
                             (224, 320), 0.7, 0.5)
        paired = nref = ngot = 0
        for b in range(nchk):
            p_, r_, g_ = _match(dets[b], rd[b][0], None, 0.7, (224, 320))
            paired, nref, ngot = paired + p_, nref + r_, ngot + g_
        assert nref > 0, 'synthetic weights must produce detections (conf bias -4, SURVEY 8(d))'
        # NMS is discontinuous: a score swap or an IoU within fp16 drift of the threshold changes survivors.
        # >= 97 % of the reference detections must be reproduced (and vice versa); measured ~99 %.
        assert paired >= 0.97 * nref and paired >= 0.97 * ngot, (name, paired, nref, ngot)
    # decode of the GPU's own logits must agree with the oracle decode exactly (pure decode parity at full size)
    rd = dr.decode_batch([o.reshape(B, o.shape[1], o.shape[2], 3, 25) for o in outs], VOC_ANCHORS, (224, 320),
                         (224, 320), 0.7, 0.5)
    for b in range(B):
        assert len(dets[b]) == len(rd[b][0])
        assert np.array_equal(dets[b][:, 5], rd[b][0][:, 5])
        np.testing.assert_allclose(dets[b][:, :5], rd[b][0][:, :5], rtol=1e-5, atol=1e-3)


def test_size_independent_properties_full_batch():
    """Properties that hold at any size: per-image independence (shuffling the batch permutes the
    detections), determinism (bitwise equal reruns), and NMS idempotence on the emitted boxes."""
    spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
    w = spec.init_weights(seed=1)
    frames = np.random.default_rng(7).integers(0, 256, (32, 224, 320, 3), dtype=np.uint8)
    o1, d1 = _gpu(spec, w, frames, 0.7, 0.5)
    o2, d2 = _gpu(spec, w, frames, 0.7, 0.5)
    perm = np.random.default_rng(1).permutation(32)
    o3, d3 = _gpu(spec, w, np.ascontiguousarray(frames[perm]), 0.7, 0.5)
    for a, b in zip(o1, o2):
        np.testing.assert_array_equal(a, b)
    for i, p in enumerate(perm):
        np.testing.assert_array_equal(d3[i], d1[p])
        np.testing.assert_array_equal(d1[p], d2[p])
    for d in d1:
        for c in np.unique(d[:, 5]):
            k = d[d[:, 5] == c]
            assert dr.non_max_suppression(k[:, :4], k[:, 4], 30, 0.5) == list(range(len(k)))
            assert len(k) <= 30 and (k[:, 4] >= 0.7).all()


@pytest.mark.parametrize('precision', ['f16', 'f16x2'])
def test_baseline_config3_tiny_yolo_416_end_to_end(precision):
    """BASELINE configs[2]: tiny_yolo 416x416, two scales (13x13, 26x26); per-GPU shard of the batch of 64 = 8 images (yolonet.py:107-158).
    The reference's hard-coded Reshape((7,10,..)) cannot run this shape; here out_hw is derived.  f16x2 (the conforming mode) is
    held to the north-star bar on all 8 images: (class, box index) sets exact, scores / corners within 1e-3 of the fp32 oracle; f16 to its
    storage budget against the fp16-emulating oracle."""
    import torch
    from k210_yolo_framework_amd import engine
    spec = ns.tiny_yolo((416, 416, 3), 3, 20)
    assert spec.out_hw() == [(13, 13), (26, 26)]
    w = spec.init_weights(seed=1)
    B = 8
    frames = np.random.default_rng(3).integers(0, 256, (B, 416, 416, 3), dtype=np.uint8)
    if precision == 'f16x2':
        outs, dets = _gpu(spec, w, frames, 0.7, 0.5, precision='f16x2', schedule='throughput')
        assert [o.shape for o in outs] == [(B, 13, 13, 75), (B, 26, 26, 75)]
        ref32 = oracle.net_forward(spec.compile_plan(w), oracle.normalise_u8(frames), emulate_f16=False, out_ids=spec.outputs)
        for g, r in zip(outs, ref32):
            assert np.abs(g - r).max() <= 1e-4 * np.abs(r).max()
        rd = dr.decode_batch([r.reshape(B, r.shape[1], r.shape[2], 3, 25) for r in ref32], VOC_ANCHORS, (416, 416), (416, 416), 0.7, 0.5)
        n = _assert_north_star(dets, [x[0] for x in rd], 'configs[2] f16x2')
        assert n > 100, n
        _gpu.last_dets_cls = [d[:, 5].astype(int) for d in dets]
        assert _assert_exact_indices(_gpu.last_index, rd, 'configs[2] f16x2') == n
        return
    plan = engine.Plan(spec, w, max_batch=B, precision='f16')
    plan.run_u8(torch.from_numpy(frames).cuda())
    cfg = engine.make_decode_cfg(VOC_ANCHORS, 20, (416, 416), spec.out_hw())
    dets, counts = engine.decode_py(cfg, plan.outputs(), B, None, 0.7, 0.5)
    torch.cuda.synchronize()
    outs = [o[:B].cpu().numpy() for o in plan.outputs()]
    assert [o.shape for o in outs] == [(B, 13, 13, 75), (B, 26, 26, 75)]
    ref = oracle.net_forward(spec.compile_plan(w), oracle.normalise_u8(frames[:2]), emulate_f16=True, out_ids=spec.outputs)
    for g, r in zip(outs, ref):
        assert np.abs(g[:2] - r).max() <= 1.5e-2 * max(1.0, np.abs(r).max())
    rd = dr.decode_batch([o.reshape(B, o.shape[1], o.shape[2], 3, 25) for o in outs], VOC_ANCHORS, (416, 416), (416, 416), 0.7, 0.5)
    d, c = dets.cpu().numpy(), counts.cpu().numpy()
    for b in range(B):
        assert c[b] == len(rd[b][0])
        assert np.array_equal(d[b, :c[b], 5], rd[b][0][:, 5])
        np.testing.assert_allclose(d[b, :c[b], :5], rd[b][0][:, :5], rtol=1e-5, atol=2e-3)
    plan.close()


@pytest.mark.parametrize('precision', ['f16x2', 'f16'])
def test_pipeline_three_batches_in_flight_equal_one_at_a_time(precision):
    """engine.Pipeline: different batches interleaved on 3 streams give exactly the detections of the same batches run alone."""
    import torch
    from k210_yolo_framework_amd import engine, netspec as ns
    from k210_yolo_framework_amd.helper import VOC_ANCHORS
    spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
    w = spec.init_weights(seed=1)
    B, N = 16, 9
    g = torch.Generator(device='cuda').manual_seed(5)
    frames = [torch.randint(0, 256, (B, 224, 320, 3), dtype=torch.uint8, device='cuda', generator=g) for _ in range(N)]
    plan = engine.Plan(spec, w, max_batch=B, precision=precision, schedule='throughput')      # what a Pipeline of depth 3 builds
    cfg = engine.make_decode_cfg(VOC_ANCHORS, 20, spec.in_hw, spec.out_hw())
    ref = []
    for f in frames:
        plan.run_u8(f)
        d, c = engine.decode_py(cfg, plan.outputs(), B, None, 0.7, 0.5)
        torch.cuda.synchronize()
        ref.append((d.cpu().numpy().copy(), c.cpu().numpy().copy()))
    pipe = engine.Pipeline(spec, w, VOC_ANCHORS, max_batch=B, depth=3, precision=precision)
    for rounds in range(3):                                    # several passes: slots are reused
        got = [pipe.submit(f) for f in frames[:3]] if rounds == 0 else None
        res = []
        for k in range(0, N, 3):
            batch = [pipe.submit(f) for f in frames[k:k + 3]]
            pipe.wait()
            res += [(d.cpu().numpy(), c.cpu().numpy()) for d, c, _ in batch]
        for (d, c), (rd, rc) in zip(res, ref):
            assert np.array_equal(c, rc)
            for b in range(B):
                assert np.array_equal(d[b, :c[b]], rd[b, :rc[b]])
    pipe.close()
    plan.close()
