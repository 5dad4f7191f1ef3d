"""N3 on the GPU: InputPipeline (thread-pool decode, GPU letterbox + normalise, prefetch, per-rank rows) delivers bit for bit what the
host-only `training.batches` (= tools/utils.py:417-450 restated) builds, for every rank of a 2-rank job, with mixed image sizes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_pipeline_equals_the_host_generator_for_every_rank(tmp_path):
    import torch
    from PIL import Image
    from k210_yolo_framework_amd import pipeline, training
    from k210_yolo_framework_amd.helper import Helper, VOC_ANCHORS
    h = Helper(None, 20, VOC_ANCHORS, [[224, 320]], [[7, 10], [14, 20]])
    rng = np.random.default_rng(5)
    items = []
    for k in range(22):
        hw = [(240, 320), (375, 500), (333, 500), (224, 320)][k % 4]
        img = rng.integers(0, 256, (*hw, 3), dtype=np.uint8)
        n = int(rng.integers(1, 4))
        boxes = np.concatenate([rng.integers(0, 20, (n, 1)).astype(float), rng.uniform(0.2, 0.8, (n, 2)), rng.uniform(0.05, 0.3, (n, 2))], 1)
        if k % 3 == 0:                                                        # some samples come from files, like the VOC list
            p = tmp_path / f'{k}.png'
            Image.fromarray(img).save(p)
            items.append((str(p), boxes))
        else:
            items.append((img, boxes))
    GB, world = 8, 2
    order = pipeline.epoch_order(len(items), seed=3, epoch=1, shuffle=True)
    # host twin on the same order
    class _Fixed:
        def permutation(self, n):
            return order
    want = list(training.batches(h, items, GB, _Fixed(), shuffle=True))
    assert len(want) == len(items) // GB
    for rank in range(world):
        pipe = pipeline.InputPipeline(h, items, GB, rank, world, seed=3, epoch=1, shuffle=True, workers=4, prefetch=2)
        got = [(x.cpu().numpy(), [y.cpu().numpy() for y in ys]) for x, ys in pipe]
        pipe.close()
        assert len(got) == len(want)
        sl = slice(rank * GB // world, (rank + 1) * GB // world)
        for (gx, gys), (wx, wys) in zip(got, want):
            np.testing.assert_array_equal(gx, wx[sl])                          # letterbox bit-exact, normalisation correctly rounded
            for gy, wy in zip(gys, wys):
                np.testing.assert_array_equal(gy, wy[sl])
        assert pipe.producer_images_per_sec() > 0


def test_worker_errors_reach_the_consumer():
    from k210_yolo_framework_amd import pipeline
    from k210_yolo_framework_amd.helper import Helper, VOC_ANCHORS
    h = Helper(None, 20, VOC_ANCHORS, [[224, 320]], [[7, 10], [14, 20]])
    items = [('/nonexistent/file.jpg', np.zeros((1, 5)))] * 4
    pipe = pipeline.InputPipeline(h, items, 4, 0, 1, shuffle=False)
    with pytest.raises(Exception):
        list(pipe)
    pipe.close()


def test_pinned_staging_memory_is_bounded_on_many_image_sizes(monkeypatch):
    """Real VOC lists hold hundreds of image sizes and the count of each size varies per batch: the staging rings are keyed by capacity
    bucket and capped, so page-locked memory stays bounded - and the batches are still the host generator's."""
    from k210_yolo_framework_amd import pipeline, training
    from k210_yolo_framework_amd.helper import Helper, VOC_ANCHORS
    h = Helper(None, 20, VOC_ANCHORS, [[224, 320]], [[7, 10], [14, 20]])
    rng = np.random.default_rng(9)
    items = []
    for k in range(96):
        hw = (int(rng.integers(200, 400)), int(rng.integers(300, 520)))         # (nearly) every image its own size
        img = rng.integers(0, 256, (*hw, 3), dtype=np.uint8)
        boxes = np.concatenate([rng.integers(0, 20, (2, 1)).astype(float), rng.uniform(0.2, 0.8, (2, 2)), rng.uniform(0.05, 0.3, (2, 2))], 1)
        items.append((img, boxes))
    pipeline._PINNED.clear()
    monkeypatch.setattr(pipeline, '_PINNED_CAP_BYTES', 24 << 20)
    order = pipeline.epoch_order(len(items), seed=1, epoch=0, shuffle=True)

    class _Fixed:
        def permutation(self, n):
            return order
    want = list(training.batches(h, items, 8, _Fixed(), shuffle=True))
    pipe = pipeline.InputPipeline(h, items, 8, 0, 1, seed=1, epoch=0, shuffle=True)
    peak = 0
    for (x, ys), (wx, wys) in zip(pipe, want):
        np.testing.assert_array_equal(x.cpu().numpy(), wx)
        for gy, wy in zip(ys, wys):
            np.testing.assert_array_equal(gy.cpu().numpy(), wy)
        peak = max(peak, pipeline.pinned_bytes())
    pipe.close()
    assert 0 < peak <= (24 << 20) + (4 << 20), peak                             # one ring may overshoot by its own newest buffer
    assert all(pipeline._bucket(n) >= n and pipeline._bucket(n) < 2 * max(n, 4096) for n in (1, 4095, 4096, 4097, 10_000_000))


def test_close_after_an_early_break_stops_the_producer():
    import threading
    import time
    from k210_yolo_framework_amd import pipeline
    from k210_yolo_framework_amd.helper import Helper, VOC_ANCHORS
    h = Helper(None, 20, VOC_ANCHORS, [[224, 320]], [[7, 10], [14, 20]])
    rng = np.random.default_rng(2)
    items = [(rng.integers(0, 256, (240, 320, 3), dtype=np.uint8), np.array([[1, .5, .5, .2, .2]])) for _ in range(64)]
    before = threading.active_count()
    pipe = pipeline.InputPipeline(h, items, 4, 0, 1, prefetch=1)
    for k, _ in enumerate(pipe):
        if k == 1:
            break                                                               # the producer is (about to be) blocked in put
    t0 = time.time()
    pipe.close()
    assert time.time() - t0 < 5.0
    assert pipe._thread is None and threading.active_count() <= before + 1      # (pool threads are joined by shutdown)
