"""The training input pipeline: what tools/utils.py:417-450 builds with tf.data -

    Dataset.from_tensor_slices(list).shuffle().map(py_function(read + letterbox + box_to_label), num_parallel_calls=AUTOTUNE)
           .batch(batch_size, drop_remainder=True).prefetch(AUTOTUNE)

- rebuilt so that it can feed the HIP training step (2.5 k images/s/GPU) instead of starving it:

  * every rank decodes ONLY ITS SHARE of the global batch: the epoch order is one permutation drawn from the shared seed, global batch g
    is order[g*GB:(g+1)*GB], rank r takes rows [r*per, (r+1)*per) of it (round 2 built the whole global batch on every rank and sliced);
  * file decode (`Helper._read_img`, PIL releases the GIL) and label scatter (`Helper.box_to_label`) run on a small thread pool;
  * the letterbox (tools/utils.py:378-399) and `img / np.max(img)` (:405) run on the GPU - `yk_letterbox_u8` (bit-exact against
    scikit-image, tests/golden/letterbox_golden.npz) and `yk_normalise_u8` - on a side stream, frames travel as u8 (a quarter of the
    float bytes) from pinned buffers;
  * a producer thread keeps `prefetch` batches ahead of the consumer (`.prefetch`); the consumer gets device tensors whose producing
    stream it must wait on (`batch.ready` is a recorded event; `__iter__` does the wait for the current stream).

Images of one batch may have different sizes (VOC): they are letterboxed in groups of equal size.  Augmentation (imgaug,
utils.py:357-376) is out of scope as everywhere in this build.
"""
from __future__ import annotations

import os
import queue
import threading
import time
from concurrent.futures import ThreadPoolExecutor
from typing import List, Optional, Sequence

import numpy as np

from . import engine
from .helper import Helper


def epoch_order(n_items: int, seed: int, epoch: int, shuffle: bool) -> np.ndarray:
    """The epoch's sample order, identical on every rank (shared seed)."""
    return np.random.default_rng([seed, epoch]).permutation(n_items) if shuffle else np.arange(n_items)


def rank_rows(order: np.ndarray, global_batch: int, rank: int, world: int) -> List[np.ndarray]:
    """Per step: the dataset rows THIS rank decodes (drop_remainder=True, utils.py:447)."""
    if global_batch % world:
        raise ValueError(f'global batch {global_batch} must divide by {world} ranks')
    per = global_batch // world
    return [order[s + rank * per: s + (rank + 1) * per] for s in range(0, len(order) - global_batch + 1, global_batch)]


def letterbox_boxes(h: Helper, img_hw, boxes: np.ndarray) -> np.ndarray:
    """The box half of Helper._process_img (utils.py:386-389): centre / size fractions of the source image -> of the network tensor."""
    boxes = np.array(boxes, np.float64, copy=True)
    if boxes.size:
        scale, translation = h.letterbox_params(img_hw)
        src_wh, net_wh = np.tile(np.array(img_hw[::-1], float), 2), np.tile(h.in_hw[0][::-1].astype(float), 2)
        moved = boxes[:, 1:5] * src_wh * np.tile(scale, 2)
        moved[:, :2] += translation
        boxes[:, 1:5] = moved / net_wh
    return boxes


def letterbox_boxes_batch(h: Helper, img_hws, boxes_list) -> List[np.ndarray]:
    """`letterbox_boxes` for every sample of a batch with one pass of array arithmetic (same element-wise operations in the same order:
    bit-identical), the letterbox parameters computed once per distinct image size."""
    per = [np.array(b, np.float64, copy=True).reshape(-1, 5) for b in boxes_list]
    counts = [len(b) for b in per]
    if sum(counts) == 0:
        return per
    params = {}
    for hw in img_hws:
        if tuple(hw) not in params:
            scale, translation = h.letterbox_params(hw)
            params[tuple(hw)] = (np.tile(np.array(hw[::-1], float), 2), np.tile(scale, 2), np.asarray(translation, float))
    allb = np.concatenate(per)
    src = np.repeat(np.stack([params[tuple(hw)][0] for hw in img_hws]), counts, axis=0)
    sc = np.repeat(np.stack([params[tuple(hw)][1] for hw in img_hws]), counts, axis=0)
    tr = np.repeat(np.stack([params[tuple(hw)][2] for hw in img_hws]), counts, axis=0)
    moved = allb[:, 1:5] * src * sc
    moved[:, :2] += tr
    allb[:, 1:5] = moved / np.tile(h.in_hw[0][::-1].astype(float), 2)
    return np.split(allb, np.cumsum(counts)[:-1])


class Batch:
    __slots__ = ('x', 'labels', 'ready', 'n')

    def __init__(self, x, labels, ready, n):
        self.x, self.labels, self.ready, self.n = x, labels, ready, n


# Pinned host staging buffers, shared by the epochs' pipelines of a process (allocating pinned memory per batch costs more than the
# batch).  Keyed by CAPACITY BUCKET, not by shape: real VOC lists hold hundreds of image sizes and the count of every size varies from
# batch to batch - a ring per exact (count, h, w) would pin a new buffer for nearly every batch and never give it back.  A request takes
# a reshaped view of the prefix of a power-of-two sized byte buffer; the total is capped and the least recently used ring is released.
_PINNED: "dict" = {}          # (device, what, bucket bytes) -> {'ring': [[pinned uint8 tensor, event of its last H2D copy], ...], 'tick': last use}
_PINNED_CAP_BYTES = 1 << 30   # upper bound of page-locked staging memory per process
_pinned_clock = 0
_pinned_lock = threading.Lock()


def _bucket(nbytes: int) -> int:
    return 1 << max(12, (int(nbytes) - 1).bit_length())


def pinned_bytes() -> int:
    return sum(sl[0].numel() for e in _PINNED.values() for sl in e['ring'])


def _evict(keep_key, need: int) -> None:
    """Release least recently used rings (never `keep_key`) until `need` more bytes fit under the cap."""
    while _PINNED and pinned_bytes() + need > _PINNED_CAP_BYTES:
        victims = [k for k in _PINNED if k != keep_key]
        if not victims:
            return
        k = min(victims, key=lambda kk: _PINNED[kk]['tick'])
        for sl in _PINNED[k]['ring']:
            if sl[1] is not None:
                sl[1].synchronize()
        del _PINNED[k]


class InputPipeline:
    """Iterable over one epoch: yields (x [per_rank,H,W,3] float32 cuda, [labels per layer, float32 cuda])."""

    def __init__(self, h: Helper, items: Sequence, global_batch: int, rank: int = 0, world: int = 1, seed: int = 0, epoch: int = 0,
                 shuffle: bool = True, workers: int = 8, prefetch: int = 2, device: Optional[int] = None):
        import torch
        engine.require_gpu()
        self.h, self.items = h, items
        self.rows = rank_rows(epoch_order(len(items), seed, epoch, shuffle), global_batch, rank, world)
        self.per = global_batch // world
        self.dev = torch.device('cuda', torch.cuda.current_device() if device is None else device)
        self.pool = ThreadPoolExecutor(max_workers=max(1, workers))
        self.q: "queue.Queue" = queue.Queue(maxsize=max(1, prefetch))
        self.stream = torch.cuda.Stream(device=self.dev)
        self.images = 0
        self.seconds = 0.0
        self._thread = None
        self._stop = False

    def __len__(self):
        return len(self.rows)

    # one image on a worker thread (file decode releases the GIL; arrays already in memory are taken as they are)
    def _image(self, i: int):
        img = self.items[int(i)][0]
        if isinstance(img, (str, os.PathLike)):
            img = self.h._read_img(str(img))
        return np.ascontiguousarray(img[..., :3], np.uint8)

    def _slot(self, key, shape, dtype):
        """[view of a pinned host staging buffer shaped `shape`, its slot] from a small ring per capacity bucket."""
        import torch
        global _pinned_clock
        nbytes = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        bk = _bucket(nbytes)
        k = (self.dev.index, key, bk)
        with _pinned_lock:
            _pinned_clock += 1
            e = _PINNED.setdefault(k, {'ring': [], 'tick': 0})
            e['tick'] = _pinned_clock
            ring = e['ring']
            nslots = self.q.maxsize + 2                                     # more than the batches that can be outstanding
            if len(ring) < nslots:
                _evict(k, bk)
                ring.append([torch.empty((bk,), dtype=torch.uint8).pin_memory(), None])
                slot = ring[-1]
            else:
                slot = ring[self._tick % nslots]
        if slot[1] is not None:
            slot[1].synchronize()                                           # its last H2D copy has left the buffer
        return slot[0][:nbytes].view(dtype).view(tuple(shape)), slot

    def _produce(self):
        import torch
        H, W = int(self.h.in_hw[0][0]), int(self.h.in_hw[0][1])
        L = engine.lib()
        self._tick = 0
        try:
            for rows in self.rows:
                if self._stop:
                    break
                t0 = time.perf_counter()
                # decode on the pool only when there is something to decode: for rows that are arrays already the pool's hand-off
                # costs more than it buys (threads of numpy work share the GIL: 1.9 vs 1.4 ms per 16 samples)
                from_files = isinstance(self.items[int(rows[0])][0], (str, os.PathLike))
                imgs = list(self.pool.map(self._image, rows)) if from_files else [self._image(i) for i in rows]
                n = len(imgs)
                self._tick += 1
                # labels of the whole batch in a handful of array operations (utils.py:207-230 on the letterboxed boxes), then ONE copy per
                # layer into the pinned staging buffer (pinned memory is slow to write piecemeal from the CPU)
                labs = self.h.batch_box_to_label(letterbox_boxes_batch(self.h, [im.shape[:2] for im in imgs],
                                                                       [self.items[int(i)][1] for i in rows]))
                lab_slots = [self._slot(('lab', l), labs[l].shape, torch.float32) for l in range(len(labs))]
                for (view, _), lab in zip(lab_slots, labs):
                    view.numpy()[...] = lab
                with torch.cuda.stream(self.stream):
                    frames = torch.empty((n, H, W, 3), dtype=torch.uint8, device=self.dev)
                    by_size = {}
                    for k, img in enumerate(imgs):
                        by_size.setdefault(img.shape[:2], []).append(k)
                    for (sh, sw), idx in by_size.items():                   # equal-sized images are letterboxed in one launch
                        view, slot = self._slot('img', (len(idx), sh, sw, 3), torch.uint8)
                        hv = view.numpy()
                        for j, k in enumerate(idx):
                            hv[j] = imgs[k]
                        src = view.to(self.dev, non_blocking=True)
                        slot[1] = torch.cuda.Event()
                        slot[1].record(self.stream)
                        out = engine.letterbox_u8(src, (H, W), stream=self.stream)
                        if len(by_size) == 1:
                            frames = out
                        else:
                            frames[torch.as_tensor(idx, device=self.dev)] = out
                    x = torch.empty((n, H, W, 3), dtype=torch.float32, device=self.dev)
                    engine._check(L.yk_normalise_u8(engine._ptr(frames), n, engine.C.c_size_t(H * W * 3), engine._ptr(x),
                                                    engine._stream(self.stream)), 'yk_normalise_u8')
                    labels = []
                    for view, slot in lab_slots:
                        labels.append(view.to(self.dev, non_blocking=True))
                        slot[1] = torch.cuda.Event()
                        slot[1].record(self.stream)
                    ready = torch.cuda.Event()
                    ready.record(self.stream)
                self.images += n
                self.seconds += time.perf_counter() - t0
                if not self._put(Batch(x, labels, ready, n)):
                    return
        except BaseException as e:                                          # surface worker errors in the consumer
            self._put(e)
            return
        self._put(None)

    def _put(self, item) -> bool:
        """queue.put that gives up when the consumer has gone away (close() after an early break): never blocks forever."""
        while not self._stop:
            try:
                self.q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def __iter__(self):
        import torch
        self._stop = False
        self._thread = threading.Thread(target=self._produce, daemon=True)
        self._thread.start()
        while True:
            b = self.q.get()
            if b is None:
                break
            if isinstance(b, BaseException):
                raise b
            cur = torch.cuda.current_stream()
            cur.wait_event(b.ready)
            # the tensors were allocated on the producer's side stream and are consumed on this one: tell the caching allocator, or a
            # block could go back to the producer while kernels queued here still read it
            b.x.record_stream(cur)
            for t in b.labels:
                t.record_stream(cur)
            yield b.x, b.labels
        self._thread.join()

    def close(self):
        """Stop the producer and wait for it: drains the queue until the thread has exited (it may be blocked in put), then the pool."""
        self._stop = True
        t = self._thread
        while t is not None and t.is_alive():
            try:
                while True:
                    self.q.get_nowait()
            except queue.Empty:
                pass
            t.join(timeout=0.05)
        try:
            while True:
                self.q.get_nowait()
        except queue.Empty:
            pass
        self._thread = None
        self.pool.shutdown(wait=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def producer_images_per_sec(self) -> float:
        """Rate of the producer alone (decode + labels + H2D + GPU letterbox / normalise), not limited by the consumer."""
        return self.images / self.seconds if self.seconds else 0.0
