"""`make train` (reference keras_train.py:27-115) on the HIP training step.

Same CLI and the same `main(...)` parameter list as the reference script.  What `train_model.fit` did inside TensorFlow
is `train.Trainer.step` here; the tf.data pipeline (tools/utils.py:417-450 `_create_dataset`: shuffle, read, letterbox,
`box_to_label`, batch) is `pipeline.InputPipeline` (row N3: per-rank shards, thread-pool decode, GPU letterbox / normalise, two batches of prefetch); the
plain-python `batches()` generator below is its host-only twin (tests compare the two bit for bit).  Validation runs the fp16 inference
engine on the exported weights (BatchNorm with moving statistics, like Keras' test phase).

Checkpoints: `log/<time>/yolo_model.h5` in Keras' WEIGHTS-ONLY HDF5 layout (`model.save_weights` format: readable by the reference's
`load_weights`, keras_inference.py:80; it is not a `save_model` file - no `model_config` - so `keras_freeze.py`'s `load_model` cannot
open it; keras_io / h5lite, no h5py needed) plus the same arrays as `yolo_model.npz`; `--pre_ckpt` takes either.  Differences, reported at run time: imgaug augmentation and tfmot pruning are out of
scope and raise instead of silently doing nothing."""
from __future__ import annotations

import argparse
import os
import sys
import time
from datetime import datetime
from pathlib import Path

import numpy as np

from . import engine, netspec
from .helper import ERROR, INFO, Helper, write_arguments_to_file


def synthetic_list(n: int, in_hw, class_num: int, seed: int):
    """In-memory stand-in for data/<set>_img_ann.npy: [(uint8 image, boxes[cls,x,y,w,h])]: coloured rectangles on noise."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        img = rng.integers(0, 64, (in_hw[0], in_hw[1], 3), dtype=np.uint8)
        k = int(rng.integers(1, 4))
        boxes = np.zeros((k, 5))
        for j in range(k):
            c = int(rng.integers(0, class_num))
            w, h = rng.uniform(0.15, 0.5, 2)
            x, y = rng.uniform(w / 2, 1 - w / 2), rng.uniform(h / 2, 1 - h / 2)
            boxes[j] = [c, x, y, w, h]
            x0, x1 = int((x - w / 2) * in_hw[1]), int((x + w / 2) * in_hw[1])
            y0, y1 = int((y - h / 2) * in_hw[0]), int((y + h / 2) * in_hw[0])
            img[y0:y1, x0:x1] = (np.array([37, 91, 151]) * (c + 1)) % 200 + 55
        out.append((img, boxes))
    return out


def batches(h: Helper, items, batch_size: int, rng, shuffle: bool):
    """tools/utils.py:417-450: (normalised image [B,H,W,3] float32, labels per layer [B,h,w,A,5+C] float32)."""
    order = rng.permutation(len(items)) if shuffle else np.arange(len(items))
    for s in range(0, len(order) - batch_size + 1, batch_size):             # drop_remainder=True (utils.py:447)
        xs, ys = [], [[] for _ in range(len(h.anchors))]
        for i in order[s:s + batch_size]:
            img, boxes = items[i]
            if isinstance(img, (str, os.PathLike)):
                img = h._read_img(str(img))
            boxes = np.array(boxes, np.float64, copy=True)
            img, boxes = h._process_img(img, boxes, is_training=False, is_resize=True)
            xs.append(img.astype(np.float32))
            for l, lab in enumerate(h.box_to_label(boxes)):
                ys[l].append(lab)
        yield np.stack(xs), [np.stack(y).astype(np.float32) for y in ys]


def main(args, train_set, class_num, pre_ckpt, model_def, depth_multiplier, is_augmenter, image_size, output_size, batch_size,
         rand_seed, max_nrof_epochs, init_learning_rate, learning_rate_decay_factor, obj_weight, noobj_weight, wh_weight,
         obj_thresh, iou_thresh, vaildation_split, log_dir, is_prune, initial_sparsity=0.5, final_sparsity=0.9, end_epoch=5,
         frequency=100, synthetic=0, max_steps=0):
    import torch
    from .train import Trainer
    if is_prune == 'True':
        raise engine.YkError('tfmot magnitude pruning (keras_train.py:60-71) is out of scope of this build')
    if is_augmenter == 'True':
        raise engine.YkError('imgaug augmentation (tools/utils.py:357-376) is out of scope; run with IAA=False')
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    engine.require_gpu()
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device(f'cuda:{local}'))
    log_dir = Path(log_dir) / datetime.strftime(datetime.now(), '%Y%m%d-%H%M%S')
    if rank == 0:
        log_dir.mkdir(parents=True, exist_ok=True)
        write_arguments_to_file(args, str(log_dir / 'args.txt'))                # keras_train.py:41
    in_hw, out_hw = np.reshape(np.array(image_size), (-1, 2)), np.reshape(np.array(output_size), (-1, 2))
    anchors = f'data/{train_set}_anchor.npy'
    if synthetic:
        h = Helper(None, class_num, anchors, in_hw, out_hw, vaildation_split)
        items = synthetic_list(synthetic, in_hw[0], class_num, rand_seed)
        nval = int(len(items) * vaildation_split)
        h.test_list, h.train_list = items[:nval], items[nval:]
    else:
        ann = Path(f'data/{train_set}_img_ann.npy')
        if not ann.exists():
            raise engine.YkError(f'{ann} not found (make_voc_list.py output); pass --synthetic N for generated data')
        h = Helper(str(ann), class_num, anchors, in_hw, out_hw, vaildation_split)
        h.train_list = [(a[0], a[1]) for a in h.train_list]
        h.test_list = [(a[0], a[1]) for a in h.test_list]
    if batch_size % world:
        raise engine.YkError(f'batch_size {batch_size} must divide by the {world} ranks')
    per_rank = batch_size // world
    h.batch_size = batch_size
    spec = netspec.NETWORKS[model_def]([image_size[0], image_size[1], 3], len(h.anchors[0]), class_num, alpha=depth_multiplier)
    assert [tuple(x) for x in spec.out_hw()] == [tuple(x) for x in out_hw], (spec.out_hw(), out_hw)
    weights = spec.init_keras_default(rand_seed)
    if pre_ckpt not in (None, 'None', ''):                                       # keras_train.py:52-57
        if 'h5' in str(pre_ckpt):
            from . import keras_io
            weights, _ = keras_io.load_keras_weights(spec, str(pre_ckpt), base=weights, strict=True)
            print(INFO, f' Load CKPT {str(pre_ckpt)}')
        elif str(pre_ckpt).endswith('.npz'):
            weights.update({k: v for k, v in np.load(pre_ckpt).items()})
            print(INFO, f' Load CKPT {str(pre_ckpt)}')
        else:
            print(ERROR, ' Pre CKPT path is unvalid')
    tr = Trainer(spec, weights, h.anchors, per_rank, obj_thresh=obj_thresh, iou_thresh=iou_thresh, obj_weight=obj_weight,
                 noobj_weight=noobj_weight, wh_weight=wh_weight, lr=init_learning_rate, decay=learning_rate_decay_factor, device=local,
                 world_size=world)
    from .pipeline import InputPipeline
    steps = 0
    for epoch in range(max_nrof_epochs):
        t0, seen, run = time.time(), 0, 0.0
        # tools/utils.py:417-450: each rank decodes only its rows of the global batch, on a thread pool, two batches ahead;
        # letterbox + normalise on the GPU (pipeline.py)
        pipe = InputPipeline(h, h.train_list, batch_size, rank, world, seed=rand_seed, epoch=epoch, shuffle=True, device=local)
        try:                                                                    # an exception in the step must not leave the producer running
            for x, ys in pipe:
                out = tr.step(x, ys)
                seen, run, steps = seen + 1, run + out['loss'], steps + 1
                if rank == 0 and (seen % 10 == 0 or seen == 1):
                    pr = tr.precision_recall()
                    print(f'epoch {epoch + 1} step {seen}: loss {out["loss"]:.4f} ' +
                          ' '.join(f'l{i + 1}_p {p:.3f} l{i + 1}_r {r:.3f}' for i, (p, r) in enumerate(pr)), flush=True)
                if max_steps and steps >= max_steps:
                    break
            pipe_rate = pipe.producer_images_per_sec()
        finally:
            pipe.close()
        val = validate(tr, h, spec, per_rank, rank) if rank == 0 and len(h.test_list) >= per_rank else None
        if rank == 0:
            print(f'epoch {epoch + 1}: {seen} steps, mean loss {run / max(seen, 1):.4f}, ' +
                  (f'val_loss {val:.4f}, ' if val is not None else '') + f'{time.time() - t0:.1f}s, input pipeline {pipe_rate:.0f} images/s/rank',
                  flush=True)
        for c in tr.counts:
            c.zero_()                                                           # Keras resets metrics every epoch
        if max_steps and steps >= max_steps:
            break
    if rank == 0:
        from . import keras_io
        ckpt = log_dir / 'yolo_model.h5'                                        # keras_train.py:105-109
        final = tr.export_weights()
        keras_io.save_keras_model(spec, final, str(ckpt))                      # save_model layout: /model_weights + model_config
        np.savez(log_dir / 'yolo_model.npz', **final)
        print()
        print(INFO, f' Save Model as {str(ckpt)}')
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return tr


def validate(tr, h: Helper, spec, batch: int, rank: int) -> float:
    """validation_data pass (keras_train.py:96-98): inference-mode forward on the engine (the f16x2 mode, the boundary default) + the same loss."""
    import torch
    tot, n = 0.0, 0
    e = 5 + spec.class_num
    with engine.Plan(spec, tr.export_weights(), max_batch=batch, device=tr.dev.index or 0) as plan:   # freed on exit, every epoch
        for x, ys in batches(h, h.test_list, batch, np.random.default_rng(0), shuffle=False):
            plan.run_f32(torch.from_numpy(x).cuda())
            for li, (o, yt) in enumerate(zip(plan.outputs(), ys)):
                yp = o[:batch].reshape(batch, *spec.tensors[spec.outputs[li]][:2], spec.anchor_num, e).contiguous()
                loss6, _, _ = engine.yolo_loss(torch.from_numpy(yt).cuda(), yp, tr.anchors[li], batch_size=batch, want_grad=False, **tr.hyper)
                tot += float(loss6[0])
            n += 1
    return tot / max(n, 1)


def cli(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--train_set', type=str, default='voc')
    p.add_argument('--class_num', type=int, default=20)
    p.add_argument('--pre_ckpt', type=str, default='None')
    p.add_argument('--model_def', type=str, default='yolo_mobilev2')
    p.add_argument('--depth_multiplier', type=float, choices=[0.5, 0.75, 1.0], default=1.0)
    p.add_argument('--augmenter', type=str, choices=['True', 'False'], default='False')
    p.add_argument('--image_size', type=int, default=(224, 320), nargs='+')
    p.add_argument('--output_size', type=int, default=(7, 10, 14, 20), nargs='+')
    p.add_argument('--batch_size', type=int, default=16)
    p.add_argument('--rand_seed', type=int, default=6)
    p.add_argument('--max_nrof_epochs', type=int, default=10)
    p.add_argument('--init_learning_rate', type=float, default=0.001)
    p.add_argument('--learning_rate_decay_factor', type=float, default=0)
    p.add_argument('--obj_weight', type=float, default=5.0)
    p.add_argument('--noobj_weight', type=float, default=0.5)
    p.add_argument('--wh_weight', type=float, default=0.5)
    p.add_argument('--obj_thresh', type=float, default=0.7)
    p.add_argument('--iou_thresh', type=float, default=0.3)
    p.add_argument('--vaildation_split', type=float, default=0.1)
    p.add_argument('--log_dir', type=str, default='log')
    p.add_argument('--is_prune', type=str, choices=['True', 'False'], default='False')
    p.add_argument('--prune_initial_sparsity', type=float, default=0.5)
    p.add_argument('--prune_final_sparsity', type=float, default=0.9)
    p.add_argument('--prune_end_epoch', type=int, default=5)
    p.add_argument('--prune_frequency', type=int, default=100)
    p.add_argument('--synthetic', type=int, default=0, help='train on N generated images instead of data/<set>_img_ann.npy')
    p.add_argument('--max_steps', type=int, default=0)
    a = p.parse_args(sys.argv[1:] if argv is None else argv)
    return main(a, a.train_set, a.class_num, a.pre_ckpt, a.model_def, a.depth_multiplier, a.augmenter, a.image_size, a.output_size,
                a.batch_size, a.rand_seed, a.max_nrof_epochs, a.init_learning_rate, a.learning_rate_decay_factor, a.obj_weight,
                a.noobj_weight, a.wh_weight, a.obj_thresh, a.iou_thresh, a.vaildation_split, a.log_dir, a.is_prune,
                a.prune_initial_sparsity, a.prune_final_sparsity, a.prune_end_epoch, a.prune_frequency, a.synthetic, a.max_steps)


if __name__ == '__main__':
    cli()
