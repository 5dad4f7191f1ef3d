"""Host-side Helper mirror (k210_yolo_framework_amd/helper.py): every member pinned by hand-derived known answers of the
behaviour tools/utils.py documents (utils.py:54-82,140-307,378-385,408-450,492-521).  Pure host logic, no GPU, no oracle."""
import numpy as np
import pytest

from k210_yolo_framework_amd.helper import (ERROR, Helper, VOC_ANCHORS, letterbox_bilinear, tf_iou, tf_xywh_to_all,
                                            tf_xywh_to_grid)


def _h():
    return Helper(None, 20, VOC_ANCHORS, [[224, 320]], [[7, 10], [14, 20]])


def test_constructor_tables():
    h = _h()
    assert h.anchor_number == 3 and h.output_number == 2
    assert np.allclose(h.grid_wh, [[1 / 10, 1 / 7], [1 / 20, 1 / 14]])            # (w,h) order, utils.py:70
    assert h.xy_offset[0].shape == (7, 10, 1, 2) and h.xy_offset[1].shape == (14, 20, 1, 2)
    assert h.xy_offset[1][3, 5, 0].tolist() == [5, 3]                             # [...,0]=col (x), [...,1]=row (y)
    assert h.xy_offset[0][6, 9, 0].tolist() == [9, 6] and h.xy_offset[0][0, 0, 0].tolist() == [0, 0]
    assert np.allclose(h.wh_scale[0], VOC_ANCHORS[0] * [1 / 10, 1 / 7])
    assert np.allclose(h.wh_scale[1][2], [0.08497349 / 20, 0.1527057 / 14])
    assert h.output_shapes == [[None, 7, 10, 3, 25], [None, 14, 20, 3, 25]]
    assert len(h.colormap) == 80 and h.colormap[0] == (255, 82, 0) and h.colormap[19] == (255, 0, 245) and h.colormap[79] == (11, 200, 200)


def test_three_scale_tables_have_ragged_shapes():
    anc = np.tile(VOC_ANCHORS[:1], (3, 1, 1))
    h = Helper(None, 20, anc, [[416, 416]], [[13, 13], [26, 26], [52, 52]])
    assert [t.shape for t in h.xy_offset] == [(13, 13, 1, 2), (26, 26, 1, 2), (52, 52, 1, 2)]
    assert h.xy_offset[2][51, 7, 0].tolist() == [7, 51]


def test_image_list_split(tmp_path):
    rows = np.empty(10, dtype=object)
    for i in range(10):
        rows[i] = np.array([f'img{i}.jpg', np.zeros((1, 5)), np.array([10, 10])], dtype=object)
    np.save(tmp_path / 'ann.npy', rows, allow_pickle=True)
    h = Helper(str(tmp_path / 'ann.npy'), 20, VOC_ANCHORS, [[224, 320]], [[7, 10], [14, 20]], validation_split=0.3)
    assert h.test_total_data == 3 and h.train_total_data == 7                     # head of the list = validation (utils.py:62-66)
    assert h.test_list[0][0] == 'img0.jpg' and h.train_list[0][0] == 'img3.jpg'


def test_anchor_file_matches_main_c_constants(tmp_path):
    a = np.load('data/voc_anchor.npy')
    assert a.shape == (2, 3, 2) and a.dtype == np.float64
    assert np.allclose(a[0].ravel(), [0.76120044, 0.57155991, 0.6923348, 0.88535553, 0.47163042, 0.34163313])  # main.c:46-48
    assert np.allclose(a[1].ravel(), [0.33340788, 0.70065861, 0.18124964, 0.38986752, 0.08497349, 0.1527057])  # main.c:50-52
    h = Helper(None, 20, 'data/voc_anchor.npy', [[224, 320]], [[7, 10], [14, 20]])
    assert np.array_equal(h.anchors, a)


def test_xy_grid_index_known_cells():
    h = _h()
    assert h._xy_grid_index(np.array([0.52, 0.48]), 0).tolist() == [5, 3]          # floor(.52*10), floor(.48*7)
    assert h._xy_grid_index(np.array([0.52, 0.48]), 1).tolist() == [10, 6]
    assert h._xy_grid_index(np.array([0.999, 0.0]), 1).tolist() == [19, 0]


def test_fake_iou_known_values():
    assert np.isclose(Helper._fake_iou(np.array([0.5, 0.5]), np.array([0.5, 0.5])), 1.0)
    assert np.isclose(Helper._fake_iou(np.array([0.2, 0.4]), np.array([0.4, 0.2])), 0.04 / (0.08 + 0.08 - 0.04))
    t = Helper._fake_iou(np.array([0.3, 0.3]), VOC_ANCHORS)                        # broadcast over [L, A, 2]
    assert t.shape == (2, 3) and np.isclose(t[1, 0], 0.09 / (0.09 + 0.33340788 * 0.70065861 - 0.09))
    assert np.isclose(t[0, 2], 0.09 / (0.47163042 * 0.34163313))                   # box inside the anchor


def test_get_anchor_index_known_answers():
    h = _h()
    assert tuple(int(v) for v in h._get_anchor_index(np.array([0.70, 0.60]))) == (0, 0)
    assert tuple(int(v) for v in h._get_anchor_index(np.array([0.10, 0.16]))) == (1, 2)
    assert tuple(int(v) for v in h._get_anchor_index(np.array([0.18, 0.39]))) == (1, 1)


def test_box_to_label_and_back():
    h = _h()
    boxes = np.array([[11, 0.52, 0.48, 0.70, 0.60], [6, 0.20, 0.30, 0.10, 0.16]], np.float64)
    labels = h.box_to_label(boxes)
    assert [l.shape for l in labels] == [(7, 10, 3, 25), (14, 20, 3, 25)] and labels[0].dtype == np.float32
    # box 0: best anchor is layer 0 (big anchors); cell = floor(xy * (w,h)) = (5, 3)
    assert labels[0][3, 5, 0, 4] == 1 and labels[0][3, 5, 0, 5 + 11] == 1
    assert np.allclose(labels[0][3, 5, 0, 0:4], boxes[0, 1:5])
    # box 1: small -> layer 1, anchor 2, cell (4, 4)
    assert labels[1][4, 4, 2, 4] == 1 and labels[1][4, 4, 2, 5 + 6] == 1
    back = h.label_to_box(labels)
    assert sorted(back[:, 0].tolist()) == [6, 11] and back.shape == (2, 5)
    assert sum(int(l[..., 4].sum()) for l in labels) == 2
    assert sum(float(l.sum()) for l in h.box_to_label(np.zeros((0, 5)))) == 0.0


def test_box_to_label_clips_and_later_box_wins_the_slot():
    h = _h()
    boxes = np.array([[3, 0.52, 0.48, 0.70, 0.60], [7, 0.55, 0.50, 0.72, 0.58]])   # same cell (5,3), same anchor (0,0)
    lab = h.box_to_label(boxes)[0][3, 5, 0]
    assert np.allclose(lab[:4], [0.55, 0.50, 0.72, 0.58])                          # xywh of the LATER box
    assert lab[5 + 3] == 1 and lab[5 + 7] == 1                                    # class bits accumulate, as in the reference loop
    tall = h.box_to_label(np.array([[2, 0.5, 0.5, 0.7, 1.3]]))                     # h > 1 is clipped to 1 (utils.py:226)
    hot = np.concatenate([l[l[..., 4] > 0] for l in tall])
    assert np.allclose(hot[0, :4], [0.5, 0.5, 0.7, 1.0])
    z = h.box_to_label(np.array([[0, 0.5, 0.5, 0.0, 0.3]]))                        # w = 0 -> clipped up to 1e-8
    hot = np.concatenate([l[l[..., 4] > 0] for l in z])
    assert np.isclose(hot[0, 2], 1e-8)


def test_xy_wh_to_all_in_place():
    h = _h()
    labs = [np.zeros((7, 10, 3, 25), np.float32), np.zeros((14, 20, 3, 25), np.float32)]
    labs[0][2, 3, :, 0:2] = 0.5
    h._xy_to_all(labs)
    assert np.allclose(labs[0][2, 3, 0, 0:2], [0.5 * 0.1 + 3, 0.5 / 7 + 2])          # xy*grid_wh + offset, exactly what utils.py:280 computes
    labs[1][5, 6, 1, 2:4] = np.log(2.0)
    h._wh_to_all(labs)
    assert np.allclose(labs[1][5, 6, 1, 2:4], 2.0 * VOC_ANCHORS[1][1]) and np.allclose(labs[1][0, 0, 0, 2:4], VOC_ANCHORS[1][0])


def test_letterbox_params_known_answers():
    h = _h()
    s, t = h.letterbox_params((224, 320))          # dog.jpg: identity
    assert np.allclose(s, 1.0) and t.tolist() == [0, 0]
    s, t = h.letterbox_params((374, 499))          # people.jpg: scale 0.598930, translation (10, 0)
    assert np.allclose(s, 224 / 374) and abs(s[0] - 0.598930) < 1e-6 and t.tolist() == [10, 0]
    s, t = h.letterbox_params((240, 320))          # 320x240 camera frame: scale .9333, x offset 10
    assert np.allclose(s, 224 / 240) and t.tolist() == [10, 0]


def test_letterbox_identity_and_fill():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (224, 320, 3), dtype=np.uint8)
    assert np.array_equal(letterbox_bilinear(img, (224, 320), 1.0, np.array([0, 0])), img)
    big = rng.integers(1, 256, (374, 499, 3), dtype=np.uint8)
    out = letterbox_bilinear(big, (224, 320), 224 / 374, np.array([10, 0]))
    assert out.shape == (224, 320, 3) and out.dtype == np.uint8
    assert (out[:, :9] == 0).all() and (out[:, 311:] == 0).all()      # zero fill outside the 299-px wide content
    assert out[:, 12:305].mean() > 50


def test_process_img_normalises_by_own_max_and_moves_boxes():
    h = _h()
    img = np.full((224, 320, 3), 100, np.uint8)
    img[0, 0, 0] = 200
    out, _ = h._process_img(img, None, is_training=False, is_resize=True)
    assert out.dtype == np.float64 and out.max() == 1.0 and np.isclose(out[5, 5, 0], 0.5)   # utils.py:405
    box = np.array([[1, 0.5, 0.5, 0.2, 0.4]])
    _, moved = h._process_img(np.full((240, 320, 3), 9, np.uint8), box, is_training=False, is_resize=True)
    s = 224 / 240                                                                  # x: (0.5*320*s + 10)/320, w: 0.2*320*s/320
    assert np.allclose(moved[0], [1, (160 * s + 10) / 320, 120 * s / 224, 0.2 * s, 0.4 * 240 * s / 224])
    with pytest.raises(NotImplementedError):
        h._process_img(img, None, is_training=True, is_resize=True)


def test_read_img_modes(tmp_path):
    from PIL import Image
    h = _h()
    rgb = np.random.default_rng(1).integers(0, 256, (6, 8, 3), dtype=np.uint8)
    Image.fromarray(rgb).save(tmp_path / 'a.png')
    assert np.array_equal(h._read_img(str(tmp_path / 'a.png')), rgb)
    Image.fromarray(rgb[..., 0]).save(tmp_path / 'g.png')                          # gray -> three equal channels (gray2rgb)
    g = h._read_img(str(tmp_path / 'g.png'))
    assert g.shape == (6, 8, 3) and np.array_equal(g[..., 1], rgb[..., 0])
    Image.fromarray(np.dstack([rgb, np.full((6, 8), 77, np.uint8)]), 'RGBA').save(tmp_path / 'r.png')   # alpha dropped (img[..., :3])
    assert np.array_equal(h._read_img(str(tmp_path / 'r.png')), rgb)
    Image.fromarray(rgb).convert('P', palette=Image.ADAPTIVE, colors=8).save(tmp_path / 'p.png')        # palette -> RGB, not indices
    p = h._read_img(str(tmp_path / 'p.png'))
    assert p.shape == (6, 8, 3) and p.max() > 8


def test_set_dataset_and_generator():
    h = _h()
    rng = np.random.default_rng(2)
    rows = [(rng.integers(1, 255, (240, 320, 3), dtype=np.uint8), np.array([[i % 20, 0.5, 0.5, 0.3, 0.3]])) for i in range(11)]
    h.test_list, h.train_list = rows[:3], rows[3:]
    h.train_total_data, h.test_total_data = 8, 3
    h.set_dataset(4, 6, is_training=False)
    assert h.batch_size == 4 and h.train_epoch_step == 2 and h.test_epoch_step == 0        # utils.py:447-450 (floor division)
    x, ys = h.get_iter(True)
    assert x.shape == (4, 224, 320, 3) and x.dtype == np.float32 and x.max() == 1.0
    assert [y.shape for y in ys] == [(4, 7, 10, 3, 25), (4, 14, 20, 3, 25)] and sum(float(y[..., 4].sum()) for y in ys) == 4.0
    seen = [h.get_iter(True)[0].shape for _ in range(5)]                            # repeats for ever, batches stay full
    assert all(s == (4, 224, 320, 3) for s in seen)
    img, boxes = next(h.generator(False, True, False, rows[:1]))
    assert img.shape == (224, 320, 3) and boxes.shape == (1, 5) and rows[0][1][0, 3] == 0.3   # the annotation itself is not modified
    # a validation split SMALLER than one batch (3 rows, batches of 4): the reference repeats before it batches (utils.py:438-441), so
    # its batches run across passes; an empty list is an error, not a silent spin
    xv, yv = h.get_iter(False)
    assert xv.shape == (4, 224, 320, 3) and sum(float(y[..., 4].sum()) for y in yv) == 4.0
    assert h.get_iter(False)[0].shape == (4, 224, 320, 3)
    with pytest.raises(ValueError):
        next(h._create_dataset([], 4, 6, False, True))
    assert list(h._create_dataset([], 4, 6, False, True, repeat=False)) == []


def test_center_corner_roundtrip():
    h = _h()
    b = np.array([[0.5, 0.5, 0.2, 0.4], [0.1, 0.9, 0.05, 0.1]])
    c = h.center_to_corner(b)
    assert np.allclose(c[0], [(0.5 - 0.1) * 320, (0.5 - 0.2) * 224, (0.5 + 0.1) * 320, (0.5 + 0.2) * 224])
    assert np.allclose(h.corner_to_center(c), b)
    assert np.allclose(h.center_to_corner(b, to_all_scale=False)[1], [0.075, 0.85, 0.125, 0.95])
    assert np.allclose(h.corner_to_center(np.array([[0.075, 0.85, 0.125, 0.95]]), from_all_scale=False), b[1:])


def test_free_functions_shapes_and_values():
    h = _h()
    z = np.zeros((7, 10, 3, 2))
    xy, wh = tf_xywh_to_all(z, z, 0, h)
    assert np.allclose(xy[2, 3, 0], [(3 + .5) / 10, (2 + .5) / 7]) and np.allclose(wh[0, 0], VOC_ANCHORS[0])
    gxy, gwh = tf_xywh_to_grid(xy, wh, 0, h)
    assert np.allclose(gxy, 0.5) and np.allclose(gwh, 0.0)
    iou = tf_iou(xy, wh, np.array([[0.35, 0.357142857]]), np.array([VOC_ANCHORS[0][0]]))
    assert iou.shape == (7, 10, 3, 1) and np.isclose(iou[2, 3, 0, 0], 1.0)
    two = tf_iou(np.array([[[[0.5, 0.5]]]]), np.array([[[[0.2, 0.2]]]]), np.array([[0.6, 0.5], [0.9, 0.9]]), np.array([[0.2, 0.2], [0.1, 0.1]]))
    assert two.shape == (1, 1, 1, 2) and np.isclose(two[0, 0, 0, 0], 0.02 / (0.04 + 0.04 - 0.02)) and two[0, 0, 0, 1] == 0.0


def test_error_tag_text():
    assert ERROR == '[ ERROR ]'


def test_read_img_equals_real_skimage_imread_for_every_pil_mode():
    """tools/utils.py:352-355 (skimage.io.imread, gray2rgb, drop alpha).  Expected arrays were produced by the REAL scikit-image
    (tests/golden/make_imread_golden.py); lossless files only - JPEG decoders differ between libjpeg builds."""
    import os
    from k210_yolo_framework_amd.helper import Helper
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    exp = np.load(os.path.join(here, 'imread_golden.npz'))
    h = Helper.__new__(Helper)
    assert len(exp.files) == 9
    for name in exp.files:
        got = h._read_img(os.path.join(here, 'imread', name))
        assert got.dtype == exp[name].dtype and got.shape == exp[name].shape, name
        assert np.array_equal(got, exp[name]), name


def test_xy_to_grid_is_the_inverse_of_xy_to_all_on_a_known_cell():
    """utils.py:107-122: image-relative centre -> offset inside its cell.  (0.53, 0.47) on the 14x20 grid lies in column 10, row 6,
    0.6 of a cell from its left edge and 0.58 from its top."""
    h = _h()
    xy = np.zeros((14, 20, 3, 2), np.float32)
    xy[6, 10, :, :] = (0.53, 0.47)
    g = h._xy_to_grid(xy, 1)
    assert g.shape == (14, 20, 3, 2)
    assert np.allclose(g[6, 10, 0], [0.53 * 20 - 10, 0.47 * 14 - 6]) and np.allclose(g[6, 10, 0], [0.6, 0.58], atol=1e-6)
    assert np.allclose(g[0, 0, 0], [0, 0]) and np.allclose(g[3, 5, 1], [-5, -3])           # empty cells: minus their own offset


def test_tf_reshape_box_pairs_every_prediction_with_every_true_box():
    """utils.py:575-614: shapes [B,h,w,A,n,2] on both sides, values repeated, nothing copied."""
    from k210_yolo_framework_amd.helper import tf_reshape_box
    h = _h()
    h.batch_size = 2
    rng = np.random.default_rng(0)
    t_xy, t_wh = rng.uniform(size=(5, 2)), rng.uniform(size=(5, 2))
    p_xy, p_wh = rng.uniform(size=(2, 7, 10, 3, 2)), rng.uniform(size=(2, 7, 10, 3, 2))
    tc, tw, pc, pw = tf_reshape_box(t_xy, t_wh, p_xy, p_wh, 0, h)
    assert tc.shape == tw.shape == pc.shape == pw.shape == (2, 7, 10, 3, 5, 2)
    assert np.array_equal(tc[1, 6, 9, 2], t_xy) and np.array_equal(tw[0, 0, 0, 0], t_wh)
    assert np.array_equal(pc[1, 3, 4, 2, 4], p_xy[1, 3, 4, 2]) and np.array_equal(pw[0, 6, 9, 0, 0], p_wh[0, 6, 9, 0])
    with pytest.raises(ValueError):
        tf_reshape_box(t_xy, t_wh, p_xy[:1], p_wh[:1], 0, h)


def test_write_arguments_to_file(tmp_path):
    """keras_train.py:23-26,41: `key: value` per line, declaration order."""
    import argparse
    from k210_yolo_framework_amd.helper import write_arguments_to_file
    ns_ = argparse.Namespace(train_set='voc', batch_size=16, image_size=[224, 320], pre_ckpt=None)
    write_arguments_to_file(ns_, str(tmp_path / 'args.txt'))
    assert (tmp_path / 'args.txt').read_text() == 'train_set: voc\nbatch_size: 16\nimage_size: [224, 320]\npre_ckpt: None\n'
