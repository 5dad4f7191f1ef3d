"""Host-side Helper mirror (k210_yolo_framework_amd/helper.py) vs hand-derived known answers of the reference's
numpy members (tools/utils.py:54-82,140-307,378-385,492-521).  Pure host logic, no GPU, no oracle."""
import numpy as np

from k210_yolo_framework_amd.helper import Helper, VOC_ANCHORS, letterbox_bilinear, tf_iou, tf_xywh_to_all, tf_xywh_to_grid


def _h():
    return Helper(None, 20, VOC_ANCHORS, [[224, 320]], [[7, 10], [14, 20]])


def test_constructor_tables():
    h = _h()
    assert h.anchor_number == 3 and h.output_number == 2
    assert np.allclose(h.grid_wh, [[1 / 10, 1 / 7], [1 / 20, 1 / 14]])            # (w,h) order, utils.py:70
    assert h.xy_offset[0].shape == (7, 10, 1, 2) and h.xy_offset[1].shape == (14, 20, 1, 2)
    assert h.xy_offset[1][3, 5, 0].tolist() == [5, 3]                             # [...,0]=col (x), [...,1]=row (y)
    assert np.allclose(h.wh_scale[0], VOC_ANCHORS[0] * [1 / 10, 1 / 7])
    assert h.output_shapes == [[None, 7, 10, 3, 25], [None, 14, 20, 3, 25]]


def test_anchor_file_matches_main_c_constants(tmp_path):
    a = np.load('data/voc_anchor.npy')
    assert a.shape == (2, 3, 2) and a.dtype == np.float64
    assert np.allclose(a[0].ravel(), [0.76120044, 0.57155991, 0.6923348, 0.88535553, 0.47163042, 0.34163313])  # main.c:46-48
    assert np.allclose(a[1].ravel(), [0.33340788, 0.70065861, 0.18124964, 0.38986752, 0.08497349, 0.1527057])  # main.c:50-52
    h = Helper(None, 20, 'data/voc_anchor.npy', [[224, 320]], [[7, 10], [14, 20]])
    assert np.array_equal(h.anchors, a)


def test_box_to_label_and_back():
    h = _h()
    boxes = np.array([[11, 0.52, 0.48, 0.70, 0.60], [6, 0.20, 0.30, 0.10, 0.16]], np.float64)
    labels = h.box_to_label(boxes)
    assert [l.shape for l in labels] == [(7, 10, 3, 25), (14, 20, 3, 25)]
    # box 0: best anchor is layer 0 (big anchors); cell = floor(xy * (w,h)) = (5, 3)
    l, n = h._get_anchor_index(boxes[0, 3:5])
    assert l == 0 and labels[0][3, 5, n, 4] == 1 and labels[0][3, 5, n, 5 + 11] == 1
    assert np.allclose(labels[0][3, 5, n, 0:4], boxes[0, 1:5])
    # box 1: small -> layer 1, cell (4, 4)
    l, n = h._get_anchor_index(boxes[1, 3:5])
    assert l == 1 and labels[1][4, 4, n, 4] == 1
    back = h.label_to_box(labels)
    assert sorted(back[:, 0].tolist()) == [6, 11]
    assert sum(int(l[..., 4].sum()) for l in labels) == 2


def test_fake_iou_known_values():
    assert np.isclose(Helper._fake_iou(np.array([0.5, 0.5]), np.array([0.5, 0.5])), 1.0)
    assert np.isclose(Helper._fake_iou(np.array([0.2, 0.4]), np.array([0.4, 0.2])), 0.04 / (0.08 + 0.08 - 0.04))


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


def test_process_img_normalises_by_own_max():
    h = _h()
    img = np.full((224, 320, 3), 100, np.uint8)
    img[0, 0, 0] = 200
    out, _ = h._process_img(img, None, is_training=False, is_resize=True)
    assert out.dtype == np.float64 and out.max() == 1.0 and np.isclose(out[5, 5, 0], 0.5)   # utils.py:405


def test_center_corner_roundtrip():
    h = _h()
    b = np.array([[0.5, 0.5, 0.2, 0.4], [0.1, 0.9, 0.05, 0.1]])
    c = h.center_to_corner(b)
    assert np.allclose(c[0], [(0.5 - 0.1) * 320, (0.5 - 0.2) * 224, (0.5 + 0.1) * 320, (0.5 + 0.2) * 224])
    assert np.allclose(h.corner_to_center(c), b)


def test_free_functions_shapes_and_values():
    h = _h()
    z = np.zeros((7, 10, 3, 2))
    xy, wh = tf_xywh_to_all(z, z, 0, h)
    assert np.allclose(xy[2, 3, 0], [(3 + .5) / 10, (2 + .5) / 7]) and np.allclose(wh[0, 0], VOC_ANCHORS[0])
    gxy, gwh = tf_xywh_to_grid(xy, wh, 0, h)
    assert np.allclose(gxy, 0.5) and np.allclose(gwh, 0.0)
    iou = tf_iou(xy, wh, np.array([[0.35, 0.357142857]]), np.array([VOC_ANCHORS[0][0]]))
    assert iou.shape == (7, 10, 3, 1) and np.isclose(iou[2, 3, 0, 0], 1.0)
