"""A small detector made of the layer kinds and NAMES the reference's Keras models use (shared by tests/golden/make_h5_golden.py,
which runs under an interpreter with h5py, and tests/test_h5lite.py)."""
from k210_yolo_framework_amd import netspec as ns


def mini_spec(class_num=20):
    """conv1 -> 2 x (depthwise + pointwise) -> the y1/y2 head pattern of yolonet.py:27-38, 8..32 channels."""
    s = ns.NetSpec('mini', (32, 32), anchor_num=3, class_num=class_num)
    x = s._new_tensor(32, 32, 3)
    x = s.conv(x, 8, 3, 2, ns.K210_S2_PAD, act=ns.LEAKY03, name='conv1')
    x = s.dwconv(x, 1, ns.SAME3, act=ns.RELU, name='conv_dw_1')
    x1 = s.conv(x, 16, 1, act=ns.LEAKY03, name='conv_pw_1')
    x = s.dwconv(x1, 2, ns.K210_S2_PAD, act=ns.RELU, name='conv_dw_2')
    x = s.conv(x, 32, 1, act=ns.LEAKY03, name='conv_pw_2')
    ns._head(s, x1, x, 24, 16, 8, 3 * (class_num + 5), [0])
    return s
