"""GPU parity: yk_letterbox_u8 vs oracle/preprocess_ref.py (bit-exact: integer output, same float64 operation order),
plus the host mirror Helper._process_img on the same frames."""
import numpy as np
import pytest

from oracle import preprocess_ref as pr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('src_hw', [(240, 320), (224, 320), (374, 499), (100, 60), (480, 640)])
def test_letterbox_vs_oracle_bit_exact(src_hw):
    import torch
    from k210_yolo_framework_amd import engine
    from k210_yolo_framework_amd.helper import Helper, VOC_ANCHORS, letterbox_bilinear
    rng = np.random.default_rng(src_hw[0])
    frames = rng.integers(0, 256, (3, *src_hw, 3), dtype=np.uint8)
    out = engine.letterbox_u8(torch.from_numpy(frames).cuda(), (224, 320))
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    h = Helper(None, 20, VOC_ANCHORS, [[224, 320]], [[7, 10], [14, 20]])
    for b in range(3):
        ref = pr.letterbox(frames[b], (224, 320))
        assert np.array_equal(out[b], ref)
        s, t = h.letterbox_params(src_hw)
        assert np.array_equal(letterbox_bilinear(frames[b], (224, 320), float(s[0]), t), ref)   # host mirror agrees too
    if src_hw == (224, 320):
        assert np.array_equal(out, frames)                       # dog.jpg case: identity
    if src_hw == (240, 320):
        assert (out[:, :, :10] == 0).all() and (out[:, :, 309:] == 0).all()   # 299-px content, x offset 10
