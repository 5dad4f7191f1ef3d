"""GPU parity: yk_letterbox_u8 vs the REAL scikit-image (tests/golden/letterbox_golden.npz, see test_oracle_pre.py) and vs
oracle/preprocess_ref.py on further sizes — bit-exact (integer output, the same float64 operations in the same order)."""
import numpy as np
import pytest

from oracle import preprocess_ref as pr

pytestmark = pytest.mark.gpu


def test_letterbox_bit_exact_vs_skimage_golden(golden_dir):
    import torch
    from k210_yolo_framework_amd import engine
    g = np.load(golden_dir / 'letterbox_golden.npz')
    i = 0
    while f'img{i}' in g.files:
        img, ref, par = g[f'img{i}'], g[f'out{i}'], g[f'par{i}']
        out = engine.letterbox_u8(torch.from_numpy(np.ascontiguousarray(img[None])).cuda(), (int(par[3]), int(par[4])))
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out[0].cpu().numpy(), ref, err_msg=f'case {i} {img.shape}')
        i += 1
    assert i >= 6


@pytest.mark.parametrize('src_hw', [(240, 320), (224, 320), (480, 640), (77, 200)])
def test_letterbox_batched_vs_oracle_bit_exact(src_hw):
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
        s, t = h.letterbox_params(src_hw)
        ref = letterbox_bilinear(frames[b], (224, 320), float(s[0]), t)             # host mirror (itself pinned to skimage)
        assert np.array_equal(out[b], ref)
    assert np.array_equal(out[0], pr.letterbox(frames[0], (224, 320)))              # oracle, one image (slow loops)
    if src_hw == (224, 320):
        assert np.array_equal(out, frames)                       # dog.jpg case: identity
    if src_hw == (240, 320):
        assert (out[:, :, :10] == 0).all() and (out[:, :, 309:] == 0).all()   # 299-px content, x offset 10
