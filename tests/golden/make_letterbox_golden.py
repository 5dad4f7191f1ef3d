"""Golden vectors for the letterbox (Helper._process_img, tools/utils.py:378-399) from the REAL scikit-image.

Run with an interpreter that has scikit-image (here: /opt/conda/bin/python3.9, scikit-image 0.18.3 — the reference pins 0.15,
whose warp(order=1, mode='constant', cval=0) is the same algorithm):
    /opt/conda/bin/python3.9 tests/golden/make_letterbox_golden.py
The calls below are the reference's own statements (scale / translation rule, AffineTransform, warp(... preserve_range=True)
.astype('uint8')) applied to seeded random images; inputs and outputs are stored, nothing of the reference's source is."""
import numpy as np
import skimage
import skimage.transform

out = {'skimage_version': np.array(skimage.__version__)}
CASES = [((240, 320), (224, 320)), ((374, 499), (224, 320)), ((100, 60), (56, 80)), ((120, 160), (56, 80)), ((37, 91), (64, 48)),
         ((56, 80), (56, 80)), ((333, 500), (112, 160))]
for i, ((h, w), dst) in enumerate(CASES):
    in_hw = np.array(dst)
    rng = np.random.default_rng(1000 + i)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    img_wh = np.array([w, h])
    in_wh = in_hw[::-1]
    scale = in_wh / img_wh
    scale[:] = np.min(scale)
    translation = ((in_wh - img_wh * scale) / 2).astype(int)
    aff = skimage.transform.AffineTransform(scale=scale, translation=translation)
    res = skimage.transform.warp(img, aff.inverse, output_shape=in_hw, preserve_range=True).astype('uint8')
    out[f'img{i}'] = img
    out[f'out{i}'] = res
    out[f'par{i}'] = np.array([scale[0], translation[0], translation[1], dst[0], dst[1]])
np.savez_compressed('tests/golden/letterbox_golden.npz', **out)
print('wrote tests/golden/letterbox_golden.npz')
