"""Golden vectors for Helper._read_img: small images in every PIL mode the reference's loader can meet, read by the REAL
skimage.io.imread + gray2rgb + [..., :3] (tools/utils.py:352-355).

    /opt/conda/bin/python3.9 tests/golden/make_imread_golden.py        (needs scikit-image + Pillow; only lossless formats: JPEG
                                                                         decoders differ between libjpeg builds)
Writes tests/golden/imread/*.{png,bmp} and tests/golden/imread_golden.npz (expected arrays keyed by file name)."""
import os
import numpy as np
import skimage.color
import skimage.io
from PIL import Image

here = os.path.dirname(os.path.abspath(__file__))
out = os.path.join(here, 'imread')
os.makedirs(out, exist_ok=True)
rng = np.random.default_rng(7)
base = rng.integers(0, 256, (12, 16, 4), dtype=np.uint8)
rgb = Image.fromarray(base[..., :3], 'RGB')
rgb.save(os.path.join(out, 'rgb.png'))
rgb.save(os.path.join(out, 'rgb.bmp'))
Image.fromarray(base, 'RGBA').save(os.path.join(out, 'rgba.png'))
Image.fromarray(base[..., 0], 'L').save(os.path.join(out, 'gray.png'))
Image.fromarray(base[..., :2].copy(), 'LA').save(os.path.join(out, 'la.png'))
rgb.convert('P', palette=Image.ADAPTIVE, colors=32).save(os.path.join(out, 'pal.png'))
rgb.convert('P', palette=Image.ADAPTIVE, colors=16).save(os.path.join(out, 'pal_t.png'), transparency=3)
Image.fromarray(base[..., 0].astype(np.uint16) * 257, 'I;16').save(os.path.join(out, 'gray16.png'))
Image.fromarray(base[..., 0] > 127).save(os.path.join(out, 'bilevel.png'))
exp = {}
for name in sorted(os.listdir(out)):
    img = skimage.io.imread(os.path.join(out, name))
    if len(img.shape) != 3:
        img = skimage.color.gray2rgb(img)
    exp[name] = img[..., :3]
np.savez_compressed(os.path.join(here, 'imread_golden.npz'), **exp)
print({k: (v.shape, str(v.dtype)) for k, v in exp.items()})
