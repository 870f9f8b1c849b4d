"""G14: golden vectors of the image transform's integer stage, produced with Pillow (third-party; the version is stored in
the fixture) in the build container:  Image.fromarray(x).resize((ow, oh), Image.BICUBIC)  for RGB uint8 images of several
shapes (down- and up-scaling, one unchanged axis, identity, saturated checkerboards that overshoot), with (oh, ow) from
torchvision's Resize(n_px) rule.  tests/test_oracle_image.py holds oracle/oracle.c to these bit for bit.
    python tests/golden/make_golden_image.py"""
import os
import sys

import numpy as np
import PIL
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.c_oracle import resize_geometry  # noqa: E402  (geometry only: integers)

rng = np.random.default_rng(14)
cases = [((60, 80), 32), ((80, 60), 32), ((33, 47), 32), ((20, 30), 32), ((200, 150), 64), ((64, 64), 64), ((90, 64), 64),
         ((188, 250), 112), ((57, 301), 48)]
out = {"pillow_version": np.array(PIL.__version__)}
for idx, ((h, w), n) in enumerate(cases):
    if idx % 3 == 2:       # saturated checkerboard + noise: exercises the clamp after the cubic overshoot
        yy, xx = np.mgrid[0:h, 0:w]
        img = (((yy // 3 + xx // 5) % 2) * 255).astype(np.uint8)[..., None].repeat(3, axis=2)
        img[..., 1] = 255 - img[..., 1]
        img[::7, ::4, 2] = rng.integers(0, 256, img[::7, ::4, 2].shape, dtype=np.uint8)
    else:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    oh, ow, top, left = resize_geometry(h, w, n)
    res = np.asarray(Image.fromarray(img, "RGB").resize((ow, oh), Image.BICUBIC))
    out[f"c{idx}_img"], out[f"c{idx}_res"] = img, res
    out[f"c{idx}_geom"] = np.array([n, oh, ow, top, left])
out["n_cases"] = np.array(len(cases))
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "g14_image.npz"), **out)
print("wrote g14_image.npz", {k: v.shape for k, v in out.items() if k.endswith("_res")})
