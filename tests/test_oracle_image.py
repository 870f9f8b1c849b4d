"""Pins the C restatement of the image transform (oracle/oracle.c: Pillow's 8-bit BICUBIC resample, torchvision's
Resize / CenterCrop geometry, ToTensor + Normalize) to G14 -- arrays produced by Pillow itself
(tests/golden/make_golden_image.py).  Integer work: bit-exact."""
import os

import numpy as np

from oracle import c_oracle

G = os.path.join(os.path.dirname(__file__), "golden")


def test_g14_bicubic_resample_is_bit_exact_with_pillow():
    z = np.load(os.path.join(G, "g14_image.npz"))
    for c in range(int(z["n_cases"])):
        img, want = z[f"c{c}_img"], z[f"c{c}_res"]
        n, oh, ow, top, left = (int(v) for v in z[f"c{c}_geom"])
        assert (oh, ow, top, left) == c_oracle.resize_geometry(img.shape[0], img.shape[1], n)
        got = c_oracle.resize_bicubic(img, oh, ow)
        assert np.array_equal(got, want), (c, img.shape, (oh, ow), int(np.abs(got.astype(int) - want.astype(int)).max()))


def test_resize_geometry_and_normalisation():
    # torchvision rule: long side truncated, crop offsets rounded half to even
    assert c_oracle.resize_geometry(375, 500, 224) == (224, 298, 0, 37)
    assert c_oracle.resize_geometry(500, 375, 224) == (298, 224, 37, 0)
    assert c_oracle.resize_geometry(333, 500, 224) == (224, 336, 0, 56)
    assert c_oracle.resize_geometry(300, 401, 224) == (224, 299, 0, 38)      # 37.5 -> 38 (even)
    assert c_oracle.resize_geometry(224, 224, 224) == (224, 224, 0, 0)
    img = np.arange(6 * 8 * 3, dtype=np.uint8).reshape(6, 8, 3) * 3
    mean, std = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)
    out = c_oracle.clip_preprocess(img, 6, mean, std)       # 6 x 8 -> resize is the identity on the short side, crop 1 column each side
    crop = img[:, 1:7].astype(np.float32) / np.float32(255.0)
    want = (crop.transpose(2, 0, 1) - np.float32(mean)[:, None, None]) / np.float32(std)[:, None, None]
    assert out.dtype == np.float32 and np.array_equal(out, want.astype(np.float32))
