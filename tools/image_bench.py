#!/usr/bin/env python
"""Input-pipeline micro-benchmark: the device image transform (uniir_image_preprocess, 500 x 375 RGB -> 3 x 224 x 224 fp32,
CLIP geometry) against the same transform with Pillow on one host core.
    python tools/image_bench.py [--images 512]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uniir_amd import clip_front  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=512)
    args = ap.parse_args()
    rng = np.random.default_rng(0)
    dev = torch.device("cuda")
    host = [rng.integers(0, 256, (375, 500, 3), dtype=np.uint8) for _ in range(args.images)]
    resident = [torch.from_numpy(x).to(dev) for x in host]
    out = torch.empty(args.images, 3, 224, 224, device=dev)
    for src, what in ((resident, "uint8 already in HBM"), (host, "uint8 in pageable host memory (H2D included)")):
        clip_front.preprocess_on_device(src[:8], 224, dev, out=out[:8])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        clip_front.preprocess_on_device(src, 224, dev, out=out)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"device transform, {what}: {args.images / dt:8.0f} images/s ({dt / args.images * 1e6:6.1f} us per image)")
    from PIL import Image
    fn = clip_front._preprocess(224)
    pil = [Image.fromarray(x, "RGB") for x in host[:64]]
    t0 = time.perf_counter()
    for im in pil:
        fn(im)
    dt = time.perf_counter() - t0
    print(f"Pillow + torch on one host core: {64 / dt:8.0f} images/s ({dt / 64 * 1e6:6.1f} us per image)")


if __name__ == "__main__":
    main()
