#!/usr/bin/env python
"""Secondary benchmark: CLIP_FF ViT-L/14 train step (towers without pooling -> 2-layer T5 fusion over 334 tokens ->
mean pooling -> InfoNCE), synthetic data, one MI355X.   python tools/bench_clipff.py --pairs 256 --steps 4"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "uniir_amd", "src"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=256)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    a = ap.parse_args()
    from bench import synth_batch
    from models.uniir_clip.clip_featurefusion.clip_ff import CLIPFeatureFusion
    from uniir_amd.clip_model import CLIP_CONFIGS
    from uniir_amd.trainer import NativeTrainer
    dev = torch.device("cuda:0")
    config = SimpleNamespace(model=SimpleNamespace(gather_embeddings=True), data_config=SimpleNamespace(in_batch_neg_num=0))
    model = CLIPFeatureFusion("ViT-L/14", device=dev, config=config)
    tr = NativeTrainer(model, lr=1e-5, t_total=1000)
    batch = synth_batch(CLIP_CONFIGS["ViT-L/14"], a.pairs, 2023, dev)
    for _ in range(a.warmup):
        out = tr.train_step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = tr.train_step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"metric": "train_pairs_per_s (CLIP_FF ViT-L/14)", "value": a.pairs / dt, "unit": "pairs/s",
                      "ms_per_step": dt * 1e3, "pairs": a.pairs, "loss": float(out["loss"].detach()),
                      "peak_mem_GB": torch.cuda.max_memory_allocated() / 2**30}))


if __name__ == "__main__":
    main()
