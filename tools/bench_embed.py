#!/usr/bin/env python
"""Secondary benchmark: forward-only embedding extraction, CLIP_SF ViT-L/14, the reference's
`model(batch, encode_mbeir_batch=True)` entry point + `.half()` per batch, synthetic items resident in HBM.
    python tools/bench_embed.py --items 2048 --steps 5"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "uniir_amd", "src"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--model", default="ViT-L/14")
    ap.add_argument("--precision", default=None, choices=(None, "fp16", "bf16"), help="16-bit type of the towers (default: the embedder's own)")
    a = ap.parse_args()
    from bench import synth_batch
    from models.uniir_clip.clip_scorefusion.clip_sf import CLIPScoreFusion
    from uniir_amd.clip_model import CLIP_CONFIGS
    dev = torch.device("cuda:0")
    model = CLIPScoreFusion(model_name=a.model, device=dev).float().eval()
    if a.precision:
        model.clip_model.precision = a.precision
    batch = synth_batch(CLIP_CONFIGS[a.model], a.items // 2, 2023, dev)
    batch["did_list"] = list(range(a.items))
    with torch.no_grad():
        emb, ids = model(batch, encode_mbeir_batch=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            emb, ids = model(batch, encode_mbeir_batch=True)
            out = emb.half()
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    fwd_flop = 175.33e9      # forward FLOPs per item, ViT-L/14 (vision 162.03 + text 13.30)
    print(json.dumps({"metric": "embedding items/s (CLIP_SF-L forward only)", "value": a.items / dt, "unit": "items/s",
                      "ms_per_batch": dt * 1e3, "items_per_batch": a.items, "out": list(out.shape), "precision": model.clip_model.precision,
                      "mfma_frac_e2e": a.items / dt * fwd_flop / 2.5e15 if a.model == "ViT-L/14" else None,
                      "peak_mem_GB": torch.cuda.max_memory_allocated() / 2**30}))


if __name__ == "__main__":
    main()
