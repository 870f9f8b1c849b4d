#!/usr/bin/env python
"""Secondary benchmark: BLIP_FF large train step (ViT-L/16 @224 + MED BERT-base,
b pairs/rank, queue 57344, 100 text tokens, alpha 0.4) on one MI355X.  Prints one JSON line.
    python tools/bench_blip.py --pairs 256 --steps 5 --warmup 2"""
import argparse
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FLOP_PER_PAIR = 1.212e12     # item fwd 151.46 GFLOP x 2 items x (3 online + 1 momentum)


def synth(pairs, L, vocab, seed, device):
    g = torch.Generator().manual_seed(seed)
    M = 2 * pairs
    ids = torch.randint(1000, vocab - 2, (M, L), generator=g)
    ids[:, 0] = 101
    valid = torch.randint(5, L + 1, (M,), generator=g)
    mask = (torch.arange(L).unsqueeze(0) < valid.unsqueeze(1)).long()
    ids = ids * mask
    img = torch.randn(M, 3, 224, 224, generator=torch.Generator(device=device).manual_seed(seed), device=device)
    return {"txt_batched": types.SimpleNamespace(input_ids=ids.to(device), attention_mask=mask.to(device)),
            "image_batched": img,
            "p_did_list": torch.arange(pairs) + 1000 * seed,
            "index_mapping": {"query": [[2 * i] for i in range(pairs)], "pos_cand": [[2 * i + 1] for i in range(pairs)]}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=256)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--vit", default="large")
    ap.add_argument("--queue", type=int, default=57344)
    ap.add_argument("--len", type=int, default=100)
    a = ap.parse_args()
    from uniir_amd.blip_model import BLIPFeatureFusion
    from uniir_amd.trainer import NativeAdamW
    dev = torch.device("cuda:0")
    t0 = time.time()
    model = BLIPFeatureFusion(med_config={}, vit=a.vit, queue_size=a.queue, momentum=0.995,
                              config=types.SimpleNamespace(tokenizer_max_length=a.len)).to(dev)
    model.check_masks = False
    opt = NativeAdamW(model, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05, allreduce=False)
    batches = [synth(a.pairs, a.len, 30524, s, dev) for s in range(2)]
    print(f"# init {time.time() - t0:.1f}s", file=sys.stderr)

    def step(i):
        opt.zero_grad()
        out = model(batches[i % 2], alpha=0.4)
        out["loss"].backward()
        opt.step()
        return out

    for i in range(a.warmup):
        out = step(i)
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(a.steps):
        out = step(i)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / a.steps
    pps = a.pairs / dt
    print(json.dumps({"metric": "train_pairs_per_s", "value": pps, "unit": "pairs/s", "ms_per_step": dt * 1e3,
                      "config": {"workload": f"BLIP_FF vit-{a.vit}/16@224 + MED BERT-base train step", "pairs": a.pairs,
                                 "queue_size": a.queue, "text_len": a.len},
                      "mfma_frac_e2e": pps * FLOP_PER_PAIR / 2.5e15 if a.vit == "large" else None,
                      "loss": out["loss"].item(), "peak_mem_GB": torch.cuda.max_memory_allocated() / 2**30}))


if __name__ == "__main__":
    main()
