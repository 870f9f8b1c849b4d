#!/usr/bin/env python
"""Times THE REFERENCE ITSELF on this container's CPU at BASELINE.json configs[0] (CLIP_SF ViT-B/32, batch 32, 1 process):
UniIR's own CLIPScoreFusion (clip_sf.py) + engine.train_one_epoch (engine.py:7-55) + the AdamW groups of
clip_scorefusion/train.py:52-61,195-199, imported read-only from /root/reference/src through the same `clip`-module shim the
golden generator uses (tests/golden/make_golden.py).  openai/CLIP is not installed anywhere offline, so the encoder behind
the shim is this repo's restatement of it (oracle/clip_oracle.py, pinned against transformers.CLIPModel by golden G5).
The reference never travels to the GPU box: this number exists only here and is recorded in BASELINE.md.
    python tools/time_reference_cpu.py [--threads N]"""
import argparse
import copy
import json
import os
import statistics
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ap.add_argument("--pairs", type=int, default=32)
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    import make_golden as MG            # installs nothing by itself; gives the stubs and the reference import path
    MG._install_stubs()
    torch.set_num_threads(a.threads)
    from oracle import clip_oracle as O
    from models.uniir_clip import engine
    from models.uniir_clip.clip_scorefusion.clip_sf import CLIPScoreFusion
    from torch import optim
    from torch.cuda.amp import GradScaler
    from torch.optim.lr_scheduler import CosineAnnealingLR
    cfg = O.CLIP_CONFIGS["ViT-B/32"]
    MG._STATE["model"] = O.OracleCLIP(cfg, seed=0)
    model = CLIPScoreFusion("stub", "cpu", config=MG._cfg(False))
    model.float()
    exclude = lambda n, p: p.ndim < 2 or any(s in n for s in ["bn", "ln", "bias", "logit_scale"])
    gain = [p for n, p in model.named_parameters() if exclude(n, p) and p.requires_grad]
    rest = [p for n, p in model.named_parameters() if not exclude(n, p) and p.requires_grad]
    opt = optim.AdamW([{"params": gain, "weight_decay": 0.0}, {"params": rest, "weight_decay": 0.2}], lr=1e-5,
                      betas=(0.9, 0.98), eps=1.0e-6)
    sched = CosineAnnealingLR(opt, T_max=1000, eta_min=0)
    config = types.SimpleNamespace(trainer_config=types.SimpleNamespace(print_freq=1000, gradient_accumulation_steps=1))
    batch = O.synthetic_batch(cfg, a.pairs, seed=2023)
    times = []
    for i in range(a.steps + 1):          # one warm-up epoch of one step, then timed ones
        t0 = time.perf_counter()
        engine.train_one_epoch(model, [copy.deepcopy(batch)], opt, i, "cpu", sched, i, GradScaler(enabled=False), config)
        dt = time.perf_counter() - t0
        if i:
            times.append(dt)
    t = statistics.median(times)
    print(json.dumps({"config": "BASELINE configs[0]: CLIP_SF ViT-B/32, batch 32, fp32, 1 process (reference clip_sf.py + engine.py + "
                                "AdamW, imported)", "pairs_per_s": round(a.pairs / t, 3), "s_per_step": round(t, 3),
                      "threads": a.threads, "cpu_count": os.cpu_count(), "steps_timed": len(times)}))


if __name__ == "__main__":
    main()
