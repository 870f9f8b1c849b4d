#!/usr/bin/env python
"""Where do BLIP_FF's ~500 small device-to-device copies per train step come from?  (profiles/r06_blip_kernel_stats.txt:
__amd_rocclr_copyBuffer 2033 calls / 4 steps.)  One step under torch.profiler, aten::copy_ / aten::contiguous / aten::clone / aten::cat
grouped by the innermost uniir_amd / bench frame.   python tools/r6/blip_copies.py"""
import collections
import os
import sys
import types

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from uniir_amd.blip_model import BLIPFeatureFusion  # noqa: E402
from uniir_amd.trainer import NativeAdamW  # noqa: E402

dev = torch.device("cuda:0")
model = BLIPFeatureFusion(med_config={}, vit="large", queue_size=57344, momentum=0.995,
                          config=types.SimpleNamespace(tokenizer_max_length=100)).to(dev)
model.check_masks = False
opt = NativeAdamW(model, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05, allreduce=False)
batches = [bench._blip_synth(64, 100, 30524, s, dev) for s in range(2)]


def step(i):
    opt.zero_grad()
    out = model(batches[i % 2], alpha=0.4)
    out["loss"].backward()
    opt.step()


for i in range(2):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(0)
    torch.cuda.synchronize()
agg = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::cat", "aten::index_select", "aten::to", "aten::_to_copy"):
        frame = next((f for f in (ev.stack or []) if ("uniir_amd" in f or "bench.py" in f) and "ops.py" not in f), (ev.stack or ["?"])[0] if ev.stack else "?")
        agg[(ev.name, frame.split("/")[-1][:110])] += 1
for (name, frame), n in agg.most_common(40):
    print(f"{n:5d}  {name:18s} {frame}")
kern = collections.Counter(ev.name[:60] for ev in prof.events() if ev.device_type is not None and str(ev.device_type).endswith("CUDA"))
print("device kernels:", [(k, v) for k, v in kern.most_common(12)])
