"""Native train step for the CLIP_SF in-batch contrastive path on MI355X.

Mirrors UniIR src/models/uniir_clip/engine.py:19-50 (train_one_epoch inner loop: forward, loss /
accumulation_steps, backward, optimizer step every accumulation_steps, scheduler.step) and
clip_scorefusion/train.py:52-61,195-199,281-284 (AdamW lr 1e-5 betas (0.9,0.98) eps 1e-6, wd 0 for gains/biases and
0.2 for the rest, CosineAnnealingLR(T_max, eta_min=0)).  Differences, all result-preserving:
  * bf16 MFMA compute needs no GradScaler (the reference's fp16 autocast does);
  * gradients live in one flat fp32 buffer: DDP's bucketed all-reduce(mean) becomes one RCCL all-reduce(sum) of
    that buffer and a 1/world factor folded into the fused AdamW kernel;
  * AdamW is one fused kernel per weight-decay group and refreshes the bf16 weight shadow in the same pass.
"""
import math

import torch
import torch.distributed as dist

from . import ops


class CosineLR:
    def __init__(self, base_lr, t_total):
        self.base_lr, self.t_total, self.step_count = base_lr, max(1, t_total), 0

    def lr(self):
        return self.base_lr * (1 + math.cos(math.pi * self.step_count / self.t_total)) / 2

    def step(self):
        self.step_count += 1


class NativeTrainer:
    def __init__(self, model, lr=1e-5, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.2, t_total=1000,
                 accumulation_steps=1):
        self.model = model
        self.clip = model.clip_model
        self.betas, self.eps, self.wd = betas, eps, weight_decay
        self.sched = CosineLR(lr, t_total)
        self.accum = accumulation_steps
        self.micro = 0
        self.opt_step = 0
        self.m = self.v = None

    def _state(self):
        fl = self.clip._ensure_flat()
        if self.m is None or self.m.numel() != fl["total"] or self.m.device != fl["p32"].device:
            self.m = torch.zeros_like(fl["p32"])
            self.v = torch.zeros_like(fl["p32"])
        return fl

    def zero_grad(self):
        self.clip._ensure_flat()
        self.clip.zero_grad()

    def optimizer_step(self):
        fl = self._state()
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        if world > 1:
            dist.all_reduce(fl["g32"], op=dist.ReduceOp.SUM)   # RCCL; mean is folded into grad_scale below
        self.opt_step += 1
        lr = self.sched.lr()
        split, total = fl["split"], fl["total"]
        gs = 1.0 / world
        for lo, hi, wd in ((0, split, 0.0), (split, total, self.wd)):
            if hi > lo:
                ops.call("uniir_adamw_step", fl["p32"][lo:hi], fl["g32"][lo:hi], self.m[lo:hi], self.v[lo:hi],
                         fl["w16"][lo:hi], hi - lo, lr, self.betas[0], self.betas[1], self.eps, wd, self.opt_step, gs)
        self.clip._refresh_conv()
        self.sched.step()

    def train_step(self, batch):
        """one micro-batch: returns the reference's outputs dict (loss un-scaled, as logged by engine.py:48)."""
        if self.micro == 0:
            self.zero_grad()
        self.model.train()
        out = self.model(batch)
        (out["loss"] / self.accum).backward()
        self.micro += 1
        if self.micro == self.accum:
            self.optimizer_step()
            self.micro = 0
        return out
