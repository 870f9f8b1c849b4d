"""Native optimizer + train step for the CLIP_SF in-batch contrastive path on MI355X.

Mirrors UniIR src/models/uniir_clip/engine.py:19-50 (forward, loss / accumulation_steps, backward, optimizer step every
accumulation_steps, scheduler.step) and clip_scorefusion/train.py:52-61,195-199,281-284 (AdamW lr 1e-5 betas
(0.9,0.98) eps 1e-6, weight decay 0 for gains / biases / logit_scale and 0.2 for the rest,
CosineAnnealingLR(T_max, eta_min=0)).  Differences, all result-preserving:
  * bf16 MFMA compute needs no GradScaler (the reference's fp16 autocast does);
  * gradients live in one flat fp32 buffer: DDP's bucketed all-reduce(mean) becomes RCCL all-reduce(sum) of ranges of
    that buffer, launched per finished residual block while backward still runs (comm.GradReducer; one blocking
    all-reduce of the remainder at step()), with the 1/world factor folded into the fused AdamW kernel;
  * AdamW is one fused kernel per weight-decay group and refreshes the bf16 weight shadow in the same pass.
"""
import math

import torch

from . import comm, ops


class NativeAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics over the CLIP module's flat parameter buffer (two groups: [0, split) without weight
    decay, [split, total) with).  It is a real torch Optimizer (param_groups / lr schedulers / state_dict work)."""

    def __init__(self, clip_model, lr=1e-5, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.2, allreduce=True, extra=None,
                 overlap=True, bucket_bytes=64 << 20):
        """extra: further flat parameter stores, each one param group of its own -- objects with .params (list of
        nn.Parameter), .store() (-> FlatStore with p32 / g32 / w16_buf / total), .weight_decay and .lr (CLIP_FF's T5
        stack: clip_featurefusion/train.py:52-61 gives it weight decay 0.2 on everything and its own learning rate)"""
        self.clip = clip_model
        self.extra = list(extra or [])
        self.extra_mv = [None] * len(self.extra)
        if hasattr(clip_model, "optimizer_groups"):      # e.g. BLIPFeatureFusion: one group, uniform weight decay
            nd, d = clip_model.optimizer_groups()
        else:
            from .clip_model import _is_no_decay
            named = list(clip_model.named_parameters())
            nd = [p for n, p in named if _is_no_decay(n, p)]
            d = [p for n, p in named if not _is_no_decay(n, p)]
        groups = [{"params": nd, "weight_decay": 0.0}, {"params": d, "weight_decay": weight_decay}]
        groups += [{"params": e.params, "weight_decay": e.weight_decay, "lr": e.lr} for e in self.extra]
        super().__init__(groups, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.allreduce = allreduce
        self.overlap, self.bucket_bytes = overlap, bucket_bytes
        self.reducer = None
        self.last_collectives = 0
        self.m = self.v = None
        self.opt_step = 0

    def _buffers(self):
        fl = self.clip._ensure_flat()
        if self.m is None or self.m.numel() != fl["total"] or self.m.device != fl["p32"].device:
            m, v = torch.zeros_like(fl["p32"]), torch.zeros_like(fl["p32"])
            if self.m is not None and self.m.numel() == fl["total"]:
                m.copy_(self.m)
                v.copy_(self.v)
            self.m, self.v = m, v
        if self.allreduce and self.overlap and comm.world() > 1:
            if self.reducer is None or self.reducer.flat.data_ptr() != fl["g32"].data_ptr():
                self.reducer = comm.GradReducer(fl["g32"], self.bucket_bytes)
        else:
            self.reducer = None
        return fl

    def arm_overlap(self, last_micro_batch=True):
        """call before backward(): on the LAST micro-batch of an accumulation window the tower backward hands every finished
        residual block's weight gradients to the reducer (DDP's no_sync() on the earlier micro-batches)"""
        self._buffers()
        if self.reducer is not None:
            self.reducer.reset()           # an armed backward that never reached step() must not leak into this one
        self.clip._grad_reducer = self.reducer if last_micro_batch else None

    @torch.no_grad()
    def step(self, closure=None):
        fl = self._buffers()
        world = comm.world() if self.allreduce else 1
        extra_reduced = set()
        if world > 1:
            if self.reducer is not None:           # blocks already reduced during backward; now the remainder + wait
                self.last_collectives, extra_reduced = self.reducer.finish()
            else:
                comm.allreduce_sum_(fl["g32"])     # one RCCL all-reduce; the mean is folded into grad_scale
                self.last_collectives = 1
        self.clip._grad_reducer = None
        self.opt_step += 1
        split, total = fl["split"], fl["total"]
        # (lo, hi, param-group index); default: [0, split) without weight decay, [split, total) with
        ranges = fl.get("ranges") or [(0, split, 0), (split, total, 1)]
        for lo, hi, gi in ranges:
            group = self.param_groups[gi]
            if hi > lo:
                b1, b2 = group["betas"]
                ops.call("uniir_adamw_step", fl["p32"][lo:hi], fl["g32"][lo:hi], self.m[lo:hi], self.v[lo:hi],
                         fl["w16"][lo:hi], hi - lo, float(group["lr"]), b1, b2, group["eps"], group["weight_decay"],
                         self.opt_step, 1.0 / world)
        self.clip._refresh_conv()
        for i, (e, group) in enumerate(zip(self.extra, self.param_groups[2:])):
            st = e.store()
            if self.extra_mv[i] is None or self.extra_mv[i][0].numel() != st.total:
                self.extra_mv[i] = (torch.zeros_like(st.p32), torch.zeros_like(st.p32))
            elif self.extra_mv[i][0].device != st.p32.device:     # resumed from a checkpoint mapped to the CPU
                self.extra_mv[i] = tuple(t.to(st.p32.device) for t in self.extra_mv[i])
            if world > 1 and st.g32.data_ptr() not in extra_reduced:     # not handed to the reducer during backward
                comm.allreduce_sum_(st.g32)
                self.last_collectives += 1
            b1, b2 = group["betas"]
            m, v = self.extra_mv[i]
            ops.call("uniir_adamw_step", st.p32, st.g32, m, v, st.w16_buf, st.total, float(group["lr"]), b1, b2,
                     group["eps"], group["weight_decay"], self.opt_step, 1.0 / world)

    def zero_grad(self, set_to_none=False):
        self.clip._ensure_flat()
        self.clip.zero_grad()
        for e in self.extra:
            e.store().g32.zero_()

    def state_dict(self):
        return {"opt_step": self.opt_step, "exp_avg": self.m, "exp_avg_sq": self.v, "extra": self.extra_mv,
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        self.opt_step = sd["opt_step"]
        self.m, self.v = sd["exp_avg"], sd["exp_avg_sq"]
        self.extra_mv = list(sd.get("extra", self.extra_mv))
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)


class CosineLR:
    """closed form of torch CosineAnnealingLR(T_max=t_total, eta_min=0) driving NativeAdamW.param_groups"""

    def __init__(self, optimizer, t_total):
        self.opt, self.t_total, self.step_count = optimizer, max(1, t_total), 0
        self.base = [g["lr"] for g in optimizer.param_groups]

    def step(self):
        self.step_count += 1
        for g, b in zip(self.opt.param_groups, self.base):
            g["lr"] = b * (1 + math.cos(math.pi * self.step_count / self.t_total)) / 2

    def state_dict(self):
        return {"step_count": self.step_count, "base": self.base, "t_total": self.t_total}

    def load_state_dict(self, sd):
        self.step_count, self.base, self.t_total = sd["step_count"], sd["base"], sd["t_total"]
        self.step_count -= 1
        self.step()


class NativeTrainer:
    def __init__(self, model, lr=1e-5, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.2, t_total=1000,
                 accumulation_steps=1):
        self.model = model
        self.clip = model.clip_model
        extra = [model.t5_optimizer_group(lr=lr, weight_decay=weight_decay)] if hasattr(model, "t5_optimizer_group") else None
        self.opt = NativeAdamW(self.clip, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, extra=extra)
        self.sched = CosineLR(self.opt, t_total)
        self.accum = accumulation_steps
        self.micro = 0
        self._stash_reviewed = False

    def train_step(self, batch):
        """one micro-batch: returns the reference's outputs dict (loss un-scaled, as logged by engine.py:48)."""
        if self.micro == 0:
            self.opt.zero_grad()
        self.model.train()
        out = self.model(batch)
        self.opt.arm_overlap(self.micro == self.accum - 1)
        (out["loss"] / self.accum).backward()
        self.micro += 1
        if self.micro == self.accum:
            self.opt.step()
            self.sched.step()
            self.micro = 0
            if not self._stash_reviewed:          # once, after the first complete step (clip_model.CLIP.review_stash)
                self._stash_reviewed = True
                clip = getattr(self.model, "clip_model", None)
                if clip is not None and hasattr(clip, "review_stash"):
                    clip.review_stash(comm.all_reduce_min_float if comm.world() > 1 else None)
        return out
