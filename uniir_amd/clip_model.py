"""MI355X-native CLIP towers behind the openai/CLIP module surface that UniIR's clip_sf.py uses
(`clip.load()` -> model with .encode_image / .encode_text / .logit_scale, state-dict keys of the published
checkpoints: SURVEY.md section 3.2; reference call sites src/models/uniir_clip/clip_scorefusion/clip_sf.py:25-26,
44,47,66).

The arithmetic is entirely in libuniir_hip.so (include/uniir_hip.h): bf16 MFMA GEMMs with fused epilogues,
fp32 LayerNorm, fused attention, fp32 residual stream.  This file is host plumbing only: parameter bookkeeping
(one flat fp32 master buffer + flat grad buffer + bf16 shadow), the per-layer launch sequence of forward and
backward, and the activation stash.  There is no torch fallback.
"""
import contextlib
import ctypes as C
import math
import os

import torch
from torch import nn

from . import _lib, ops

# True (tests): run the CLIP towers as the per-op launch sequence below (what CLIP_FF and BLIP use for their variants) instead of
# the single-call C towers of csrc/tower.hip -- same kernels, same order
_PY_TOWERS = False
_SIDE_STREAMS = {}          # device index -> the torch stream that carries the text leg of two-stream towers (CLIP.side_leg)

ALIGN = 64  # elements; every parameter starts on a 256-B boundary of the flat buffers

def _tensor_version(t):
    try:
        return t._version
    except RuntimeError:            # inference tensors (torch.inference_mode) have no version counter: never cache on them
        return None


def text_row_offsets(tok):
    """(row_off int32 [n + 1] on the tokens' device, live rows on the host) of a token batch [n, ctx]: caption i owns
    argmax(tok[i]) + 1 rows (upstream pools at the EOT = the largest token id, first occurrence).  The host needs the total to size
    the GEMMs, which costs one device -> host read per NEW batch; a prefetcher that still has the tokens on the host can attach
    the lengths as `tok._uniir_lens` (+ `tok._uniir_lens_version`; host_utils.DevicePrefetcher does).  The hint is used only if it is
    a host tensor of n lengths in [1, ctx] attached to THIS version of the tensor -- anything else (tokens modified in place after the
    prefetch, a foreign attribute) falls back to the device read, so a stale hint can never size or index the packed kernels.  The
    result is remembered ON the tensor object (with its version counter), so a batch that is fed repeatedly (benchmarks, gradient
    accumulation replays) pays the read once; nothing is keyed by address, so a recycled allocation can never produce a stale answer."""
    ver = _tensor_version(tok)
    hit = getattr(tok, "_uniir_row_off", None)
    if hit is not None and ver is not None and hit[0] == ver and hit[1] == tuple(tok.shape):
        return hit[2]
    n, ctx = tok.shape
    lens = getattr(tok, "_uniir_lens", None)
    if lens is not None:
        ok = (isinstance(lens, torch.Tensor) and not lens.is_cuda and lens.dim() == 1 and lens.shape[0] == n
              and not lens.is_floating_point() and getattr(tok, "_uniir_lens_version", ver) == ver)
        if ok:
            lens = lens.to(torch.int64)
            ok = n == 0 or bool(((lens >= 1) & (lens <= ctx)).all())
        if not ok:
            lens = None
    if lens is None:
        lens = (tok.argmax(dim=-1) + 1).to(torch.int64).cpu()          # always in [1, ctx]
    off = torch.zeros(n + 1, dtype=torch.int64)
    off[1:] = torch.cumsum(lens, 0)
    out = (off.to(torch.int32).to(tok.device), int(off[-1]))
    if ver is not None:
        try:
            tok._uniir_row_off = (ver, tuple(tok.shape), out)
        except Exception:
            pass
    return out

CLIP_CONFIGS = {
    "ViT-B/32": dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768, vision_patch_size=32,
                     context_length=77, vocab_size=49408, transformer_width=512, transformer_heads=8,
                     transformer_layers=12),
    "ViT-B/16": dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768, vision_patch_size=16,
                     context_length=77, vocab_size=49408, transformer_width=512, transformer_heads=8,
                     transformer_layers=12),
    "ViT-L/14": dict(embed_dim=768, image_resolution=224, vision_layers=24, vision_width=1024, vision_patch_size=14,
                     context_length=77, vocab_size=49408, transformer_width=768, transformer_heads=12,
                     transformer_layers=12),
}


def _is_no_decay(name, p):
    # UniIR clip_scorefusion/train.py:195: p.ndim < 2 or name contains bn / ln / bias / logit_scale
    return p.ndim < 2 or any(s in name for s in ["bn", "ln", "bias", "logit_scale"])


class _P(nn.Module):
    """bare parameter holder so that state-dict keys match upstream module paths"""

    def __init__(self, **params):
        super().__init__()
        for k, v in params.items():
            self.register_parameter(k, nn.Parameter(v))


def _resblocks(width, layers, gen):
    proj_std = (width ** -0.5) * ((2 * layers) ** -0.5)
    attn_std = width ** -0.5
    fc_std = (2 * width) ** -0.5

    def rn(*shape, std):
        return torch.randn(*shape, generator=gen) * std

    blocks = nn.ModuleList()
    for _ in range(layers):
        blk = nn.Module()
        blk.attn = _P(in_proj_weight=rn(3 * width, width, std=attn_std), in_proj_bias=torch.zeros(3 * width))
        blk.attn.out_proj = _P(weight=rn(width, width, std=proj_std), bias=torch.zeros(width))
        blk.ln_1 = _P(weight=torch.ones(width), bias=torch.zeros(width))
        blk.mlp = nn.Module()
        blk.mlp.c_fc = _P(weight=rn(4 * width, width, std=fc_std), bias=torch.zeros(4 * width))
        blk.mlp.c_proj = _P(weight=rn(width, 4 * width, std=proj_std), bias=torch.zeros(width))
        blk.ln_2 = _P(weight=torch.ones(width), bias=torch.zeros(width))
        blocks.append(blk)
    t = nn.Module()
    t.resblocks = blocks
    return t


class CLIP(nn.Module):
    """Same attribute / key layout as upstream clip.model.CLIP (visual.*, transformer.*, token_embedding.weight, ...)."""

    def __init__(self, cfg, seed=0):
        super().__init__()
        self.cfg = dict(cfg)
        g = torch.Generator().manual_seed(seed)
        vw, tw, E, P = cfg["vision_width"], cfg["transformer_width"], cfg["embed_dim"], cfg["vision_patch_size"]
        grid = cfg["image_resolution"] // P
        scale = vw ** -0.5
        self.visual = nn.Module()
        self.visual.conv1 = _P(weight=torch.randn(vw, 3, P, P, generator=g) * (1.0 / (3 * P * P)) ** 0.5)
        self.visual.register_parameter("class_embedding", nn.Parameter(torch.randn(vw, generator=g) * scale))
        self.visual.register_parameter("positional_embedding",
                                       nn.Parameter(torch.randn(grid * grid + 1, vw, generator=g) * scale))
        self.visual.ln_pre = _P(weight=torch.ones(vw), bias=torch.zeros(vw))
        self.visual.transformer = _resblocks(vw, cfg["vision_layers"], g)
        self.visual.ln_post = _P(weight=torch.ones(vw), bias=torch.zeros(vw))
        self.visual.register_parameter("proj", nn.Parameter(torch.randn(vw, E, generator=g) * scale))
        self.transformer = _resblocks(tw, cfg["transformer_layers"], g)
        self.token_embedding = _P(weight=torch.randn(cfg["vocab_size"], tw, generator=g) * 0.02)
        self.positional_embedding = nn.Parameter(torch.randn(cfg["context_length"], tw, generator=g) * 0.01)
        self.ln_final = _P(weight=torch.ones(tw), bias=torch.zeros(tw))
        self.text_projection = nn.Parameter(torch.randn(tw, E, generator=g) * tw ** -0.5)
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / 0.07))
        # flat storage (built lazily on the device)
        self._flat = None
        self.kpad = (3 * P * P + 63) // 64 * 64
        # "bf16": MFMA towers, forward + backward (training / fast extraction).  "fp16": the same MFMA towers with fp16 operands and
        # activations, FORWARD ONLY -- the reference embedder's precision (mbeir_embedder.py:52-56: autocast(fp16) then .half()),
        # same speed as bf16 (the fp16 MFMA rate is the bf16 rate), 3 more mantissa bits: embeddings 1e-3 instead of 6e-3 from the
        # fp32 path; weights are read from an fp16 shadow that is refreshed with the bf16 one.  "fp32": the reference's model.float()
        # forward in exact-fp32 MFMA GEMMs + fp32 attention (csrc/fp32_path.hip): reference-precision embeddings, no backward
        self.precision = "bf16"
        # exact text-row packing (csrc/tower.hip *_packed): the text tower runs on the tokens up to each caption's EOT only.  Rows
        # behind the EOT never reach the pooled feature under the causal mask (clip_sf.py:43-44), so results are those of the dense
        # tower (embeddings and activation gradients bitwise, weight gradients up to the order of fp32 additions).
        self.pack_text = True
        # act(f) of every MLP kept per layer by the forward (uniir_clip_tower.stash_act) instead of being re-materialised by the c_proj
        # dgrad epilogue: +2 B x rows x 4 width per layer of workspace (52 GB at ViT-L/14 x 1024 items), bitwise the same results.
        # None = automatic, True / False = forced.  Automatic: decided ONCE per tower -- at the first training forward, from
        # the device's free memory with stash_margin_bytes to spare -- and then kept (the choice moves weight-gradient bits by fp32
        # summation order, so a run must not flip it from step to step or differ between ranks: see review_stash for W > 1);
        # an out-of-memory error on the stash allocation re-plans that tower without it; review_stash() revisits the choice after the
        # first complete step, when collectives' buffers and the optimizer state exist.
        self.stash_act = None
        # The last residual block of a POOLING tower on the pooled rows only (uniir_clip_tower.pool_last_block, csrc/tower.hip): the
        # class token / the EOT row is all that leaves the tower (clip_sf.py:44,47 -> upstream ln_post(x[:, 0, :]) /
        # x[arange, text.argmax(-1)]), so in the last block only ln_1 and the K | V projection need every row.  Exact: the same
        # embedding, the same parameter gradients without their zero terms.  False = every row through every sublayer (A/B, tests).
        self.pool_last_block = os.environ.get("UNIIR_POOL_LAST_BLOCK", "1") != "0"
        self.stash_margin_bytes = 16 << 30
        self.stash_review_headroom_bytes = 6 << 30
        self._stash_choice = {}           # tower -> bool: the automatic decision, made once (at the first training batch)
        self.stash_log = []               # human-readable record of every decision (bench.py prints it)
        self.last_stash_act = {}          # tower -> what the last training forward chose
        self.last_text_rows = None      # (live rows, dense rows) of the last packed text-tower call (bench: executed FLOPs)
        # The two towers of a batch are independent until the fusion: with overlap_towers the TEXT leg of a model forward (side_leg()
        # in clip_sf.encode_multimodal_input) is enqueued on a second HIP stream and joined before the fusion; autograd runs the
        # leg's backward on that stream again (nodes run on the stream of their forward) and the tower backward joins it to the
        # stream the leg forked from.  The text tower's small GEMMs (1.6 rounds of 256-tiles at 35 k rows) then run on the compute
        # units the image tower's kernels leave idle in their last round of tiles and between launches.  No kernel changes and the
        # towers' buffers are disjoint (split-K slabs and workspaces are per tower): losses and embeddings are bitwise those of the
        # one-stream order, gradients equal up to the order of the fp32 atomic column sums, which neither order fixes.
        # True / False = forced (UNIIR_OVERLAP_TOWERS=1 / 0); None = automatic: ON in a single-rank job, OFF when a process group
        # with more than one rank exists.  With W > 1 the overlapped gradient reducer (comm.GradReducer) hands ranges to the
        # collective library from whichever stream announces them, and RCCL's async all-reduce orders against the CURRENT stream only:
        # that ordering has run over gloo (host copies, synchronous) but never over RCCL -- no two-GPU box in this environment -- so
        # a multi-rank job keeps the one-stream order until it has (bench.py --overlap-towers forces it for an A/B).
        env = os.environ.get("UNIIR_OVERLAP_TOWERS")
        self.overlap_towers = None if env is None else env != "0"
        self._side_streams = _SIDE_STREAMS          # one second stream per DEVICE, shared by every model of the process
        self._leg_main = None           # inside side_leg(): the stream the leg forked from

    # ---- flat parameter / gradient / bf16-shadow storage -----------------------------------------------------
    def _ensure_flat(self):
        params = list(self.named_parameters())
        dev = self.logit_scale.device
        if dev.type != "cuda":
            raise RuntimeError("uniir_amd CLIP runs on an MI355X only (no CPU path); move the model to cuda")
        fl = self._flat
        if fl is not None and fl["dev"] == dev and all(p.data_ptr() == fl["p32"].data_ptr() + 4 * fl["off"][n]
                                                        for n, p in params):
            return fl
        order = [(n, p) for n, p in params if _is_no_decay(n, p)] + [(n, p) for n, p in params if not _is_no_decay(n, p)]
        n_nodecay = sum(1 for n, p in params if _is_no_decay(n, p))
        off, cur, split = {}, 0, 0
        for i, (n, p) in enumerate(order):
            if i == n_nodecay:
                split = cur
            off[n] = cur
            cur += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        if n_nodecay == len(order):
            split = cur
        p32 = torch.zeros(cur, device=dev, dtype=torch.float32)
        g32 = torch.zeros(cur, device=dev, dtype=torch.float32)
        for n, p in order:
            view = p32[off[n]:off[n] + p.numel()].view(p.shape)
            view.copy_(p.data.float())
            p.data = view
            p.grad = g32[off[n]:off[n] + p.numel()].view(p.shape)
        w16 = torch.empty(cur, device=dev, dtype=torch.bfloat16)
        self._flat = fl = dict(dev=dev, p32=p32, g32=g32, w16=w16, off=off, total=cur, split=split, version=-1,
                               shapes={n: p.shape for n, p in order})
        self._conv16 = torch.zeros(self.cfg["vision_width"], self.kpad, device=dev, dtype=torch.bfloat16)
        self._dconv = torch.zeros(self.cfg["vision_width"], self.kpad, device=dev, dtype=torch.float32)
        self.refresh_shadow()
        return fl

    def refresh_shadow(self):
        """bf16 copies of the master weights (called after every optimizer step; the fused AdamW does it itself)."""
        fl = self._flat
        ops.call("uniir_cast_f32_to_bf16", fl["p32"], fl["w16"], fl["total"])
        self._refresh_conv()
        fl["version"] = self._param_version()
        fl["h16_version"] = None            # the fp16 shadow (precision = "fp16") is rebuilt on demand

    def _sync_half_shadow(self):
        """fp16 copies of the master weights for the fp16 forward: built on first use, rebuilt when the parameters moved"""
        fl = self._sync_shadow()
        ver = self._param_version()
        if fl.get("h16") is None:
            fl["h16"] = torch.empty(fl["total"], device=fl["dev"], dtype=torch.float16)
            self._conv16h = torch.zeros(self.cfg["vision_width"], self.kpad, device=fl["dev"], dtype=torch.float16)
        if fl.get("h16_version") != ver:
            ops.call("uniir_cast_f32_to_f16", fl["p32"], fl["h16"], fl["total"])
            vw, P = self.cfg["vision_width"], self.cfg["vision_patch_size"]
            ops.call("uniir_cast_pad_rows_f16", self.visual.conv1.weight.data, self._conv16h, vw, 3 * P * P, self.kpad)
            fl["h16_version"] = ver
        return fl

    def _refresh_conv(self):
        vw, P = self.cfg["vision_width"], self.cfg["vision_patch_size"]
        ops.call("uniir_cast_pad_rows", self.visual.conv1.weight.data, self._conv16, vw, 3 * P * P, self.kpad)

    def _param_version(self):
        return sum(p._version for _, p in self.named_parameters())

    def _sync_shadow(self):
        fl = self._ensure_flat()
        if fl["version"] != self._param_version():
            self.refresh_shadow()
        return fl

    def w16(self, name):
        fl = self._flat
        o = fl["off"][name]
        return fl["w16"][o:o + math.prod(fl["shapes"][name])].view(fl["shapes"][name])

    def grad_view(self, name):
        fl = self._flat
        o = fl["off"][name]
        return fl["g32"][o:o + math.prod(fl["shapes"][name])].view(fl["shapes"][name])

    def tower_desc(self, which, half=False):
        """the POD description uniir_clip_tower_{fwd,bwd} take (include/uniir_hip.h [TOWER]): raw device pointers into the flat
        fp32 / bf16-shadow / gradient buffers; cached until the flat storage is rebuilt.  half: the fp16 forward's description
        (dtype16 = 1, 16-bit weights from the fp16 shadow)"""
        import ctypes as C
        from ._lib import ClipBlock, ClipTower
        fl = self._flat
        key = (which, fl["p32"].data_ptr(), bool(half))
        cache = fl.setdefault("tower_desc", {})
        if key in cache:
            return cache[key][0]
        cfg = self.cfg

        def p(name):
            return fl["p32"].data_ptr() + 4 * fl["off"][name]

        def h(name):
            return (fl["h16"] if half else fl["w16"]).data_ptr() + 2 * fl["off"][name]

        def g(name):
            return fl["g32"].data_ptr() + 4 * fl["off"][name]

        image = which == "image"
        prefix = "visual.transformer" if image else "transformer"
        L = cfg["vision_layers"] if image else cfg["transformer_layers"]
        W = cfg["vision_width"] if image else cfg["transformer_width"]
        blocks = (ClipBlock * L)()
        for i in range(L):
            b, r = blocks[i], f"{prefix}.resblocks.{i}"
            b.ln1_w, b.ln1_b, b.ln2_w, b.ln2_b = p(f"{r}.ln_1.weight"), p(f"{r}.ln_1.bias"), p(f"{r}.ln_2.weight"), p(f"{r}.ln_2.bias")
            b.wqkv16, b.wo16 = h(f"{r}.attn.in_proj_weight"), h(f"{r}.attn.out_proj.weight")
            b.wfc16, b.wproj16 = h(f"{r}.mlp.c_fc.weight"), h(f"{r}.mlp.c_proj.weight")
            b.bqkv, b.bo = p(f"{r}.attn.in_proj_bias"), p(f"{r}.attn.out_proj.bias")
            b.bfc, b.bproj = p(f"{r}.mlp.c_fc.bias"), p(f"{r}.mlp.c_proj.bias")
            b.g_ln1_w, b.g_ln1_b, b.g_ln2_w, b.g_ln2_b = g(f"{r}.ln_1.weight"), g(f"{r}.ln_1.bias"), g(f"{r}.ln_2.weight"), g(f"{r}.ln_2.bias")
            b.g_wqkv, b.g_bqkv = g(f"{r}.attn.in_proj_weight"), g(f"{r}.attn.in_proj_bias")
            b.g_wo, b.g_bo = g(f"{r}.attn.out_proj.weight"), g(f"{r}.attn.out_proj.bias")
            b.g_wfc, b.g_bfc = g(f"{r}.mlp.c_fc.weight"), g(f"{r}.mlp.c_fc.bias")
            b.g_wproj, b.g_bproj = g(f"{r}.mlp.c_proj.weight"), g(f"{r}.mlp.c_proj.bias")
        t = ClipTower()
        t.dtype16 = int(bool(half))
        t.is_text, t.layers, t.width, t.embed_dim = int(not image), L, W, cfg["embed_dim"]
        t.blocks = C.cast(blocks, C.POINTER(ClipBlock))
        # split-K slabs: one scratch buffer PER TOWER (the two towers may run on two streams at the same time)
        skw = ops._splitk_workspace(fl["dev"], 128 << 20, tag=which)
        t.splitk_ws, t.splitk_ws_bytes = skw.data_ptr(), skw.numel()
        if image:
            P = cfg["vision_patch_size"]
            t.heads, t.tokens = W // 64, (cfg["image_resolution"] // P) ** 2 + 1
            t.resolution, t.patch, t.kpad = cfg["image_resolution"], P, self.kpad
            t.conv16 = (self._conv16h if half else self._conv16).data_ptr()
            t.class_emb, t.pos_emb = p("visual.class_embedding"), p("visual.positional_embedding")
            t.ln_pre_w, t.ln_pre_b = p("visual.ln_pre.weight"), p("visual.ln_pre.bias")
            t.ln_post_w, t.ln_post_b, t.proj16 = p("visual.ln_post.weight"), p("visual.ln_post.bias"), h("visual.proj")
            t.g_conv, t.g_class, t.g_pos = g("visual.conv1.weight"), g("visual.class_embedding"), g("visual.positional_embedding")
            t.g_ln_pre_w, t.g_ln_pre_b = g("visual.ln_pre.weight"), g("visual.ln_pre.bias")
            t.g_ln_post_w, t.g_ln_post_b, t.g_proj = g("visual.ln_post.weight"), g("visual.ln_post.bias"), g("visual.proj")
        else:
            t.heads, t.tokens, t.vocab = cfg["transformer_heads"], cfg["context_length"], cfg["vocab_size"]
            t.pos_emb, t.token_emb = p("positional_embedding"), p("token_embedding.weight")
            t.ln_post_w, t.ln_post_b, t.proj16 = p("ln_final.weight"), p("ln_final.bias"), h("text_projection")
            t.g_pos, t.g_token = g("positional_embedding"), g("token_embedding.weight")
            t.g_ln_post_w, t.g_ln_post_b, t.g_proj = g("ln_final.weight"), g("ln_final.bias"), g("text_projection")
        cache[key] = (t, blocks, skw)          # keep the host array and the scratch alive with the description
        return t

    def layer_grad_range(self, prefix, i):
        """[lo, hi) of the flat buffers that holds the four weight matrices of residual block i (adjacent: the weight-decay
        section keeps named_parameters() order and the block's gains / biases live in the other section)"""
        fl = self._flat
        names = [f"{prefix}.resblocks.{i}.{n}" for n in ("attn.in_proj_weight", "attn.out_proj.weight", "mlp.c_fc.weight",
                                                         "mlp.c_proj.weight")]
        lo = min(fl["off"][n] for n in names)
        hi = max(fl["off"][n] + (math.prod(fl["shapes"][n]) + ALIGN - 1) // ALIGN * ALIGN for n in names)
        if hi - lo != sum((math.prod(fl["shapes"][n]) + ALIGN - 1) // ALIGN * ALIGN for n in names):
            raise RuntimeError("flat layout: block weights are not adjacent")
        return lo, hi

    def zero_grad(self, set_to_none=False):
        """Gradients live in one flat buffer; zeroing it is one memset (they are never set to None)."""
        if self._flat is not None:
            self._flat["g32"].zero_()
            for n, p in self.named_parameters():
                if p.grad is None or p.grad.data_ptr() != self.grad_view(n).data_ptr():
                    p.grad = self.grad_view(n)
        else:
            super().zero_grad(set_to_none=set_to_none)

    def review_stash(self, all_reduce_min=None):
        """Call after the first COMPLETE training step (trainer.NativeTrainer does): the automatic act(f) stash was sized before the
        first backward, i.e. before RCCL's channel buffers, the reducer's in-flight buckets and the optimizer state existed.  If the
        step's measured peak (torch's peak + what lives outside torch's allocator on this device) leaves less than
        stash_review_headroom_bytes of the device, the stash is switched off for the following steps -- on EVERY rank when
        all_reduce_min (a callable reducing a python float with MIN over the ranks) is given, so that replicas keep computing bitwise
        the same gradients.  Returns the headroom in bytes (None when nothing was decided automatically)."""
        # Only RANK-UNIFORM conditions may return before the collectives: the automatic decision itself is per rank (each rank's own
        # free memory, an out-of-memory re-plan), so a rank that chose "off" for every tower must still take part in both reductions
        # -- skipping them would leave its peers' MIN all-reduces to pair with this rank's next gradient bucket.
        if self.stash_act is not None or self._flat is None:
            return None
        mine = any(self._stash_choice.values())
        if all_reduce_min is None and not mine:
            return None
        headroom, peak, outside = self._measured_headroom() if mine else (float("inf"), 0.0, 0.0)
        agreed = 1.0 if (mine and all(self._stash_choice.values())) else 0.0
        if all_reduce_min is not None:          # one mode on every rank from here on
            headroom = float(all_reduce_min(headroom))
            agreed = float(all_reduce_min(agreed))
        if not mine:
            if self._stash_choice:
                self.stash_log.append("act(f) stash was off on this rank already; took part in the ranks' review")
            return None
        if headroom < self.stash_review_headroom_bytes or agreed == 0.0:
            for k in self._stash_choice:
                self._stash_choice[k] = False
            self.stash_log.append(f"act(f) stash switched off after the first step: measured headroom {headroom / 2**30:.1f} GiB "
                                  f"(floor {self.stash_review_headroom_bytes / 2**30:.0f} GiB), ranks agreed on the stash: {bool(agreed)}")
        else:
            self.stash_log.append(f"act(f) stash kept after the first step: measured headroom {headroom / 2**30:.1f} GiB "
                                  f"(peak {peak / 2**30:.1f} GiB in torch, {outside / 2**30:.1f} GiB outside)")
        return headroom

    def _measured_headroom(self):
        """(bytes of the device the finished step left unused at its peak, torch's peak, bytes held outside torch's allocator)"""
        dev = self._flat["dev"]
        free, total = torch.cuda.mem_get_info(dev)
        outside = total - free - torch.cuda.memory_reserved(dev)          # other processes / RCCL / HIP runtime on this device
        peak = torch.cuda.max_memory_allocated(dev)
        return float(total - outside - peak), float(peak), float(outside)

    # ---- two-stream towers -------------------------------------------------------------------------------------------
    def overlap_on(self):
        """overlap_towers resolved: forced value, or (automatic) on exactly when this is a single-rank job"""
        if self.overlap_towers is not None:
            return bool(self.overlap_towers)
        import torch.distributed as dist
        return not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)

    @contextlib.contextmanager
    def side_leg(self, dev):
        """with model.side_leg(dev) as leg: everything enqueued inside (one tower call and the torch ops around it) goes to this
        model's second stream of `dev`, ordered after all work enqueued so far on the current stream; leg is None when the overlap is
        off (CPU tensors, overlap_towers False, the Python tower sequences) and the body then runs on the current stream as before.
        The caller hands the tensors the leg produced to join_leg() before the current stream reads them."""
        dev = torch.device(dev)
        if not self.overlap_on() or dev.type != "cuda" or _PY_TOWERS or self.precision == "fp32" or self._leg_main is not None:
            yield None
            return
        # every lazily built shared buffer (flat storage, 16-bit shadows of the weights) is brought up to date on the forking stream,
        # BEFORE the fork: both towers read them
        (self._sync_half_shadow if self.precision == "fp16" else self._sync_shadow)()
        main = torch.cuda.current_stream(dev)
        side = self._side_streams.get(dev.index)
        if side is None:
            side = self._side_streams[dev.index] = torch.cuda.Stream(dev)
        side.wait_stream(main)
        self._leg_main = main
        try:
            with torch.cuda.stream(side):
                yield side
        finally:
            self._leg_main = None

    @staticmethod
    def join_leg(leg, *tensors):
        """the current stream waits for everything side_leg() enqueued; `tensors` (allocated inside the leg) are about to be read on
        the current stream"""
        if leg is None:
            return
        main = torch.cuda.current_stream(leg.device)
        main.wait_stream(leg)
        for t in tensors:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(main)

    # ---- public encoder API (upstream names) ---------------------------------------------------------------------
    @property
    def dtype(self):
        return torch.float32

    def encode_image(self, image):
        self._sync_shadow()
        return _TowerFn.apply(self, "image", image.contiguous(), self._anchor_for(image.device))

    def encode_text(self, text):
        self._sync_shadow()
        tok = text.to(torch.int32).contiguous()
        if self.pack_text and tok is not text:       # the caller's tensor object carries the remembered row offsets
            ver = _tensor_version(tok)
            if ver is not None:
                tok._uniir_row_off = (ver, tuple(tok.shape), text_row_offsets(text))
        return _TowerFn.apply(self, "text", tok, self._anchor_for(text.device))

    def _anchor_for(self, dev):
        # a leaf that requires grad, so autograd calls the tower backward (parameter grads are written straight
        # into the flat gradient buffer, they do not travel through autograd)
        return torch.zeros(1, device=dev, requires_grad=torch.is_grad_enabled())


# ------------------------------------------------------------------------------------------------------------
# layer launch sequences
# ------------------------------------------------------------------------------------------------------------
class _Blk:
    """names of one residual block's tensors inside the flat buffers"""

    def __init__(self, model, prefix, names=None):
        p = prefix
        self.names = names or dict(wqkv=f"{p}.attn.in_proj_weight", bqkv=f"{p}.attn.in_proj_bias", wo=f"{p}.attn.out_proj.weight",
                          bo=f"{p}.attn.out_proj.bias", ln1w=f"{p}.ln_1.weight", ln1b=f"{p}.ln_1.bias",
                          wfc=f"{p}.mlp.c_fc.weight", bfc=f"{p}.mlp.c_fc.bias", wproj=f"{p}.mlp.c_proj.weight",
                          bproj=f"{p}.mlp.c_proj.bias", ln2w=f"{p}.ln_2.weight", ln2b=f"{p}.ln_2.bias")
        self.m = model

    def w16(self, k):
        return self.m.w16(self.names[k])

    def p32(self, k):
        fl = self.m._flat
        n = self.names[k]
        o = fl["off"][n]
        return fl["p32"][o:o + math.prod(fl["shapes"][n])].view(fl["shapes"][n])

    def g(self, k):
        return self.m.grad_view(self.names[k])


def _tower_fwd(model, prefix, layers, x, M, T, W, heads, causal, save, eps=1e-5, act=ops.ACT_QUICKGELU, blk=None,
               rowscale=None):
    """pre-LN residual blocks (CLIP resblocks; with eps / act / blk(i) overridden also the BLIP ViT blocks).
    rowscale fp32 [layers, 2, M]: DropPath factors (0 or 1/keep per item) of the two residual branches of each block
    (BLIP ViT-large in train mode, backbone/vit.py:79-80); None = no DropPath (the residual add stays in the GEMM)"""
    blk = blk or (lambda i: _Blk(model, f"{prefix}.resblocks.{i}"))
    R = M * T
    dev = x.device
    h = torch.empty(R, W, device=dev, dtype=torch.bfloat16)
    g = torch.empty(R, 4 * W, device=dev, dtype=torch.bfloat16)
    saved = []
    # DropPath factors per ROW (item factor repeated over its T tokens): applied inside the residual GEMM's epilogue
    rs_rows = None if rowscale is None else rowscale.repeat_interleave(T, dim=2).contiguous()
    for i in range(layers):
        b = blk(i)
        # with save, the two LayerNorm outputs are kept for the weight gradients (26 GB at ViT-L/14 x 1024 items:
        # cheaper than re-reading the fp32 stream to recompute them in backward)
        h1 = torch.empty(R, W, device=dev, dtype=torch.bfloat16) if save else h
        ops.layernorm_fwd(x, b.p32("ln1w"), b.p32("ln1b"), eps, out_bf16=h1, rows=R, width=W)
        qkv = ops.linear_fwd(h1, b.w16("wqkv"), b.p32("bqkv"))
        ao, lse = ops.attention_fwd(qkv, M, T, heads, causal)
        if rowscale is None:
            x2 = ops.linear_fwd(ao, b.w16("wo"), b.p32("bo"), epilogue=ops.EPI_RESID_F32, resid=x)
        else:
            x2 = ops.linear_fwd(ao, b.w16("wo"), b.p32("bo"), epilogue=ops.EPI_RESID_F32, resid=x, row_scale=rs_rows[i, 0])
        h2 = torch.empty(R, W, device=dev, dtype=torch.bfloat16) if save else h
        ops.layernorm_fwd(x2, b.p32("ln2w"), b.p32("ln2b"), eps, out_bf16=h2, rows=R, width=W)
        if save:
            f = torch.empty(R, 4 * W, device=dev, dtype=torch.bfloat16)
            ops.linear_fwd(h2, b.w16("wfc"), b.p32("bfc"), out=f, epilogue=ops.EPI_BIAS_ACT, C2=g, act=act)
        else:       # forward only: the pre-activation is not needed, only act(f) is written
            f = None
            ops.linear_fwd(h2, b.w16("wfc"), b.p32("bfc"), out=g, epilogue=ops.EPI_ACT_ONLY, act=act)
        if rowscale is None:
            xn = ops.linear_fwd(g, b.w16("wproj"), b.p32("bproj"), epilogue=ops.EPI_RESID_F32, resid=x2)
        else:
            xn = ops.linear_fwd(g, b.w16("wproj"), b.p32("bproj"), epilogue=ops.EPI_RESID_F32, resid=x2, row_scale=rs_rows[i, 1])
        if save:
            saved.append((x, qkv, ao, lse, x2, f, h1, h2))
        x = xn
    return x, saved


def _tower_bwd(model, prefix, layers, dx, dxb, saved, M, T, W, heads, causal, eps=1e-5, act=ops.ACT_QUICKGELU,
               blk=None, rowscale=None):
    """dx fp32 [R,W] and its bf16 copy dxb: gradient w.r.t. the tower output.  Returns d(tower input) (fp32).
    With DropPath factors (rowscale, see _tower_fwd) the gradient entering a branch is rowscale * dx: the LayerNorm backward
    that produces dx writes its bf16 copy and the branch's bias gradient already scaled (uniir_layernorm_bwd_ex); only the
    tower's incoming gradient is scaled by a separate pass."""
    # DDP overlap: with a reducer armed (trainer.NativeAdamW.arm_overlap) a finished block's weight gradients go to the
    # collective stream while the remaining blocks still run (only the stock CLIP block naming has a range lookup)
    reducer = getattr(model, "_grad_reducer", None) if blk is None else None
    blk = blk or (lambda i: _Blk(model, f"{prefix}.resblocks.{i}"))
    R = M * T
    dev = dx.device
    g = torch.empty(R, 4 * W, device=dev, dtype=torch.bfloat16)
    df = torch.empty(R, 4 * W, device=dev, dtype=torch.bfloat16)
    dh = torch.empty(R, W, device=dev, dtype=torch.bfloat16)
    # bias gradient of the last block's c_proj: column sums of the incoming gradient (the other blocks get theirs from
    # the LayerNorm backward that produces their incoming gradient)
    rs_rows = None if rowscale is None else rowscale.repeat_interleave(T, dim=2).contiguous()     # factor per row
    if rowscale is not None:
        ops.dropout_bf16_(dxb, 0.0, 0, rowscale=rowscale[layers - 1, 1], rows_per_scale=T)
    ops.call("uniir_colsum_bf16", dxb, W, blk(layers - 1).g("bproj"), R, W)
    for i in reversed(range(layers)):
        b = blk(i)
        x, qkv, ao, lse, x2, f, h1, h2 = saved[i]
        saved[i] = None
        # d(mlp): df = (dx @ Wproj) * act'(f); the same epilogue re-materialises g = act(f) for dWproj and sums
        # df's columns into the c_fc bias gradient
        ops.linear_dgrad(dxb, b.w16("wproj"), out=df, aux=f, act_out=g, colsum=b.g("bfc"), act=act)
        ops.linear_wgrad(dxb, g, b.g("wproj"))
        ops.linear_wgrad(df, h2, b.g("wfc"))
        ops.linear_dgrad(df, b.w16("wfc"), out=dh)                               # dh := d ln_2 out
        dx2 = torch.empty(R, W, device=dev, dtype=torch.float32)
        ops.layernorm_bwd(x2, b.p32("ln2w"), dh, b.g("ln2w"), b.g("ln2b"), eps, dres=dx, dx=dx2, dx_bf16=dxb,
                          rows=R, width=W, dx_colsum=b.g("bo"),             # d x2 also is d(out_proj out): its bias grad
                          branch_scale=None if rs_rows is None else rs_rows[i, 0])
        del x2, f, h2
        ops.linear_wgrad(dxb, ao, b.g("wo"))
        ops.linear_dgrad(dxb, b.w16("wo"), out=dh)                               # dh := d attn out
        dqkv = ops.attention_bwd(qkv, ao, dh, lse, M, T, heads, causal)
        del qkv, ao, lse
        ops.linear_wgrad(dqkv, h1, b.g("wqkv"), dbias=b.g("bqkv"))             # + the in_proj bias gradient, same pass over dqkv
        ops.linear_dgrad(dqkv, b.w16("wqkv"), out=dh)                            # dh := d ln_1 out
        del dqkv, h1
        ops.layernorm_bwd(x, b.p32("ln1w"), dh, b.g("ln1w"), b.g("ln1b"), eps, dres=dx2, dx=dx, dx_bf16=dxb,
                          rows=R, width=W, dx_colsum=(blk(i - 1).g("bproj") if i > 0 else None),
                          branch_scale=None if (rs_rows is None or i == 0) else rs_rows[i - 1, 1])
        del x, dx2
        if reducer is not None:
            reducer.ready(*model.layer_grad_range(prefix, i))
    return dx


# ------------------------------------------------------------------------------------------------------------
# fp32 forward (precision = "fp32"): the same block sequence on uniir_sgemm (exact-fp32 MFMA) + uniir_bias_act_f32 +
# uniir_attention_f32_fwd + the fp32 LayerNorm kernel.  openai/CLIP model.py semantics as restated in SURVEY.md 3.2.
# ------------------------------------------------------------------------------------------------------------
def _linear_f32(x, w, bias=None, *, act=-1, resid=None):
    """y[M,N] = act(x[M,K] @ w[N,K]^T + bias) (+ resid), all fp32"""
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, device=x.device, dtype=torch.float32)
    ops.call("uniir_sgemm", x, K, 1, w, 1, K, y, N, M, N, K, 1.0)
    if bias is not None or resid is not None or act >= 0:
        ops.call("uniir_bias_act_f32", y, bias, resid, M, N, act)
    return y


def _tower_fwd_f32(model, prefix, layers, x, M, T, W, heads, causal, p32):
    R = M * T
    for i in range(layers):
        p = f"{prefix}.resblocks.{i}"
        h = torch.empty(R, W, device=x.device, dtype=torch.float32)
        ops.layernorm_fwd(x, p32(f"{p}.ln_1.weight"), p32(f"{p}.ln_1.bias"), out_f32=h, rows=R, width=W)
        qkv = _linear_f32(h, p32(f"{p}.attn.in_proj_weight"), p32(f"{p}.attn.in_proj_bias"))
        ao = torch.empty(R, W, device=x.device, dtype=torch.float32)
        ops.call("uniir_attention_f32_fwd", qkv, 3 * W, qkv[:, W:], qkv[:, 2 * W:], 3 * W, ao, W, None, M, T, T, heads,
                 int(causal), 0.125)
        x = _linear_f32(ao, p32(f"{p}.attn.out_proj.weight"), p32(f"{p}.attn.out_proj.bias"), resid=x)
        ops.layernorm_fwd(x, p32(f"{p}.ln_2.weight"), p32(f"{p}.ln_2.bias"), out_f32=h, rows=R, width=W)
        g = _linear_f32(h, p32(f"{p}.mlp.c_fc.weight"), p32(f"{p}.mlp.c_fc.bias"), act=ops.ACT_QUICKGELU)
        x = _linear_f32(g, p32(f"{p}.mlp.c_proj.weight"), p32(f"{p}.mlp.c_proj.bias"), resid=x)
    return x


def _encode_fp32(model, which, inp, p32):
    cfg = model.cfg
    dev = inp.device
    E, M = cfg["embed_dim"], inp.shape[0]
    if which == "image":
        W, P, L = cfg["vision_width"], cfg["vision_patch_size"], cfg["vision_layers"]
        res = cfg["image_resolution"]
        g = res // P
        T = g * g + 1
        # im2col is a pure permutation (no arithmetic): [M,3,g,P,g,P] -> [M*g*g, 3*P*P], the order conv1.weight.view(W,-1) has
        patches = inp.float().view(M, 3, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(M * g * g, 3 * P * P).contiguous()
        po = _linear_f32(patches, p32("visual.conv1.weight").view(W, 3 * P * P))
        x0 = torch.empty(M * T, W, device=dev, dtype=torch.float32)
        ops.call("uniir_vit_assemble_f32", po, p32("visual.class_embedding"), p32("visual.positional_embedding"), x0, M, T, W)
        x = torch.empty(M * T, W, device=dev, dtype=torch.float32)
        ops.layernorm_fwd(x0, p32("visual.ln_pre.weight"), p32("visual.ln_pre.bias"), out_f32=x, rows=M * T, width=W)
        x = _tower_fwd_f32(model, "visual.transformer", L, x, M, T, W, W // 64, False, p32)
        rows = torch.empty(M, W, device=dev, dtype=torch.float32)
        ops.call("uniir_gather_rows", x, None, rows, M, T, W)
        lnw, lnb, proj = "visual.ln_post.weight", "visual.ln_post.bias", "visual.proj"
    else:
        W, L, T = cfg["transformer_width"], cfg["transformer_layers"], cfg["context_length"]
        x = torch.empty(M * T, W, device=dev, dtype=torch.float32)
        eot = torch.empty(M, device=dev, dtype=torch.int32)
        ops.call("uniir_text_embed", inp, p32("token_embedding.weight"), p32("positional_embedding"), x, eot, M, T, W,
                 cfg["vocab_size"])
        x = _tower_fwd_f32(model, "transformer", L, x, M, T, W, cfg["transformer_heads"], True, p32)
        rows = torch.empty(M, W, device=dev, dtype=torch.float32)
        ops.call("uniir_gather_rows", x, eot, rows, M, T, W)
        lnw, lnb, proj = "ln_final.weight", "ln_final.bias", "text_projection"
    pooled = torch.empty(M, W, device=dev, dtype=torch.float32)
    ops.layernorm_fwd(rows, p32(lnw), p32(lnb), out_f32=pooled, rows=M, width=W)
    emb = torch.empty(M, E, device=dev, dtype=torch.float32)
    ops.call("uniir_sgemm", pooled, W, 1, p32(proj), E, 1, emb, E, M, E, W, 1.0)      # pooled @ proj, proj [W, E]
    return emb


def _join_backward(st, demb):
    """a tower backward that ran on the side stream of CLIP.side_leg(): its parameter gradients went straight into the flat gradient
    buffer (no AccumulateGrad node the autograd engine could synchronise on), so the stream the leg forked from -- where the optimizer,
    the gradient reducer's finish() and the next step's zero_grad run -- waits for it here.  Autograd runs the later-built image tower
    backward first, so at this point the whole image backward is already enqueued on that stream and nothing of it is delayed."""
    main = st.get("join_to")
    if main is None:
        return
    cur = torch.cuda.current_stream(demb.device)
    if cur != main:
        demb.record_stream(cur)
        main.wait_stream(cur)


class _TowerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, which, inp, anchor):
        cfg = model.cfg
        need_grad = bool(ctx.needs_input_grad[3])
        dev = inp.device
        E = cfg["embed_dim"]
        M = inp.shape[0]
        ctx.model, ctx.which, ctx.M = model, which, M
        if M == 0:
            return torch.zeros(0, E, device=dev)
        fl = model._flat

        def p32(name):
            o = fl["off"][name]
            return fl["p32"][o:o + math.prod(fl["shapes"][name])].view(fl["shapes"][name])

        if model.precision == "fp32":
            if need_grad:
                raise RuntimeError("precision='fp32' is a forward-only path (embedding extraction / parity); run it under "
                                   "torch.no_grad() or switch back to precision='bf16' for training")
            return _encode_fp32(model, which, inp, p32)
        half = model.precision == "fp16"
        if half:
            if need_grad:
                raise RuntimeError("precision='fp16' is the embedder's forward-only path; run it under torch.no_grad() or switch "
                                   "back to precision='bf16' for training")
            model._sync_half_shadow()
        elif model.precision != "bf16":
            raise RuntimeError(f"unknown precision {model.precision!r}")
        if not _PY_TOWERS or half:
            # the whole tower in one C call (csrc/tower.hip); the workspace is the activation stash of the backward
            lib = _lib.load()
            desc = model.tower_desc(which, half=half)
            desc.pool_last_block = int(bool(model.pool_last_block))
            emb = torch.empty(M, E, device=dev, dtype=torch.float32)

            def ws_bytes(stash):
                desc.stash_act = int(stash)
                if which == "text" and model.pack_text:
                    return lib.uniir_clip_tower_workspace_bytes_packed(C.byref(desc), M, text_row_offsets(inp)[1], int(need_grad))
                return lib.uniir_clip_tower_workspace_bytes(C.byref(desc), M, int(need_grad))

            stash = False
            auto = False
            if need_grad and model.stash_act is not False:
                stash = True
                if model.stash_act is None:        # automatic: decided once per (tower, batch), see CLIP.__init__
                    auto = True
                    key = which
                    if key not in model._stash_choice:
                        free, _ = torch.cuda.mem_get_info(dev)
                        avail = free + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
                        need_stash = ws_bytes(True)
                        model._stash_choice[key] = need_stash + model.stash_margin_bytes <= avail
                        model.stash_log.append(f"{which} tower, {M} items: act(f) stash {'ON' if model._stash_choice[key] else 'off'} "
                                               f"(workspace {need_stash / 2**30:.1f} GiB, {avail / 2**30:.1f} GiB available, margin "
                                               f"{model.stash_margin_bytes / 2**30:.0f} GiB)")
                    stash = model._stash_choice[key]
            desc.stash_act = int(stash)
            model.last_stash_act[which] = bool(stash)

            def alloc_ws(nbytes_of):
                """the tower workspace; when the automatic act(f) stash does not fit after all, re-plan this tower without it"""
                nonlocal stash
                try:
                    n = nbytes_of()
                    return n, (torch.empty(n, device=dev, dtype=torch.uint8) if n >= 0 else None)
                except torch.OutOfMemoryError:
                    if not (auto and stash):
                        raise
                    torch.cuda.empty_cache()
                    stash = False
                    model._stash_choice[which] = False
                    model.last_stash_act[which] = False
                    desc.stash_act = 0
                    model.stash_log.append(f"{which} tower, {M} items: act(f) stash dropped after an out-of-memory error on its allocation")
                    n = nbytes_of()
                    return n, (torch.empty(n, device=dev, dtype=torch.uint8) if n >= 0 else None)

            if which == "text" and model.pack_text:
                # exact packing: only the tokens up to each caption's EOT are rows of the text tower (text_row_offsets)
                row_off, live = text_row_offsets(inp)
                need, ws = alloc_ws(lambda: lib.uniir_clip_tower_workspace_bytes_packed(C.byref(desc), M, live, int(need_grad)))
                if need < 0:
                    raise RuntimeError("uniir_clip_tower: unsupported tower geometry")
                _lib.check(lib.uniir_clip_tower_fwd_packed(C.byref(desc), inp.data_ptr(), M, row_off.data_ptr(), live, emb.data_ptr(),
                                                           ws.data_ptr(), need, int(need_grad), ops._stream()), "clip_tower_fwd_packed")
                model.last_text_rows = (live, M * cfg["context_length"])
                if need_grad:
                    ctx.stash = dict(ws=ws, inp=inp, ctower=True, row_off=row_off, live=live, stash_act=int(stash), join_to=model._leg_main,
                                     pool_last=int(desc.pool_last_block))
                return emb
            need, ws = alloc_ws(lambda: lib.uniir_clip_tower_workspace_bytes(C.byref(desc), M, int(need_grad)))
            if need < 0:
                raise RuntimeError("uniir_clip_tower: unsupported tower geometry")
            inp = inp.float().contiguous() if which == "image" else inp
            _lib.check(lib.uniir_clip_tower_fwd(C.byref(desc), inp.data_ptr(), M, emb.data_ptr(), ws.data_ptr(), need,
                                                int(need_grad), ops._stream()), "clip_tower_fwd")
            if need_grad:
                ctx.stash = dict(ws=ws, inp=inp, ctower=True, stash_act=int(stash), join_to=model._leg_main,
                                 pool_last=int(desc.pool_last_block))
            return emb
        if which == "image":
            W, P, L = cfg["vision_width"], cfg["vision_patch_size"], cfg["vision_layers"]
            res = cfg["image_resolution"]
            G = (res // P) ** 2
            T = G + 1
            heads = W // 64
            patches = torch.empty(M * G, model.kpad, device=dev, dtype=torch.bfloat16)
            ops.call("uniir_patchify", inp.float(), patches, M, res, P, model.kpad)
            po = ops.linear_fwd(patches, model._conv16)
            x0 = torch.empty(M * T, W, device=dev, dtype=torch.float32)
            ops.call("uniir_vit_assemble", po, p32("visual.class_embedding"), p32("visual.positional_embedding"), x0,
                     M, T, W)
            del po
            x = torch.empty(M * T, W, device=dev, dtype=torch.float32)
            ops.layernorm_fwd(x0, p32("visual.ln_pre.weight"), p32("visual.ln_pre.bias"), out_f32=x, rows=M * T, width=W)
            x, saved = _tower_fwd(model, "visual.transformer", L, x, M, T, W, heads, False, need_grad)
            rows = torch.empty(M, W, device=dev, dtype=torch.float32)
            ops.call("uniir_gather_rows", x, None, rows, M, T, W)
            pooled = ops.layernorm_fwd(rows, p32("visual.ln_post.weight"), p32("visual.ln_post.bias"), rows=M, width=W)
            emb = torch.empty(M, E, device=dev, dtype=torch.float32)
            ops.gemm(pooled, model.w16("visual.proj"), emb, M, E, W, W, E, E, b_tmaj=True, epilogue=ops.EPI_F32)
            if need_grad:
                ctx.stash = dict(patches=patches, x0=x0, saved=saved, rows=rows, pooled=pooled, T=T, W=W, L=L,
                                 heads=heads, idx=None)
            return emb
        # text tower
        W, L, ctxlen = cfg["transformer_width"], cfg["transformer_layers"], cfg["context_length"]
        heads, T = cfg["transformer_heads"], ctxlen
        x = torch.empty(M * T, W, device=dev, dtype=torch.float32)
        eot = torch.empty(M, device=dev, dtype=torch.int32)
        ops.call("uniir_text_embed", inp, p32("token_embedding.weight"), p32("positional_embedding"), x, eot, M, T, W,
                 cfg["vocab_size"])
        x, saved = _tower_fwd(model, "transformer", L, x, M, T, W, heads, True, need_grad)
        rows = torch.empty(M, W, device=dev, dtype=torch.float32)
        ops.call("uniir_gather_rows", x, eot, rows, M, T, W)
        pooled = ops.layernorm_fwd(rows, p32("ln_final.weight"), p32("ln_final.bias"), rows=M, width=W)
        emb = torch.empty(M, E, device=dev, dtype=torch.float32)
        ops.gemm(pooled, model.w16("text_projection"), emb, M, E, W, W, E, E, b_tmaj=True, epilogue=ops.EPI_F32)
        if need_grad:
            ctx.stash = dict(text=inp, saved=saved, rows=rows, pooled=pooled, T=T, W=W, L=L, heads=heads, idx=eot)
        return emb

    @staticmethod
    def backward(ctx, demb):
        model, which, M = ctx.model, ctx.which, ctx.M
        if M == 0:
            return None, None, None, None
        st = ctx.stash
        ctx.stash = None
        if st.get("ctower"):
            lib = _lib.load()
            desc = model.tower_desc(which)
            desc.stash_act = st["stash_act"]        # part of the workspace layout: the value the forward ran with
            desc.pool_last_block = st["pool_last"]
            ws, inp, stream = st["ws"], st["inp"], ops._stream()
            demb = demb.contiguous().float()
            need = ws.numel()
            reducer = getattr(model, "_grad_reducer", None)
            L = desc.layers
            prefix = "visual.transformer" if which == "image" else "transformer"
            if "row_off" in st:         # the text tower on packed rows
                ro, live = st["row_off"].data_ptr(), st["live"]
                _lib.check(lib.uniir_clip_tower_bwd_head_packed(C.byref(desc), demb.data_ptr(), M, ro, live, ws.data_ptr(), need, stream),
                           "tower_bwd_head_packed")
                for lo, hi in ([(0, L)] if reducer is None else [(i, i + 1) for i in reversed(range(L))]):
                    _lib.check(lib.uniir_clip_tower_bwd_blocks_packed(C.byref(desc), M, ro, live, lo, hi, ws.data_ptr(), need, stream),
                               "tower_bwd_blocks_packed")
                    if reducer is not None:
                        reducer.ready(*model.layer_grad_range(prefix, lo))
                _lib.check(lib.uniir_clip_tower_bwd_stem_packed(C.byref(desc), inp.data_ptr(), M, ro, live, ws.data_ptr(), need, stream),
                           "tower_bwd_stem_packed")
                _join_backward(st, demb)
                return None, None, None, None
            _lib.check(lib.uniir_clip_tower_bwd_head(C.byref(desc), demb.data_ptr(), M, ws.data_ptr(), need, stream), "tower_bwd_head")
            if reducer is None:
                _lib.check(lib.uniir_clip_tower_bwd_blocks(C.byref(desc), M, 0, L, ws.data_ptr(), need, stream), "tower_bwd_blocks")
            else:       # DDP overlap: hand every finished block's weight gradients to the collective stream
                for i in reversed(range(L)):
                    _lib.check(lib.uniir_clip_tower_bwd_blocks(C.byref(desc), M, i, i + 1, ws.data_ptr(), need, stream),
                               "tower_bwd_blocks")
                    reducer.ready(*model.layer_grad_range(prefix, i))
            _lib.check(lib.uniir_clip_tower_bwd_stem(C.byref(desc), inp.data_ptr(), M, ws.data_ptr(), need, stream), "tower_bwd_stem")
            _join_backward(st, demb)
            return None, None, None, None
        cfg = model.cfg
        fl = model._flat
        dev = demb.device
        E = cfg["embed_dim"]
        T, W, L, heads = st["T"], st["W"], st["L"], st["heads"]
        R = M * T

        def p32(name):
            o = fl["off"][name]
            return fl["p32"][o:o + math.prod(fl["shapes"][name])].view(fl["shapes"][name])

        image = which == "image"
        proj_n = "visual.proj" if image else "text_projection"
        lnw, lnb = (("visual.ln_post.weight", "visual.ln_post.bias") if image else ("ln_final.weight", "ln_final.bias"))
        prefix = "visual.transformer" if image else "transformer"
        demb16 = demb.contiguous().to(torch.bfloat16)
        # dproj[W,E] += pooled^T @ demb ; dpooled[M,W] = demb @ proj^T
        ops.gemm(st["pooled"], demb16, model.grad_view(proj_n), W, E, M, W, E, E, a_tmaj=True, b_tmaj=True,
                 epilogue=ops.EPI_ATOMIC_F32)
        dpooled = torch.empty(M, W, device=dev, dtype=torch.bfloat16)
        ops.gemm(demb16, model.w16(proj_n), dpooled, M, W, E, E, E, W)
        drows = ops.layernorm_bwd(st["rows"], p32(lnw), dpooled, model.grad_view(lnw), model.grad_view(lnb), rows=M,
                                  width=W)
        dx = torch.zeros(R, W, device=dev, dtype=torch.float32)
        ops.call("uniir_scatter_rows", drows, st["idx"], dx, M, T, W)
        dxb = torch.empty(R, W, device=dev, dtype=torch.bfloat16)
        ops.call("uniir_cast_f32_to_bf16", dx, dxb, dx.numel())
        dx = _tower_bwd(model, prefix, L, dx, dxb, st["saved"], M, T, W, heads, not image)
        if image:
            P = cfg["vision_patch_size"]
            G = T - 1
            dx0 = ops.layernorm_bwd(st["x0"], p32("visual.ln_pre.weight"), dx, model.grad_view("visual.ln_pre.weight"),
                                    model.grad_view("visual.ln_pre.bias"), rows=R, width=W)
            dpo = torch.empty(M * G, W, device=dev, dtype=torch.bfloat16)
            ops.call("uniir_vit_assemble_bwd", dx0, dpo, model.grad_view("visual.class_embedding"),
                     model.grad_view("visual.positional_embedding"), M, T, W)
            model._dconv.zero_()
            ops.linear_wgrad(dpo, st["patches"], model._dconv)
            ops.call("uniir_unpad_add", model._dconv, model.grad_view("visual.conv1.weight"), W, 3 * P * P, model.kpad)
        else:
            ops.call("uniir_text_embed_bwd", st["text"], dx, model.grad_view("token_embedding.weight"),
                     model.grad_view("positional_embedding"), M, T, W, cfg["vocab_size"])
        return None, None, None, None
