"""CPU ORACLE (test infrastructure, NOT the product path) -- plain PyTorch fp32 restatement of the hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; it is the checker,
never the thing measured or shipped.

What is restated, and from where (paths relative to the UniIR reference tree):
  * CLIP towers: openai/CLIP `clip/model.py` (git HEAD, un-vendored and unpinned in the reference:
    src/models/uniir_env.yml:23).  Call sites: src/models/uniir_clip/clip_scorefusion/clip_sf.py:25-26,44,47,66.
    The published algorithm is restated below (VisionTransformer.forward, CLIP.encode_text,
    ResidualAttentionBlock, QuickGELU, LayerNorm-in-fp32, build_attention_mask).  PARITY PINNING: the reference
    holds no test of the encoders; this restatement is pinned against an independent implementation of the same
    architecture (transformers.CLIPModel, tests/golden/make_golden.py -> tests/golden/g5_*.npz).
  * encode_multimodal_input / fuse:    clip_sf.py:53-63
  * in-batch contrastive loss:         clip_sf.py:68-147 (incl. the hard-negative branch :105-131)
    pinned against the reference class itself, imported with a stub `clip` module (tests/golden g1/g2/g3/g4).
  * optimizer groups + AdamW + cosine: clip_scorefusion/train.py:52-61,195-199,281-284; uniir_clip/engine.py:19-50
    pinned by g10 (the reference's own train_one_epoch run on CPU).

Rounding mode (`with rounding("bf16"):`, round 4).  The reference trains under autocast (engine.py:24-42: half-precision matmuls,
fp32 master weights and gradients); the device build does the same in bf16.  Inside the context the SAME restatement rounds to bf16
at the points where the device holds a 16-bit tensor -- GEMM operands (bf16 weight shadows, LayerNorm outputs, the packed q|k|v,
attention probabilities and outputs, the MLP pre-activation and activation, the patch embedding) and the 16-bit gradient copies the
backward GEMMs consume (gradients of those tensors, and the copy of the fp32 residual gradient that feeds each dgrad / wgrad) --
while the residual stream, LayerNorm / softmax statistics, accumulations and master gradients stay fp32.  It exists so that the
device gradients can be gated at ~1e-2 instead of the 1e-1 an fp32-everywhere oracle forces: what remains between the two is
summation order and 1-ulp intrinsics, so a parameter that stays above the gate is a bug, not noise.  Outside the context nothing
changes (fp32 everywhere, the pinned restatement).
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

CLIP_CONFIGS = {
    # geometry of the published checkpoints
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


def tiny_config(**kw):
    cfg = dict(embed_dim=64, image_resolution=64, vision_layers=2, vision_width=128, vision_patch_size=16,
               context_length=77, vocab_size=512, transformer_width=64, transformer_heads=1, transformer_layers=2)
    cfg.update(kw)
    return cfg


# ------------------------------------------------------------------------------------------------------------
# parameter construction (names = the checkpoint keys of SURVEY.md section 3.2, without the "clip_model." prefix)
# ------------------------------------------------------------------------------------------------------------
def init_state_dict(cfg, seed=0, dtype=torch.float32):
    """Random init following upstream CLIP.initialize_parameters / VisionTransformer.__init__ std choices."""
    g = torch.Generator().manual_seed(seed)

    def randn(*shape, std=1.0):
        return (torch.randn(*shape, generator=g) * std).to(dtype)

    sd = OrderedDict()
    vw, tw, E = cfg["vision_width"], cfg["transformer_width"], cfg["embed_dim"]
    P = cfg["vision_patch_size"]
    grid = cfg["image_resolution"] // P
    scale = vw ** -0.5
    fan_in = 3 * P * P
    sd["visual.conv1.weight"] = randn(vw, 3, P, P, std=(1.0 / fan_in) ** 0.5)
    sd["visual.class_embedding"] = randn(vw, std=scale)
    sd["visual.positional_embedding"] = randn(grid * grid + 1, vw, std=scale)
    sd["visual.ln_pre.weight"] = torch.ones(vw, dtype=dtype)
    sd["visual.ln_pre.bias"] = torch.zeros(vw, dtype=dtype)

    def blocks(prefix, width, layers):
        proj_std = (width ** -0.5) * ((2 * layers) ** -0.5)
        attn_std = width ** -0.5
        fc_std = (2 * width) ** -0.5
        for i in range(layers):
            p = f"{prefix}.resblocks.{i}"
            sd[f"{p}.attn.in_proj_weight"] = randn(3 * width, width, std=attn_std)
            sd[f"{p}.attn.in_proj_bias"] = randn(3 * width, std=0.01)
            sd[f"{p}.attn.out_proj.weight"] = randn(width, width, std=proj_std)
            sd[f"{p}.attn.out_proj.bias"] = randn(width, std=0.01)
            sd[f"{p}.ln_1.weight"] = 1.0 + randn(width, std=0.02)
            sd[f"{p}.ln_1.bias"] = randn(width, std=0.02)
            sd[f"{p}.mlp.c_fc.weight"] = randn(4 * width, width, std=fc_std)
            sd[f"{p}.mlp.c_fc.bias"] = randn(4 * width, std=0.01)
            sd[f"{p}.mlp.c_proj.weight"] = randn(width, 4 * width, std=proj_std)
            sd[f"{p}.mlp.c_proj.bias"] = randn(width, std=0.01)
            sd[f"{p}.ln_2.weight"] = 1.0 + randn(width, std=0.02)
            sd[f"{p}.ln_2.bias"] = randn(width, std=0.02)

    blocks("visual.transformer", vw, cfg["vision_layers"])
    sd["visual.ln_post.weight"] = torch.ones(vw, dtype=dtype)
    sd["visual.ln_post.bias"] = torch.zeros(vw, dtype=dtype)
    sd["visual.proj"] = randn(vw, E, std=scale)
    blocks("transformer", tw, cfg["transformer_layers"])
    sd["token_embedding.weight"] = randn(cfg["vocab_size"], tw, std=0.02)
    sd["positional_embedding"] = randn(cfg["context_length"], tw, std=0.01)
    sd["ln_final.weight"] = torch.ones(tw, dtype=dtype)
    sd["ln_final.bias"] = torch.zeros(tw, dtype=dtype)
    sd["text_projection"] = randn(tw, E, std=tw ** -0.5)
    sd["logit_scale"] = torch.tensor(math.log(1 / 0.07), dtype=dtype)
    return sd


# ------------------------------------------------------------------------------------------------------------
# bf16 rounding points of the device build (see the module docstring); identities outside `with rounding("bf16")`
# ------------------------------------------------------------------------------------------------------------
_ROUND = None


class rounding:
    def __init__(self, mode):
        if mode not in (None, "bf16"):
            raise ValueError(mode)
        self.mode = mode

    def __enter__(self):
        global _ROUND
        self.prev, _ROUND = _ROUND, self.mode

    def __exit__(self, *exc):
        global _ROUND
        _ROUND = self.prev
        return False


def _r16(x):
    return x.bfloat16().float()


class _Store16(torch.autograd.Function):
    """a tensor the device holds in bf16: its value and the gradient that comes back for it are both 16-bit"""
    @staticmethod
    def forward(ctx, x):
        return _r16(x)

    @staticmethod
    def backward(ctx, g):
        return _r16(g)


class _Weight16(torch.autograd.Function):
    """the bf16 shadow of an fp32 master weight (GEMM operand); its gradient is accumulated in fp32"""
    @staticmethod
    def forward(ctx, w):
        return _r16(w)

    @staticmethod
    def backward(ctx, g):
        return g


class _Grad16(torch.autograd.Function):
    """an fp32 tensor whose incoming gradient reaches the backward GEMMs through a bf16 copy (residual branches, the embedding)"""
    @staticmethod
    def forward(ctx, x):
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        return _r16(g)


def q16(x):
    return _Store16.apply(x) if _ROUND else x


def w16(w):
    return _Weight16.apply(w) if _ROUND else w


def g16(x):
    return _Grad16.apply(x) if _ROUND else x


class _Attn16(torch.autograd.Function):
    """softmax(q k^T / sqrt(d) [+ causal]) v on a packed bf16 q|k|v as the device kernels compute it (csrc/attention.hip): fp32
    logits and statistics, probabilities rounded to bf16 as matrix operands, output rounded to bf16; backward with P recomputed
    from the stored log-sum-exp, D = rowsum(dO * O), dS = P (dP - D) rounded to bf16, dq / dk / dv rounded to bf16"""
    @staticmethod
    def forward(ctx, qkv, heads, causal):
        N, L, W3 = qkv.shape
        W = W3 // 3
        hd = W // heads
        q, k, v = (t.view(N, L, heads, hd).transpose(1, 2) for t in qkv.split(W, dim=-1))
        s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
        if causal:
            s = s + torch.full((L, L), float("-inf")).triu_(1)
        m = s.max(dim=-1, keepdim=True).values
        pu = torch.exp(s - m)
        l = pu.sum(dim=-1, keepdim=True)
        o = _r16((_r16(pu) @ v) / l)
        ctx.save_for_backward(q, k, v, o, m + torch.log(l))
        ctx.geom = (N, L, W, heads, hd, causal)
        return o.transpose(1, 2).reshape(N, L, W)

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        N, L, W, heads, hd, causal = ctx.geom
        do = do.view(N, L, heads, hd).transpose(1, 2)                 # arrives bf16-valued (the out-projection's dgrad output)
        s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
        if causal:
            s = s + torch.full((L, L), float("-inf")).triu_(1)
        p = torch.exp(s - lse)
        D = (do * o).sum(dim=-1, keepdim=True)
        ds = _r16(p * (do @ v.transpose(-1, -2) - D))
        dv = _r16(_r16(p).transpose(-1, -2) @ do)
        dq = _r16((ds @ k) / math.sqrt(hd))
        dk = _r16((ds.transpose(-1, -2) @ q) / math.sqrt(hd))
        dqkv = torch.cat([t.transpose(1, 2).reshape(N, L, W) for t in (dq, dk, dv)], dim=-1)
        return dqkv, None, None


# ------------------------------------------------------------------------------------------------------------
# towers
# ------------------------------------------------------------------------------------------------------------
def layer_norm(x, w, b, eps=1e-5):
    # upstream LayerNorm subclass: compute in fp32, cast back
    return F.layer_norm(x.float(), (x.shape[-1],), w.float(), b.float(), eps).to(x.dtype)


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def attention(x, sd, p, heads, attn_mask=None):
    """nn.MultiheadAttention(x, x, x, need_weights=False, attn_mask=mask) on [N, L, W] (batch first here)."""
    N, L, W = x.shape
    hd = W // heads
    if _ROUND:       # x is the bf16 LayerNorm output; packed q|k|v in bf16; the device attention; fp32 out-projection result
        qkv = q16(x @ w16(sd[f"{p}.attn.in_proj_weight"]).t() + sd[f"{p}.attn.in_proj_bias"])
        o = _Attn16.apply(qkv, heads, attn_mask is not None)
        return o @ w16(sd[f"{p}.attn.out_proj.weight"]).t() + sd[f"{p}.attn.out_proj.bias"]
    qkv = x @ sd[f"{p}.attn.in_proj_weight"].t() + sd[f"{p}.attn.in_proj_bias"]
    q, k, v = qkv.split(W, dim=-1)
    q = q.view(N, L, heads, hd).transpose(1, 2)
    k = k.view(N, L, heads, hd).transpose(1, 2)
    v = v.view(N, L, heads, hd).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    if attn_mask is not None:
        s = s + attn_mask
    a = torch.softmax(s, dim=-1)
    o = (a @ v).transpose(1, 2).reshape(N, L, W)
    return o @ sd[f"{p}.attn.out_proj.weight"].t() + sd[f"{p}.attn.out_proj.bias"]


def resblock(x, sd, p, heads, attn_mask=None):
    # q16 / w16 / g16: identities unless `with rounding("bf16")` (the device's 16-bit tensors, see the module docstring)
    x = x + g16(attention(q16(layer_norm(x, sd[f"{p}.ln_1.weight"], sd[f"{p}.ln_1.bias"])), sd, p, heads, attn_mask))
    h = q16(layer_norm(x, sd[f"{p}.ln_2.weight"], sd[f"{p}.ln_2.bias"]))
    f = q16(h @ w16(sd[f"{p}.mlp.c_fc.weight"]).t() + sd[f"{p}.mlp.c_fc.bias"])      # the stashed pre-activation (bf16 on the device)
    h = q16(quick_gelu(f))
    return x + g16(h @ w16(sd[f"{p}.mlp.c_proj.weight"]).t() + sd[f"{p}.mlp.c_proj.bias"])


def encode_image(sd, image, cfg, return_tokens=False):
    """VisionTransformer.forward: conv1 -> [cls; patches] + pos -> ln_pre -> blocks -> ln_post(x[:,0]) @ proj."""
    vw, P = cfg["vision_width"], cfg["vision_patch_size"]
    if _ROUND:
        image = _r16(image)                                           # bf16 patches
    x = q16(F.conv2d(image, w16(sd["visual.conv1.weight"]), stride=P))   # [N, W, g, g]; bf16 on the device
    x = x.reshape(x.shape[0], vw, -1).permute(0, 2, 1)                # [N, g*g, W]
    cls = sd["visual.class_embedding"].to(x.dtype) + torch.zeros(x.shape[0], 1, vw, dtype=x.dtype)
    x = torch.cat([cls, x], dim=1) + sd["visual.positional_embedding"]
    x = layer_norm(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])
    heads = vw // 64
    for i in range(cfg["vision_layers"]):
        x = resblock(x, sd, f"visual.transformer.resblocks.{i}", heads)
    if return_tokens:
        return x
    x = q16(layer_norm(x[:, 0, :], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"]))
    return g16(x @ w16(sd["visual.proj"]))


def build_attention_mask(L):
    return torch.full((L, L), float("-inf")).triu_(1)


def encode_text(sd, text, cfg):
    """CLIP.encode_text: token_embedding + pos -> causal blocks -> ln_final -> row at argmax(text) @ text_projection."""
    x = sd["token_embedding.weight"][text.long()] + sd["positional_embedding"]
    mask = build_attention_mask(cfg["context_length"]).to(x.dtype)
    for i in range(cfg["transformer_layers"]):
        x = resblock(x, sd, f"transformer.resblocks.{i}", cfg["transformer_heads"], mask)
    x = layer_norm(x, sd["ln_final.weight"], sd["ln_final.bias"])
    x = q16(x[torch.arange(x.shape[0]), text.argmax(dim=-1)])
    return g16(x @ w16(sd["text_projection"]))


# ------------------------------------------------------------------------------------------------------------
# UniIR-owned logic
# ------------------------------------------------------------------------------------------------------------
def encode_multimodal_input(sd, cfg, txt, img, txt_mask, img_mask):
    """clip_sf.py:53-63: text_emb * txt_mask[:,None] + image_emb * img_mask[:,None] (both towers always run)."""
    txt_emb = encode_text(sd, txt, cfg) * txt_mask.unsqueeze(-1)
    img_emb = encode_image(sd, img, cfg) * img_mask.unsqueeze(-1)
    return img_emb + txt_emb  # fuse_embeddings(txt_emb, img_emb) = img_emb(arg) + txt_emb(arg): a commutative add


def inbatch_contrastive_loss(embeddings, index_mapping, logit_scale_exp, *, gather=None, rank=0,
                             in_batch_neg_num=0):
    """clip_sf.py:88-147.  `gather`: None (gather_embeddings False) or a callable p_embeds -> all_p_embeds
    (rank-major concat of every rank's p_embeds, differentiable)."""
    q = embeddings[torch.tensor(index_mapping["query"]).flatten()]
    p = embeddings[torch.tensor(index_mapping["pos_cand"]).flatten()]
    bs = q.size(0)
    q = F.normalize(q, dim=-1)
    p = F.normalize(p, dim=-1)
    if "neg_cand_list" in index_mapping:  # hard-negative branch, clip_sf.py:105-131
        n = F.normalize(embeddings[torch.tensor(index_mapping["neg_cand_list"])], dim=-1)
        nneg = min(bs - 1, in_batch_neg_num)
        mask = torch.eye(bs) == 0
        inb = p.unsqueeze(1).expand(-1, bs, -1)[mask].reshape(bs, bs - 1, -1)[:, :nneg, :]
        aug = torch.cat([n, inb], dim=1)
        pos = (q * p).sum(-1) * logit_scale_exp
        neg = (q.unsqueeze(1) * aug).sum(-1) * logit_scale_exp
        logits = torch.cat([pos.unsqueeze(-1), neg], 1)
        loss = torch.mean(-1.0 * F.log_softmax(logits, dim=1)[:, 0])
        acc = (logits.max(1)[1] == 0).sum() / bs
        return {"loss": loss, "accuracy": acc, "score": logits}
    if gather is not None:
        all_p = gather(p)
        score = torch.matmul(q, all_p.t()) * logit_scale_exp
        target = rank * bs + torch.arange(bs)
    else:
        score = torch.matmul(q, p.t()) * logit_scale_exp
        target = torch.arange(bs)
    loss = F.cross_entropy(score, target)
    acc = (score.max(1)[1] == target).sum() / bs
    return {"loss": loss, "accuracy": acc, "score": score}


def weight_decay_groups(named_parameters):
    """train.py:195-197: wd 0 for p.ndim < 2 or a name containing bn / ln / bias / logit_scale; wd 0.2 otherwise."""
    no_decay, decay = [], []
    for n, p in named_parameters:
        if not p.requires_grad:
            continue
        if p.ndim < 2 or any(s in n for s in ["bn", "ln", "bias", "logit_scale"]):
            no_decay.append((n, p))
        else:
            decay.append((n, p))
    return no_decay, decay


def cosine_lr(base_lr, step, t_total):
    """closed form of CosineAnnealingLR(T_max=t_total, eta_min=0) after `step` scheduler.step() calls."""
    return base_lr * (1 + math.cos(math.pi * step / t_total)) / 2


class OracleCLIP(torch.nn.Module):
    """nn.Module wrapper (parameters named like the checkpoint) so that autograd / torch.optim can drive the oracle
    and so that the reference's CLIPScoreFusion can be exercised with it through a stub `clip` module."""

    def __init__(self, cfg, sd=None, seed=0):
        super().__init__()
        self.cfg = cfg
        sd = sd if sd is not None else init_state_dict(cfg, seed)
        self._names = list(sd.keys())
        for k, v in sd.items():
            self.register_parameter(k.replace(".", "__"), torch.nn.Parameter(v.clone().float()))

    def sd(self):
        return {k: getattr(self, k.replace(".", "__")) for k in self._names}

    # `logit_scale` has no dot in its key, so the registered parameter is reachable as self.logit_scale directly

    def encode_image(self, image):
        return encode_image(self.sd(), image, self.cfg)

    def encode_text(self, text):
        return encode_text(self.sd(), text, self.cfg)


def synthetic_batch(cfg, pairs, seed=2023, device="cpu"):
    """SURVEY.md section 8(d) synthetic inputs: images randn, token rows [SOT, r_1..r_L, EOT, 0...], masks all 1,
    index_mapping as the collator builds it (mbeir_dataset.py:483-498: item 2i = query, 2i+1 = pos cand)."""
    g = torch.Generator().manual_seed(seed)
    M = 2 * pairs
    res, ctx, vocab = cfg["image_resolution"], cfg["context_length"], cfg["vocab_size"]
    img = torch.randn(M, 3, res, res, generator=g)
    txt = torch.zeros(M, ctx, dtype=torch.int32)
    sot, eot = vocab - 2, vocab - 1
    for i in range(M):
        L = int(torch.randint(5, min(61, ctx - 2), (1,), generator=g))
        txt[i, 0] = sot
        txt[i, 1:1 + L] = torch.randint(1, sot, (L,), generator=g, dtype=torch.int32)
        txt[i, 1 + L] = eot
    batch = {
        "txt_batched": txt.to(device),
        "image_batched": img.to(device),
        "txt_mask_batched": torch.ones(M, dtype=torch.int64, device=device),
        "image_mask_batched": torch.ones(M, dtype=torch.int64, device=device),
        "index_mapping": {"query": [[2 * i] for i in range(pairs)], "pos_cand": [[2 * i + 1] for i in range(pairs)]},
    }
    return batch
