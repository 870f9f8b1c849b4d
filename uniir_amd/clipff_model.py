"""MI355X-native CLIP feature-fusion encoder: the CLIP towers WITHOUT pooling
(src/models/uniir_clip/clip_featurefusion/clip_ff.py:35-59 vision: ln_post + proj on all tokens; :148-156 text: ln_final on
all tokens, no EOT pooling / projection), the token concat, the 2-layer T5 encoder stack (transformers T5Stack: RMS norm,
un-scaled attention with bucketed relative position bias, ReLU feed-forward, no biases) and the mean over tokens (:161-192).

All arithmetic is in libuniir_hip.so; this file is the launch sequence + the flat store of the T5 parameters (state-dict
keys `t5_layers.block.{i}.layer.{0,1}...` of transformers).  In train mode the T5 dropout sites (dropout_rate 0.1: stack
input, attention probabilities, attention output, after the ReLU, feed-forward output, after the final norm) are applied
with counter-based masks regenerated in backward (ops.DropSeeds); the CLIP towers have no dropout."""
import math

import torch

from . import ops
from .blip_model import FlatStore
from .clip_model import _tower_bwd, _tower_fwd

T5_EPS = 1e-6
T5_DROPOUT = 0.1      # transformers T5Config default dropout_rate (clip_ff.py:82,90 build T5Config() without overriding it)
T5_BUCKETS, T5_MAXDIST = 32, 128


def t5_param_shapes(d_model, heads, d_kv=64, d_ff=2048, layers=2):
    """flat order: q, k, v weights adjacent -> one [3 inner, d_model] GEMM"""
    inner = heads * d_kv
    out = []
    for i in range(layers):
        a, f = f"block.{i}.layer.0.", f"block.{i}.layer.1."
        out += [(a + "SelfAttention.q.weight", (inner, d_model)), (a + "SelfAttention.k.weight", (inner, d_model)),
                (a + "SelfAttention.v.weight", (inner, d_model)), (a + "SelfAttention.o.weight", (d_model, inner))]
        if i == 0:
            out += [(a + "SelfAttention.relative_attention_bias.weight", (T5_BUCKETS, heads))]
        out += [(a + "layer_norm.weight", (d_model,)), (f + "DenseReluDense.wi.weight", (d_ff, d_model)),
                (f + "DenseReluDense.wo.weight", (d_model, d_ff)), (f + "layer_norm.weight", (d_model,))]
    out += [("final_layer_norm.weight", (d_model,))]
    return out


def rel_bucket_table(T):
    """bucket of every offset key - query in [-(T-1), T-1] (T5Attention._relative_position_bucket, bidirectional);
    integer host logic, same float32 log as transformers"""
    rel = torch.arange(-(T - 1), T)
    nb = T5_BUCKETS // 2
    n = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(n.float().clamp_min(1) / max_exact) / math.log(T5_MAXDIST / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return ((rel > 0).long() * nb + torch.where(n < max_exact, n, large)).to(torch.int32)


_TABLES = {}


def _table(T, dev):
    key = (T, str(dev))
    if key not in _TABLES:
        _TABLES[key] = rel_bucket_table(T).to(dev)
    return _TABLES[key]


def _rms(st, x, name, R, D, out_f32=False):
    y16 = torch.empty(R, D, device=x.device, dtype=torch.bfloat16) if not out_f32 else None
    y32 = torch.empty(R, D, device=x.device, dtype=torch.float32) if out_f32 else None
    ops.call("uniir_rmsnorm_fwd", x, D, st.p(name), y16, y32, R, D, T5_EPS)
    return y32 if out_f32 else y16


def t5_forward(st, prefix, x, M, T, heads, layers, save, drop=None, p=T5_DROPOUT):
    """x fp32 [M*T, D] (the concatenated tokens; overwritten by its dropped version in train mode) -> pooled fp32 [M, D]
    (+ stash).  drop: ops.DropSeeds in train mode, None in eval mode"""
    R, D = x.shape
    inner = heads * 64
    dev = x.device
    table = _table(T, dev)
    rel = st.p(prefix + "block.0.layer.0.SelfAttention.relative_attention_bias.weight")
    saved = []
    p = p if drop is not None else 0.0
    nxt = (lambda: drop.next()) if p else (lambda: 0)
    s_in = nxt()
    if p:
        ops.dropout_f32(x, p, s_in, out_f32=x)

    def branch(x16, w, resid):
        """resid + dropout(x16 @ w^T)"""
        if not p:
            return ops.linear_fwd(x16, w, epilogue=ops.EPI_RESID_F32, resid=resid), 0
        t = ops.linear_fwd(x16, w, epilogue=ops.EPI_RESID_F32)
        sd = nxt()
        ops.dropout_f32(t, p, sd, resid=resid, out_f32=t)
        return t, sd

    for i in range(layers):
        a, f = f"{prefix}block.{i}.layer.0.", f"{prefix}block.{i}.layer.1."
        h1 = _rms(st, x, a + "layer_norm.weight", R, D)
        qkv = ops.linear_fwd(h1, st.w16(a + "SelfAttention.q.weight", (3 * inner, D)))
        ao = torch.empty(R, inner, device=dev, dtype=torch.bfloat16)
        lse = torch.empty(M, heads, T, device=dev, dtype=torch.float32)
        sa = nxt()
        ops.call("uniir_attention_rel_fwd", qkv, ao, lse, rel, table, T5_BUCKETS, 1.0, M, T, heads, p, sa)
        x2, so = branch(ao, st.w16(a + "SelfAttention.o.weight"), x)
        h2 = _rms(st, x2, f + "layer_norm.weight", R, D)
        wi = st.w16(f + "DenseReluDense.wi.weight")
        ff = torch.empty(R, wi.shape[0], device=dev, dtype=torch.bfloat16)
        g = torch.empty(R, wi.shape[0], device=dev, dtype=torch.bfloat16)
        ops.linear_fwd(h2, wi, out=ff, epilogue=ops.EPI_BIAS_ACT, C2=g, act=ops.ACT_RELU)
        sg = nxt()
        if p:
            ops.dropout_bf16_(g, p, sg)
        xn, sf = branch(g, st.w16(f + "DenseReluDense.wo.weight"), x2)
        if save:
            saved.append((x, h1, qkv, ao, lse, x2, h2, ff, (sa, so, sg, sf)))
        x = xn
    y = _rms(st, x, prefix + "final_layer_norm.weight", R, D, out_f32=True)
    s_fin = nxt()
    if p:
        ops.dropout_f32(y, p, s_fin, out_f32=y)
    pooled = torch.empty(M, D, device=dev, dtype=torch.float32)
    ops.call("uniir_meanpool_fwd", y, pooled, M, T, D)
    stash = dict(saved=saved, xf=x, M=M, T=T, p=p, s_in=s_in, s_fin=s_fin) if save else None
    return pooled, stash


def t5_backward(st, prefix, dpooled, stash, heads, layers):
    """returns d(concatenated tokens) fp32 [M*T, D]; parameter gradients accumulate into st.g32"""
    M, T = stash["M"], stash["T"]
    xf = stash["xf"]
    R, D = xf.shape
    inner = heads * 64
    dev = xf.device
    G = st.grad_view
    table = _table(T, dev)
    rel_name = prefix + "block.0.layer.0.SelfAttention.relative_attention_bias.weight"
    dy = torch.empty(R, D, device=dev, dtype=torch.float32)
    ops.call("uniir_meanpool_bwd", dpooled.contiguous(), dy, M, T, D)
    p = stash["p"]
    if p:
        ops.dropout_f32(dy, p, stash["s_fin"], out_f32=dy)
    dx = torch.empty(R, D, device=dev, dtype=torch.float32)
    dxb = torch.empty(R, D, device=dev, dtype=torch.bfloat16)
    ops.call("uniir_rmsnorm_bwd", xf, D, st.p(prefix + "final_layer_norm.weight"), dy, 1, None, dx, D, dxb,
             G(prefix + "final_layer_norm.weight"), R, D, T5_EPS)
    del dy
    for i in reversed(range(layers)):
        a, f = f"{prefix}block.{i}.layer.0.", f"{prefix}block.{i}.layer.1."
        x, h1, qkv, ao, lse, x2, h2, ff, (sa, so, sg, sf) = stash["saved"][i]
        stash["saved"][i] = None
        if p:       # the feed-forward branch sees the masked gradient; dx (fp32) keeps the residual one
            ops.dropout_bf16_(dxb, p, sf)
        wi, wo = st.w16(f + "DenseReluDense.wi.weight"), st.w16(f + "DenseReluDense.wo.weight")
        g = torch.empty_like(ff)
        df = torch.empty_like(ff)
        ops.linear_dgrad(dxb, wo, out=df, aux=ff, act_out=g, act=ops.ACT_RELU)        # df = (dx @ Wo) * relu'(ff), g = relu(ff)
        if p:       # the ReLU-output mask: on the re-materialised activation and (it commutes with relu') on its gradient
            ops.dropout_bf16_(g, p, sg)
            ops.dropout_bf16_(df, p, sg)
        ops.linear_wgrad(dxb, g, G(f + "DenseReluDense.wo.weight"))
        ops.linear_wgrad(df, h2, G(f + "DenseReluDense.wi.weight"))
        dh = ops.linear_dgrad(df, wi)
        dx2 = torch.empty(R, D, device=dev, dtype=torch.float32)
        ops.call("uniir_rmsnorm_bwd", x2, D, st.p(f + "layer_norm.weight"), dh, 0, dx, dx2, D, dxb, G(f + "layer_norm.weight"),
                 R, D, T5_EPS)
        del ff, g, df, h2, x2
        if p:
            ops.dropout_bf16_(dxb, p, so)
        ops.linear_wgrad(dxb, ao, G(a + "SelfAttention.o.weight"))
        dao = ops.linear_dgrad(dxb, st.w16(a + "SelfAttention.o.weight"))
        dqkv = torch.empty_like(qkv)
        ops.call("uniir_attention_rel_bwd", qkv, ao, dao, lse, dqkv, st.p(rel_name), table, T5_BUCKETS, 1.0, G(rel_name), M, T,
                 heads, p, sa)
        ops.linear_wgrad(dqkv, h1, G(a + "SelfAttention.q.weight", (3 * inner, D)))
        dh = ops.linear_dgrad(dqkv, st.w16(a + "SelfAttention.q.weight", (3 * inner, D)))
        ops.call("uniir_rmsnorm_bwd", x, D, st.p(a + "layer_norm.weight"), dh, 0, dx2, dx, D, dxb, G(a + "layer_norm.weight"),
                 R, D, T5_EPS)
        del qkv, ao, lse, h1, x, dx2
    if p:
        ops.dropout_f32(dx, p, stash["s_in"], out_f32=dx)
    return dx


# ------------------------------------------------------------------------------------------------------------
# towers without pooling
# ------------------------------------------------------------------------------------------------------------
def vision_tokens_fwd(model, image, save):
    """clip_ff.py:35-59 -> fp32 [M*T, E] = ln_post(all tokens) @ proj"""
    cfg, fl = model.cfg, model._flat
    dev = image.device
    M = image.shape[0]
    W, P, L, res, E = cfg["vision_width"], cfg["vision_patch_size"], cfg["vision_layers"], cfg["image_resolution"], cfg["embed_dim"]
    G_ = (res // P) ** 2
    T = G_ + 1
    heads = W // 64
    p32 = lambda n: fl["p32"][fl["off"][n]:fl["off"][n] + math.prod(fl["shapes"][n])].view(fl["shapes"][n])
    patches = torch.empty(M * G_, model.kpad, device=dev, dtype=torch.bfloat16)
    ops.call("uniir_patchify", image.float().contiguous(), patches, M, res, P, model.kpad)
    po = ops.linear_fwd(patches, model._conv16)
    x0 = torch.empty(M * T, W, device=dev, dtype=torch.float32)
    ops.call("uniir_vit_assemble", po, p32("visual.class_embedding"), p32("visual.positional_embedding"), x0, M, T, W)
    del po
    x = torch.empty(M * T, W, device=dev, dtype=torch.float32)
    ops.layernorm_fwd(x0, p32("visual.ln_pre.weight"), p32("visual.ln_pre.bias"), out_f32=x, rows=M * T, width=W)
    x, saved = _tower_fwd(model, "visual.transformer", L, x, M, T, W, heads, False, save)
    ln = ops.layernorm_fwd(x, p32("visual.ln_post.weight"), p32("visual.ln_post.bias"), rows=M * T, width=W)
    tok = torch.empty(M * T, E, device=dev, dtype=torch.float32)
    ops.gemm(ln, model.w16("visual.proj"), tok, M * T, E, W, W, E, E, b_tmaj=True, epilogue=ops.EPI_F32)
    stash = dict(patches=patches, x0=x0, saved=saved, xf=x, ln=ln, M=M, T=T) if save else None
    return tok, T, stash


def vision_tokens_bwd(model, dtok, stash):
    cfg, fl = model.cfg, model._flat
    M, T = stash["M"], stash["T"]
    W, P, L, E = cfg["vision_width"], cfg["vision_patch_size"], cfg["vision_layers"], cfg["embed_dim"]
    R, dev, heads = M * T, dtok.device, W // 64
    p32 = lambda n: fl["p32"][fl["off"][n]:fl["off"][n] + math.prod(fl["shapes"][n])].view(fl["shapes"][n])
    d16 = torch.empty(R, E, device=dev, dtype=torch.bfloat16)
    ops.call("uniir_cast_f32_to_bf16", dtok.contiguous(), d16, d16.numel())
    # dproj[W,E] += ln^T @ dtok ; dln[R,W] = dtok @ proj^T
    ops.gemm(stash["ln"], d16, model.grad_view("visual.proj"), W, E, R, W, E, E, a_tmaj=True, b_tmaj=True,
             epilogue=ops.EPI_ATOMIC_F32, k_splits=ops.wgrad_splits(R, ((W + 255) // 256) * ((E + 255) // 256)))
    dln = torch.empty(R, W, device=dev, dtype=torch.bfloat16)
    ops.gemm(d16, model.w16("visual.proj"), dln, R, W, E, E, E, W)
    dxb = torch.empty(R, W, device=dev, dtype=torch.bfloat16)
    dx = ops.layernorm_bwd(stash["xf"], p32("visual.ln_post.weight"), dln, model.grad_view("visual.ln_post.weight"),
                           model.grad_view("visual.ln_post.bias"), dx_bf16=dxb, rows=R, width=W)
    dx = _tower_bwd(model, "visual.transformer", L, dx, dxb, stash["saved"], M, T, W, heads, False)
    dx0 = ops.layernorm_bwd(stash["x0"], p32("visual.ln_pre.weight"), dx, model.grad_view("visual.ln_pre.weight"),
                            model.grad_view("visual.ln_pre.bias"), rows=R, width=W)
    dpo = torch.empty(M * (T - 1), W, device=dev, dtype=torch.bfloat16)
    ops.call("uniir_vit_assemble_bwd", dx0, dpo, model.grad_view("visual.class_embedding"),
             model.grad_view("visual.positional_embedding"), M, T, W)
    model._dconv.zero_()
    ops.linear_wgrad(dpo, stash["patches"], model._dconv)
    ops.call("uniir_unpad_add", model._dconv, model.grad_view("visual.conv1.weight"), W, 3 * P * P, model.kpad)


def text_tokens_fwd(model, text, save):
    """clip_ff.py:148-156 -> fp32 [M*77, W] = ln_final(all tokens)"""
    cfg, fl = model.cfg, model._flat
    dev = text.device
    M = text.shape[0]
    W, L, T, heads = cfg["transformer_width"], cfg["transformer_layers"], cfg["context_length"], cfg["transformer_heads"]
    p32 = lambda n: fl["p32"][fl["off"][n]:fl["off"][n] + math.prod(fl["shapes"][n])].view(fl["shapes"][n])
    x = torch.empty(M * T, W, device=dev, dtype=torch.float32)
    eot = torch.empty(M, device=dev, dtype=torch.int32)
    ops.call("uniir_text_embed", text, p32("token_embedding.weight"), p32("positional_embedding"), x, eot, M, T, W,
             cfg["vocab_size"])
    x, saved = _tower_fwd(model, "transformer", L, x, M, T, W, heads, True, save)
    tok = torch.empty(M * T, W, device=dev, dtype=torch.float32)
    ops.layernorm_fwd(x, p32("ln_final.weight"), p32("ln_final.bias"), out_f32=tok, rows=M * T, width=W)
    stash = dict(text=text, saved=saved, xf=x, M=M, T=T) if save else None
    return tok, T, stash


def text_tokens_bwd(model, dtok, stash):
    cfg, fl = model.cfg, model._flat
    M, T = stash["M"], stash["T"]
    W, L, heads = cfg["transformer_width"], cfg["transformer_layers"], cfg["transformer_heads"]
    R, dev = M * T, dtok.device
    p32 = lambda n: fl["p32"][fl["off"][n]:fl["off"][n] + math.prod(fl["shapes"][n])].view(fl["shapes"][n])
    dxb = torch.empty(R, W, device=dev, dtype=torch.bfloat16)
    dx = ops.layernorm_bwd(stash["xf"], p32("ln_final.weight"), dtok.contiguous(), model.grad_view("ln_final.weight"),
                           model.grad_view("ln_final.bias"), dx_bf16=dxb, rows=R, width=W)
    dx = _tower_bwd(model, "transformer", L, dx, dxb, stash["saved"], M, T, W, heads, True)
    ops.call("uniir_text_embed_bwd", stash["text"], dx, model.grad_view("token_embedding.weight"),
             model.grad_view("positional_embedding"), M, T, W, cfg["vocab_size"])


class FusionFn(torch.autograd.Function):
    """(clip module, T5 store, text int32 [M,77], image [M,3,H,W]) -> fused embedding fp32 [M, D]"""

    @staticmethod
    def forward(ctx, owner, text, image, anchor):
        save = bool(ctx.needs_input_grad[3])
        clip, st = owner.clip_model, owner._t5
        ttok, Tt, tst = text_tokens_fwd(clip, text, save)
        itok, Ti, ist = vision_tokens_fwd(clip, image, save)
        M, D = text.shape[0], ttok.shape[1]
        x = torch.cat([ttok.view(M, Tt, D), itok.view(M, Ti, D)], dim=1).view(M * (Tt + Ti), D)     # a copy, no arithmetic
        del ttok, itok
        drop = ops.DropSeeds() if owner.training and owner.t5_dropout > 0 else None
        pooled, fst = t5_forward(st, "", x, M, Tt + Ti, owner.t5_heads, owner.t5_layers_n, save, drop=drop,
                                 p=owner.t5_dropout)
        ctx.owner, ctx.stashes, ctx.dims = owner, (tst, ist, fst), (M, Tt, Ti, D)
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        owner = ctx.owner
        tst, ist, fst = ctx.stashes
        ctx.stashes = None
        M, Tt, Ti, D = ctx.dims
        dx = t5_backward(owner._t5, "", dpooled, fst, owner.t5_heads, owner.t5_layers_n).view(M, Tt + Ti, D)
        reducer = getattr(owner.clip_model, "_grad_reducer", None)
        if reducer is not None:        # the T5 gradients are final: their all-reduce overlaps the two tower backwards
            reducer.reduce_extra(owner._t5.g32)
        dt = dx[:, :Tt].contiguous().view(M * Tt, D)
        di = dx[:, Tt:].contiguous().view(M * Ti, D)
        del dx
        text_tokens_bwd(owner.clip_model, dt, tst)
        vision_tokens_bwd(owner.clip_model, di, ist)
        return None, None, None, None


def init_t5_store(owner, d_model, heads, device, seed=0, d_ff=2048, layers=2):
    """random init like transformers' T5 _init_weights scales (only used when no checkpoint is loaded)"""
    shapes = t5_param_shapes(d_model, heads, 64, d_ff, layers)
    st = FlatStore(shapes, device, True)
    g = torch.Generator().manual_seed(seed)
    for n, shp in shapes:
        if n.endswith("layer_norm.weight"):
            v = torch.ones(shp)
        elif "relative_attention_bias" in n:
            v = torch.randn(shp, generator=g) * d_model ** -0.5
        else:
            v = torch.randn(shp, generator=g) * shp[-1] ** -0.5
        st.p(n).copy_(v)
    st.refresh_shadow()
    return st, shapes
