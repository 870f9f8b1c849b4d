"""MI355X-native BLIP feature-fusion model: BLIP ViT + MED BERT with cross-attention,
momentum encoders, feature queues and the soft-target contrastive loss, behind the module surface of the reference's
src/models/uniir_blip/blip_featurefusion/blip_ff.py:19-310 (same constructor arguments, attribute names, state-dict
keys `visual_encoder.* / text_encoder.* / visual_encoder_m.* / text_encoder_m.* / temp / query_queue / cand_queue /
idx_queue / new_ptr_queue`, forward(batch, alpha, encode_mbeir_batch)).

All arithmetic is in libuniir_hip.so; this file is the launch sequence, the flat parameter stores and the activation
stash.  Differences from the reference, stated once:
  * bf16 MFMA GEMMs with fp32 accumulation / fp32 LayerNorm, softmax statistics, residual stream, loss (the reference
    runs fp16 autocast); no GradScaler is needed;
  * train mode applies the reference's dropout sites -- BERT hidden / attention-probability dropout 0.1 (med.py:84,
    175,198,212,341; online and momentum encoders alike, both are in train mode) and DropPath of ViT-large (vit.py:79-80,
    blip.py:249-254) -- with counter-based masks regenerated in backward (ops.DropSeeds); the masks are a different random
    stream than torch's, so only the distribution matches; eval mode (and a config with zero rates) is deterministic;
  * the BERT padding mask must be a prefix mask (tokenizer padding="max_length"), it travels as one key length per row;
  * hard negatives (blip_ff.py:127-131,159-170,233-246) are supported; like the reference their momentum features are
    used un-normalised and a host coin flip decides what is enqueued.
"""
import json
import math

import torch
from torch import nn

from . import _lib, comm, ops
from .clip_model import ALIGN, _Blk, _tower_bwd, _tower_fwd

VIT_CONFIGS = {   # src/models/uniir_blip/backbone/blip.py:229-255 (create_vit)
    "base": dict(patch_size=16, embed_dim=768, depth=12, num_heads=12, drop_path_rate=0.0),
    "large": dict(patch_size=16, embed_dim=1024, depth=24, num_heads=16, drop_path_rate=0.1),
}
MED_DEFAULT = dict(hidden_size=768, intermediate_size=3072, layer_norm_eps=1e-12, max_position_embeddings=512,
                   num_attention_heads=12, num_hidden_layers=12, vocab_size=30524, hidden_dropout_prob=0.1,
                   attention_probs_dropout_prob=0.1)   # backbone/configs/med_config.json
VIT_EPS = 1e-6


# ------------------------------------------------------------------------------------------------------------
# names
# ------------------------------------------------------------------------------------------------------------
def vit_param_shapes(cfg, img_size):
    D, P = cfg["embed_dim"], cfg["patch_size"]
    T = (img_size // P) ** 2 + 1
    out = [("cls_token", (1, 1, D)), ("pos_embed", (1, T, D)), ("patch_embed.proj.weight", (D, 3, P, P)),
           ("patch_embed.proj.bias", (D,))]
    for i in range(cfg["depth"]):
        b = f"blocks.{i}."
        out += [(b + "norm1.weight", (D,)), (b + "norm1.bias", (D,)), (b + "attn.qkv.weight", (3 * D, D)),
                (b + "attn.qkv.bias", (3 * D,)), (b + "attn.proj.weight", (D, D)), (b + "attn.proj.bias", (D,)),
                (b + "norm2.weight", (D,)), (b + "norm2.bias", (D,)), (b + "mlp.fc1.weight", (4 * D, D)),
                (b + "mlp.fc1.bias", (4 * D,)), (b + "mlp.fc2.weight", (D, 4 * D)), (b + "mlp.fc2.bias", (D,))]
    out += [("norm.weight", (D,)), ("norm.bias", (D,))]
    return out


def bert_param_shapes(cfg, enc_width, pooler=True):
    """flat order: query/key/value weights (and biases) adjacent, so one [3W,W] (self) / [2W,enc] (cross) GEMM serves them"""
    W, I = cfg["hidden_size"], cfg["intermediate_size"]
    out = [("embeddings.word_embeddings.weight", (cfg["vocab_size"], W)),
           ("embeddings.position_embeddings.weight", (cfg["max_position_embeddings"], W)),
           ("embeddings.LayerNorm.weight", (W,)), ("embeddings.LayerNorm.bias", (W,))]
    for i in range(cfg["num_hidden_layers"]):
        b = f"encoder.layer.{i}."
        s, c = b + "attention.self.", b + "crossattention.self."
        out += [(s + "query.weight", (W, W)), (s + "key.weight", (W, W)), (s + "value.weight", (W, W)),
                (s + "query.bias", (W,)), (s + "key.bias", (W,)), (s + "value.bias", (W,)),
                (b + "attention.output.dense.weight", (W, W)), (b + "attention.output.dense.bias", (W,)),
                (b + "attention.output.LayerNorm.weight", (W,)), (b + "attention.output.LayerNorm.bias", (W,)),
                (c + "query.weight", (W, W)), (c + "query.bias", (W,)),
                (c + "key.weight", (W, enc_width)), (c + "value.weight", (W, enc_width)),
                (c + "key.bias", (W,)), (c + "value.bias", (W,)),
                (b + "crossattention.output.dense.weight", (W, W)), (b + "crossattention.output.dense.bias", (W,)),
                (b + "crossattention.output.LayerNorm.weight", (W,)), (b + "crossattention.output.LayerNorm.bias", (W,)),
                (b + "intermediate.dense.weight", (I, W)), (b + "intermediate.dense.bias", (I,)),
                (b + "output.dense.weight", (W, I)), (b + "output.dense.bias", (W,)),
                (b + "output.LayerNorm.weight", (W,)), (b + "output.LayerNorm.bias", (W,))]
    if pooler:
        out += [("pooler.dense.weight", (W, W)), ("pooler.dense.bias", (W,))]
    return out


def _vit_blk_names(prefix, i):
    b = f"{prefix}blocks.{i}."
    return dict(wqkv=b + "attn.qkv.weight", bqkv=b + "attn.qkv.bias", wo=b + "attn.proj.weight", bo=b + "attn.proj.bias",
                ln1w=b + "norm1.weight", ln1b=b + "norm1.bias", wfc=b + "mlp.fc1.weight", bfc=b + "mlp.fc1.bias",
                wproj=b + "mlp.fc2.weight", bproj=b + "mlp.fc2.bias", ln2w=b + "norm2.weight", ln2b=b + "norm2.bias")


# ------------------------------------------------------------------------------------------------------------
# flat stores
# ------------------------------------------------------------------------------------------------------------
class FlatStore:
    """One flat fp32 buffer (+ optional flat gradient) + bf16 shadow for an ordered list of named tensors; every
    tensor starts on a 256-B boundary.  Offers the small interface `_Blk` / the tower launch sequences use."""

    def __init__(self, named_shapes, device, with_grad):
        self.off, self.shapes, cur = {}, {}, 0
        for n, shp in named_shapes:
            self.off[n], self.shapes[n] = cur, tuple(shp)
            cur += (math.prod(shp) + ALIGN - 1) // ALIGN * ALIGN
        self.total = cur
        self.p32 = torch.zeros(cur, device=device, dtype=torch.float32)
        self.g32 = torch.zeros(cur, device=device, dtype=torch.float32) if with_grad else None
        self.w16_buf = torch.empty(cur, device=device, dtype=torch.bfloat16)
        self._flat = dict(off=self.off, shapes=self.shapes, p32=self.p32)

    def _view(self, buf, name, shape=None, numel=None):
        shape = shape or self.shapes[name]
        o = self.off[name]
        return buf[o:o + (numel or math.prod(shape))].view(shape)

    def p(self, name, shape=None):
        return self._view(self.p32, name, shape)

    def w16(self, name, shape=None):
        return self._view(self.w16_buf, name, shape)

    def grad_view(self, name, shape=None):
        return self._view(self.g32, name, shape)

    def refresh_shadow(self):
        ops.call("uniir_cast_f32_to_bf16", self.p32, self.w16_buf, self.total)


def _attach(root, dotted, param):
    mod = root
    parts = dotted.split(".")
    for part in parts[:-1]:
        if part not in mod._modules:
            mod.add_module(part, nn.Module())
        mod = mod._modules[part]
    mod.register_parameter(parts[-1], param)


# ------------------------------------------------------------------------------------------------------------
# launch sequences
# ------------------------------------------------------------------------------------------------------------
def vit_forward(st, conv16, prefix, cfg, img_size, images, save, drop=None):
    """src/models/uniir_blip/backbone/vit.py:196-221 -> bf16 tokens [M*T, D] (after the final norm) + stash.
    drop (ops.DropSeeds, train mode): DropPath with rates linspace(0, drop_path_rate, depth) (vit.py:167), one Bernoulli
    draw per item and residual branch from torch's CPU generator"""
    D, P, depth = cfg["embed_dim"], cfg["patch_size"], cfg["depth"]
    M, dev = images.shape[0], images.device
    G = (img_size // P) ** 2
    T = G + 1
    heads = cfg["num_heads"]
    kpad = conv16.shape[1]
    patches = torch.empty(M * G, kpad, device=dev, dtype=torch.bfloat16)
    ops.call("uniir_patchify", images.float().contiguous(), patches, M, img_size, P, kpad)
    po = ops.linear_fwd(patches, conv16, st.p(prefix + "patch_embed.proj.bias"))
    x0 = torch.empty(M * T, D, device=dev, dtype=torch.float32)
    ops.call("uniir_vit_assemble", po, st.p(prefix + "cls_token"), st.p(prefix + "pos_embed"), x0, M, T, D)
    del po
    blk = lambda i: _Blk(st, None, _vit_blk_names(prefix, i))
    rowscale = None
    if drop is not None and cfg.get("drop_path_rate", 0.0) > 0 and depth > 1:
        keep = 1.0 - torch.linspace(0, cfg["drop_path_rate"], depth).view(depth, 1, 1)
        rowscale = (torch.floor(keep + torch.rand(depth, 2, M)) / keep).to(dev)
    x, saved = _tower_fwd(st, None, depth, x0, M, T, D, heads, False, save, eps=VIT_EPS, act=ops.ACT_GELU_ERF, blk=blk,
                          rowscale=rowscale)
    tok = ops.layernorm_fwd(x, st.p(prefix + "norm.weight"), st.p(prefix + "norm.bias"), VIT_EPS, rows=M * T, width=D)
    stash = dict(patches=patches, x=x, saved=saved, M=M, T=T, rowscale=rowscale) if save else None
    return tok, T, stash


def vit_backward(st, dconv, prefix, cfg, dtok, stash):
    """dtok fp32 [M*T, D]: gradient w.r.t. the normed tokens; parameter gradients accumulate into st.g32"""
    D, P, depth, heads = cfg["embed_dim"], cfg["patch_size"], cfg["depth"], cfg["num_heads"]
    M, T = stash["M"], stash["T"]
    R, dev = M * T, dtok.device
    dxb = torch.empty(R, D, device=dev, dtype=torch.bfloat16)
    dx = ops.layernorm_bwd(stash["x"], st.p(prefix + "norm.weight"), dtok, st.grad_view(prefix + "norm.weight"),
                           st.grad_view(prefix + "norm.bias"), VIT_EPS, dx_bf16=dxb, rows=R, width=D)
    blk = lambda i: _Blk(st, None, _vit_blk_names(prefix, i))
    dx = _tower_bwd(st, None, depth, dx, dxb, stash["saved"], M, T, D, heads, False, eps=VIT_EPS, act=ops.ACT_GELU_ERF,
                    blk=blk, rowscale=stash["rowscale"])
    G = T - 1
    dpo = torch.empty(M * G, D, device=dev, dtype=torch.bfloat16)
    ops.call("uniir_vit_assemble_bwd", dx, dpo, st.grad_view(prefix + "cls_token"), st.grad_view(prefix + "pos_embed"),
             M, T, D)
    ops.call("uniir_colsum_bf16", dpo, D, st.grad_view(prefix + "patch_embed.proj.bias"), M * G, D)
    dconv.zero_()
    ops.linear_wgrad(dpo, stash["patches"], dconv)
    ops.call("uniir_unpad_add", dconv, st.grad_view(prefix + "patch_embed.proj.weight"), D, 3 * P * P, dconv.shape[1])


def _ln2(st, x, wname, eps, R, W):
    """post-LN sublayer output in both precisions: fp32 for the next residual, bf16 for the next GEMM"""
    o32 = torch.empty(R, W, device=x.device, dtype=torch.float32)
    o16 = torch.empty(R, W, device=x.device, dtype=torch.bfloat16)
    ops.layernorm_fwd(x, st.p(wname + "weight"), st.p(wname + "bias"), eps, out_bf16=o16, out_f32=o32, rows=R, width=W)
    return o32, o16


class TextPack:
    """The rows of a padded [M, L] token batch up to each caption's valid length, packed back to back.  The
    reference runs BERT on all L positions and masks the padded KEYS with (1 - m) * -10000 (med.py:687-688: exp underflows to an
    exact 0 in fp32), and only token 0 reaches pooler_output (blip_ff.py:82-116) -- so the rows behind a caption's last valid token
    never influence a result and carry exactly zero gradient.  Running the 12 layers on the live rows only changes no live row's
    value (the attention kernels take per-item row ranges, uniir_attention_fwd_rows; dropout masks are drawn at the dense
    coordinates); at the bench's lengths U{5..100} of 100 it drops 47 % of BERT's rows.
    row_off int32 [M + 1] (device), row_map int32 [R] (device: the dense row m * L + t of every packed row), R (host)."""

    def __init__(self, lens_host, L, dev):
        lens = lens_host.to(torch.int64).flatten()
        M = lens.numel()
        off = torch.zeros(M + 1, dtype=torch.int64)
        off[1:] = torch.cumsum(lens, 0)
        self.R, self.M, self.L = int(off[-1]), M, int(L)
        first = torch.repeat_interleave(torch.arange(M, dtype=torch.int64) * L - off[:-1], lens)      # dense row - packed row
        row_map = first + torch.arange(self.R, dtype=torch.int64)
        self.row_off = off.to(torch.int32).to(dev, non_blocking=True)
        self.row_map = row_map.to(torch.int32).to(dev, non_blocking=True)

    @staticmethod
    def build(key_len, L, dev=None):
        """from the key lengths: a host tensor (dev = where the rows live), or the device tensor (one device -> host read; the
        prefix-mask check of _text_inputs has synchronised already).  None when packing is pointless (every caption full) or
        impossible (an empty caption)"""
        dev = key_len.device if dev is None else dev
        lens = key_len.detach().to("cpu")
        if lens.numel() == 0 or int(lens.min()) < 1 or int(lens.max()) > L or int(lens.sum()) == lens.numel() * L:
            return None
        return TextPack(lens, L, dev)


def bert_forward(st, prefix, cfg, ids, key_len, img16, Ti, save, cross=True, pool=True, drop=None, pack=None):
    """src/models/uniir_blip/backbone/med.py BertModel.forward(mode="multimodal") -> pooler_output fp32 [M,W] + stash.
    ids int32 [M,L]; key_len int32 [M]; img16 bf16 [M*Ti, enc_width] (image attention mask all ones, blip_ff.py:98,108).
    cross=False: mode "text" (the cross-attention sublayer is skipped, BLIP_SF); pool=False: add_pooling_layer=False, the
    class-token row of last_hidden_state is returned instead of the tanh pooler output.
    drop (ops.DropSeeds, train mode): hidden dropout after the embedding LayerNorm and after each of the three output
    dense layers (before the residual add), attention-probability dropout inside both attention kernels.
    pack (TextPack): run on the rows up to each caption's valid length only -- same pooled output bit for bit (train mode included),
    parameter gradients equal up to the order of the fp32 additions in the weight-gradient reductions."""
    W, I, eps = cfg["hidden_size"], cfg["intermediate_size"], cfg["layer_norm_eps"]
    heads, layers = cfg["num_attention_heads"], cfg["num_hidden_layers"]
    M, L = ids.shape
    R, dev = (pack.R if pack is not None else M * L), ids.device
    roff = pack.row_off if pack is not None else None
    rmap = pack.row_map if pack is not None else None
    e32 = torch.empty(R, W, device=dev, dtype=torch.float32)
    if pack is None:
        scratch = torch.empty(M, device=dev, dtype=torch.int32)
        ops.call("uniir_text_embed", ids, st.p(prefix + "embeddings.word_embeddings.weight"),
                 st.p(prefix + "embeddings.position_embeddings.weight"), e32, scratch, M, L, W, cfg["vocab_size"])
    else:
        ops.call("uniir_text_embed_packed", ids, st.p(prefix + "embeddings.word_embeddings.weight"),
                 st.p(prefix + "embeddings.position_embeddings.weight"), roff, e32, None, M, L, W, cfg["vocab_size"])
    h32, h16 = _ln2(st, e32, prefix + "embeddings.LayerNorm.", eps, R, W)
    ph = cfg.get("hidden_dropout_prob", 0.0) if drop is not None else 0.0
    pa = cfg.get("attention_probs_dropout_prob", 0.0) if drop is not None else 0.0
    s_emb = drop.next() if ph else 0
    if ph:
        ops.dropout_f32(h32, ph, s_emb, out_f32=h32, out_bf16=h16, row_map=rmap)
    stash = dict(ids=ids, e32=e32, layers=[], M=M, L=L, Ti=Ti, key_len=key_len, img16=img16, ph=ph, pa=pa,
                 s_emb=s_emb, pack=pack) if save else None
    # attention on dense rows (padding mask key_len) or on the packed rows (the item's own length is its key count)
    self_kw = dict(key_len=key_len) if pack is None else dict(row_off=roff, kv_packed=True, rows=R)
    cross_kw = {} if pack is None else dict(row_off=roff, kv_packed=False, rows=R)

    def out_dense(x16, name, resid):
        """dense -> dropout -> + residual (BertSelfOutput / BertOutput, med.py:196-200,339-343), pre-LayerNorm sum"""
        if not ph:
            return ops.linear_fwd(x16, st.w16(name + "weight"), st.p(name + "bias"), epilogue=ops.EPI_RESID_F32,
                                  resid=resid), 0
        t = ops.linear_fwd(x16, st.w16(name + "weight"), st.p(name + "bias"), epilogue=ops.EPI_RESID_F32)
        sd = drop.next()
        ops.dropout_f32(t, ph, sd, resid=resid, out_f32=t, row_map=rmap)
        return t, sd

    g = torch.empty(R, I, device=dev, dtype=torch.bfloat16)
    for i in range(layers):
        b = f"{prefix}encoder.layer.{i}."
        s, c = b + "attention.self.", b + "crossattention.self."
        qkv = ops.linear_fwd(h16, st.w16(s + "query.weight", (3 * W, W)), st.p(s + "query.bias", (3 * W,)))
        sa1 = drop.next() if pa else 0
        ao, lse1 = ops.attention_fwd_ex(qkv, 3 * W, qkv[:, W:], qkv[:, 2 * W:], 3 * W, M, L, L, heads, drop_p=pa, drop_seed=sa1,
                                        **self_kw)
        t1, so1 = out_dense(ao, b + "attention.output.dense.", h32)
        a32, a16 = _ln2(st, t1, b + "attention.output.LayerNorm.", eps, R, W)
        cq = ckv = co = lse2 = t2 = None
        sa2 = so2 = 0
        if cross:
            cq = ops.linear_fwd(a16, st.w16(c + "query.weight"), st.p(c + "query.bias"))
            ckv = ops.linear_fwd(img16, st.w16(c + "key.weight", (2 * W, img16.shape[1])), st.p(c + "key.bias", (2 * W,)))
            sa2 = drop.next() if pa else 0
            co, lse2 = ops.attention_fwd_ex(cq, W, ckv, ckv[:, W:], 2 * W, M, L, Ti, heads, drop_p=pa, drop_seed=sa2, **cross_kw)
            t2, so2 = out_dense(co, b + "crossattention.output.dense.", a32)
            c32, c16 = _ln2(st, t2, b + "crossattention.output.LayerNorm.", eps, R, W)
        else:
            c32, c16 = a32, a16
        f = torch.empty(R, I, device=dev, dtype=torch.bfloat16)
        ops.linear_fwd(c16, st.w16(b + "intermediate.dense.weight"), st.p(b + "intermediate.dense.bias"), out=f,
                       epilogue=ops.EPI_BIAS_ACT, C2=g, act=ops.ACT_GELU_ERF)
        t3, so3 = out_dense(g, b + "output.dense.", c32)
        if save:
            stash["layers"].append(dict(h16=h16, qkv=qkv, ao=ao, lse1=lse1, t1=t1, a16=a16, cq=cq, ckv=ckv, co=co,
                                        lse2=lse2, t2=t2, c16=c16, f=f, t3=t3, seeds=(sa1, so1, sa2, so2, so3)))
        h32, h16 = _ln2(st, t3, b + "output.LayerNorm.", eps, R, W)
    rows = torch.empty(M, W, device=dev, dtype=torch.float32)
    if pack is None:
        ops.call("uniir_gather_rows", h32, None, rows, M, L, W)
    else:
        ops.call("uniir_gather_rows", h32, roff, rows, M, 0, W)          # the class token = the first row of every item
    if not pool:
        return rows, stash
    rows16 = torch.empty(M, W, device=dev, dtype=torch.bfloat16)
    ops.call("uniir_cast_f32_to_bf16", rows, rows16, rows.numel())
    pre = ops.linear_fwd(rows16, st.w16(prefix + "pooler.dense.weight"), st.p(prefix + "pooler.dense.bias"),
                         epilogue=ops.EPI_RESID_F32)
    pooled = torch.empty_like(pre)
    ops.call("uniir_tanh_fwd", pre, pooled, pre.numel())
    if save:
        stash.update(rows16=rows16, pooled=pooled)
    return pooled, stash


def bert_backward(st, prefix, cfg, dpooled, stash, cross=True, pool=True):
    """returns d(img tokens) fp32 [M*Ti, enc_width] (None without cross-attention); parameter gradients accumulate into
    st.g32.  dpooled: gradient of what bert_forward returned (pooler output, or the class-token rows with pool=False)"""
    W, I, eps = cfg["hidden_size"], cfg["intermediate_size"], cfg["layer_norm_eps"]
    heads, layers = cfg["num_attention_heads"], cfg["num_hidden_layers"]
    M, L, Ti, key_len, img16 = stash["M"], stash["L"], stash["Ti"], stash["key_len"], stash["img16"]
    pack = stash.get("pack")
    R, dev = (pack.R if pack is not None else M * L), dpooled.device
    G = st.grad_view
    f32 = dict(device=dev, dtype=torch.float32)
    b16 = dict(device=dev, dtype=torch.bfloat16)

    def colsum(x, cols, name, shape=None):
        ops.call("uniir_colsum_bf16", x, cols, G(name, shape), x.shape[0], cols)

    if not pool:
        drows = dpooled.contiguous().float()
    else:
        drows = _pooler_backward(st, prefix, dpooled, stash, M, W, colsum)
    do = torch.zeros(R, W, **f32)
    if pack is None:
        ops.call("uniir_scatter_rows", drows, None, do, M, L, W)
    else:
        ops.call("uniir_scatter_rows", drows, pack.row_off, do, M, 0, W)
    dimg = torch.zeros(M * Ti, img16.shape[1], **f32) if cross else None
    g = torch.empty(R, I, **b16)
    return _bert_layers_backward(st, prefix, cfg, do, dimg, g, stash, cross, colsum)


def _pooler_backward(st, prefix, dpooled, stash, M, W, colsum):
    dev = dpooled.device
    G = st.grad_view
    f32 = dict(device=dev, dtype=torch.float32)
    b16 = dict(device=dev, dtype=torch.bfloat16)
    # pooler: pooled = tanh(rows @ Wp^T + bp)
    dpre = torch.empty(M, W, **f32)
    ops.call("uniir_tanh_bwd", stash["pooled"], dpooled.contiguous(), dpre, dpre.numel())
    dpre16 = torch.empty(M, W, **b16)
    ops.call("uniir_cast_f32_to_bf16", dpre, dpre16, dpre.numel())
    ops.linear_wgrad(dpre16, stash["rows16"], G(prefix + "pooler.dense.weight"))
    colsum(dpre16, W, prefix + "pooler.dense.bias")
    drows = torch.empty(M, W, **f32)
    ops.gemm(dpre16, st.w16(prefix + "pooler.dense.weight"), drows, M, W, W, W, W, W, b_tmaj=True, epilogue=ops.EPI_F32)
    return drows


def _bert_layers_backward(st, prefix, cfg, do, dimg, g, stash, cross, colsum):
    W, I, eps = cfg["hidden_size"], cfg["intermediate_size"], cfg["layer_norm_eps"]
    heads, layers = cfg["num_attention_heads"], cfg["num_hidden_layers"]
    M, L, Ti, key_len, img16 = stash["M"], stash["L"], stash["Ti"], stash["key_len"], stash["img16"]
    pack = stash.get("pack")
    R, dev = (pack.R if pack is not None else M * L), do.device
    roff = pack.row_off if pack is not None else None
    rmap = pack.row_map if pack is not None else None
    self_kw = dict(key_len=key_len) if pack is None else dict(row_off=roff, kv_packed=True)
    cross_kw = {} if pack is None else dict(row_off=roff, kv_packed=False)
    G = st.grad_view
    f32 = dict(device=dev, dtype=torch.float32)
    b16 = dict(device=dev, dtype=torch.bfloat16)
    ph, pa = stash["ph"], stash["pa"]
    for i in reversed(range(layers)):
        b = f"{prefix}encoder.layer.{i}."
        s, c = b + "attention.self.", b + "crossattention.self."
        sv = stash["layers"][i]
        stash["layers"][i] = None
        sa1, so1, sa2, so2, so3 = sv["seeds"]
        # ---- feed-forward sublayer: o = LN(g @ Wout^T + bout + c32), g = gelu(c16 @ Wi^T + bi)
        d16 = torch.empty(R, W, **b16)
        dt3 = ops.layernorm_bwd(sv["t3"], st.p(b + "output.LayerNorm.weight"), do, G(b + "output.LayerNorm.weight"),
                                G(b + "output.LayerNorm.bias"), eps, dx_bf16=d16, rows=R, width=W)
        if ph:      # the dense branch sees the masked gradient, the residual branch (dt3, fp32) the full one
            ops.dropout_bf16_(d16, ph, so3, row_map=rmap)
        df = torch.empty(R, I, **b16)
        ops.linear_dgrad(d16, st.w16(b + "output.dense.weight"), out=df, aux=sv["f"], act_out=g,
                         colsum=G(b + "intermediate.dense.bias"), act=ops.ACT_GELU_ERF)
        ops.linear_wgrad(d16, g, G(b + "output.dense.weight"), dbias=G(b + "output.dense.bias"))
        ops.linear_wgrad(df, sv["c16"], G(b + "intermediate.dense.weight"))
        dc = torch.empty(R, W, **f32)       # d c32 = df @ Wi + dt3 (the residual branch)
        ops.gemm(df, st.w16(b + "intermediate.dense.weight"), dc, R, W, I, I, W, W, b_tmaj=True,
                 epilogue=ops.EPI_RESID_F32, resid=dt3)
        if not cross:
            da = dc                      # mode "text": c32 is a32
        else:
            # ---- cross-attention sublayer: c = LN(co @ Wco^T + bco + a32)
            dt2 = ops.layernorm_bwd(sv["t2"], st.p(b + "crossattention.output.LayerNorm.weight"), dc,
                                    G(b + "crossattention.output.LayerNorm.weight"),
                                    G(b + "crossattention.output.LayerNorm.bias"), eps, dx_bf16=d16, rows=R, width=W)
            if ph:
                ops.dropout_bf16_(d16, ph, so2, row_map=rmap)
            ops.linear_wgrad(d16, sv["co"], G(b + "crossattention.output.dense.weight"),
                             dbias=G(b + "crossattention.output.dense.bias"))
            dco = ops.linear_dgrad(d16, st.w16(b + "crossattention.output.dense.weight"))
            dcq = torch.empty(R, W, **b16)
            dckv = torch.empty(M * Ti, 2 * W, **b16)
            ckv = sv["ckv"]
            ops.attention_bwd_ex(sv["cq"], W, ckv, ckv[:, W:], 2 * W, sv["co"], dco, sv["lse2"], dcq, W, dckv, dckv[:, W:],
                                 2 * W, M, L, Ti, heads, drop_p=pa, drop_seed=sa2, **cross_kw)
            Ew = img16.shape[1]
            ops.linear_wgrad(dckv, img16, G(c + "key.weight", (2 * W, Ew)), dbias=G(c + "key.bias", (2 * W,)))
            ops.gemm(dckv, st.w16(c + "key.weight", (2 * W, Ew)), dimg, M * Ti, Ew, 2 * W, 2 * W, Ew, Ew, b_tmaj=True,
                     epilogue=ops.EPI_RESID_F32, resid=dimg)          # dimg += dckv @ Wkv (in place: same element r/w)
            ops.linear_wgrad(dcq, sv["a16"], G(c + "query.weight"), dbias=G(c + "query.bias"))
            da = torch.empty(R, W, **f32)       # d a32 = dcq @ Wcq + dt2
            ops.gemm(dcq, st.w16(c + "query.weight"), da, R, W, W, W, W, W, b_tmaj=True, epilogue=ops.EPI_RESID_F32,
                     resid=dt2)
        # ---- self-attention sublayer: a = LN(ao @ Wo^T + bo + h32)
        dt1 = ops.layernorm_bwd(sv["t1"], st.p(b + "attention.output.LayerNorm.weight"), da,
                                G(b + "attention.output.LayerNorm.weight"), G(b + "attention.output.LayerNorm.bias"),
                                eps, dx_bf16=d16, rows=R, width=W)
        if ph:
            ops.dropout_bf16_(d16, ph, so1, row_map=rmap)
        ops.linear_wgrad(d16, sv["ao"], G(b + "attention.output.dense.weight"), dbias=G(b + "attention.output.dense.bias"))
        dao = ops.linear_dgrad(d16, st.w16(b + "attention.output.dense.weight"))
        qkv = sv["qkv"]
        dqkv = torch.empty(R, 3 * W, **b16)
        ops.attention_bwd_ex(qkv, 3 * W, qkv[:, W:], qkv[:, 2 * W:], 3 * W, sv["ao"], dao, sv["lse1"], dqkv, 3 * W,
                             dqkv[:, W:], dqkv[:, 2 * W:], 3 * W, M, L, L, heads, drop_p=pa, drop_seed=sa1, **self_kw)
        ops.linear_wgrad(dqkv, sv["h16"], G(s + "query.weight", (3 * W, W)), dbias=G(s + "query.bias", (3 * W,)))
        do = torch.empty(R, W, **f32)       # d h32 = dqkv @ Wqkv + dt1
        ops.gemm(dqkv, st.w16(s + "query.weight", (3 * W, W)), do, R, W, 3 * W, 3 * W, W, W, b_tmaj=True,
                 epilogue=ops.EPI_RESID_F32, resid=dt1)
    if ph:
        ops.dropout_f32(do, ph, stash["s_emb"], out_f32=do, row_map=rmap)
    de = ops.layernorm_bwd(stash["e32"], st.p(prefix + "embeddings.LayerNorm.weight"), do,
                           G(prefix + "embeddings.LayerNorm.weight"), G(prefix + "embeddings.LayerNorm.bias"), eps,
                           rows=R, width=W)
    if pack is None:
        ops.call("uniir_text_embed_bwd", stash["ids"], de, G(prefix + "embeddings.word_embeddings.weight"),
                 G(prefix + "embeddings.position_embeddings.weight"), M, L, W, cfg["vocab_size"])
    else:
        ops.call("uniir_text_embed_bwd_packed", stash["ids"], de, roff, G(prefix + "embeddings.word_embeddings.weight"),
                 G(prefix + "embeddings.position_embeddings.weight"), M, L, W, cfg["vocab_size"])
    return dimg


# ------------------------------------------------------------------------------------------------------------
# module
# ------------------------------------------------------------------------------------------------------------
class _EncodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, ids, key_len, images, anchor, pack=None):
        save = bool(ctx.needs_input_grad[4])
        st = model._online
        drop = model._drop_seeds()
        tok, Ti, vst = vit_forward(st, model._conv16, "visual_encoder.", model.vit_cfg, model.image_size, images, save,
                                   drop=drop)
        pooled, bst = bert_forward(st, "text_encoder.", model.med_cfg, ids, key_len, tok, Ti, save, drop=drop, pack=pack)
        ctx.model, ctx.vst, ctx.bst = model, vst, bst
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        model, st = ctx.model, ctx.model._online
        bst, vst = ctx.bst, ctx.vst
        ctx.bst = ctx.vst = None
        dimg = bert_backward(st, "text_encoder.", model.med_cfg, dpooled, bst)
        vit_backward(st, model._dconv, "visual_encoder.", model.vit_cfg, dimg, vst)
        return None, None, None, None, None, None


class _SoftTargetLossFn(torch.autograd.Function):
    """blip_ff.py:155-231,250-252.  Everything that reads the queues runs in forward (the queues are overwritten before
    backward): the unit gradients are kept and scaled by d(loss) in backward.  With hard negatives (`ni` [b*N] rows of
    the flat batch, `ids_neg` their candidate ids) the candidate side is [p_m | nc_m (un-normalised, like the
    reference) | cand_queue[:, b*N:]] and the id row [p ids | negative ids | idx_queue[b*N:]] (:159-170,196-205)."""

    @staticmethod
    def forward(ctx, emb, temp, model, emb_m, qi, pi, ids_row, alpha, ni, ids_neg):
        b, E, dev = qi.numel(), emb.shape[1], emb.device
        K = model.queue_size
        n = b + K
        f32 = dict(device=dev, dtype=torch.float32)

        def sel(src, idx):
            rows = idx.numel()
            out, inv = torch.empty(rows, E, **f32), torch.empty(rows, **f32)
            ops.call("uniir_select_normalize", src, idx, out, inv, rows, E)
            return out, inv

        q, invq = sel(emb, qi)
        p, invp = sel(emb, pi)
        q_m, _ = sel(emb_m, qi)
        p_m, _ = sel(emb_m, pi)
        hn = 0
        nc_m = None
        if ni is not None:
            hn = ni.numel()
            nc_m = emb_m.index_select(0, ni.long()).contiguous()      # rows emb_m[ni], not normalised (a copy, no arithmetic)
            ids_all = torch.cat([ids_row, ids_neg, model.idx_queue[0, hn:]])
        else:
            ids_all = torch.cat([ids_row, model.idx_queue[0]])
        # column blocks of the two logit matrices: (row-major features [r,E]) or (queue [E,K], first column used)
        cand_blocks = [("rows", p_m, b)] + ([("rows", nc_m, hn)] if hn else []) + [("queue", model.cand_queue, hn)]
        query_blocks = [("rows", q_m, b), ("queue", model.query_queue, 0)]

        def sims(a, blocks):
            """[a @ block_0^T | a @ block_1^T | ...] without materialising the concatenation"""
            out = torch.empty(b, n, **f32)
            col = 0
            for kind, t, arg in blocks:
                if kind == "rows":
                    ops.call("uniir_sgemm", a, E, 1, t, 1, E, out[:, col:], n, b, arg, E, 1.0)
                    col += arg
                else:
                    ops.call("uniir_sgemm", a, E, 1, t[:, arg:], K, 1, out[:, col:], n, b, K - arg, E, 1.0)
                    col += K - arg
            return out

        def dfeat(dsim, blocks):
            """d a = sum over blocks of dsim[:, cols] @ block (the momentum / queue features carry no gradient)"""
            da = torch.empty(b, E, **f32)
            col, first = 0, True
            for kind, t, arg in blocks:
                fn = "uniir_sgemm" if first else "uniir_sgemm_acc"
                if kind == "rows":
                    ops.call(fn, dsim[:, col:], n, 1, t, E, 1, da, E, b, E, arg, 1.0)
                    col += arg
                else:       # reduction over the K queue columns, 2 x 6 output tiles: the deterministic split-K form
                    lib = _lib.load()
                    need_ws = int(lib.uniir_sgemm_splitk_workspace_bytes(b, E, K - arg))
                    ws = ops._splitk_workspace(da.device, need_ws, tag="blip_dfeat")
                    ops.call("uniir_sgemm_splitk", dsim[:, col:], n, 1, t[:, arg:], 1, K, da, E, b, E, K - arg, 1.0,
                             0 if first else 1, ws, ws.numel())
                    col += K - arg
                first = False
            return da

        need = ctx.needs_input_grad[0]
        rl, hit, rdt, dfe = [], None, [], []
        for a, a_m, blocks in ((q, q_m, cand_blocks), (p, p_m, query_blocks)):
            s, s_m = sims(a, blocks), sims(a_m, blocks)
            row_loss, row_hit = torch.empty(b, **f32), torch.empty(b, **f32)
            dsim = torch.empty(b, n, **f32) if need else None
            row_dt = torch.empty(b, **f32) if need else None
            ops.call("uniir_softce", s, s_m, temp, ids_row, ids_all, b, n, float(alpha), 1.0 / (2 * b), None, row_loss,
                     row_hit, dsim, row_dt)
            rl.append(row_loss)
            hit = row_hit if hit is None else hit
            if need:
                dfe.append(dfeat(dsim, blocks))
                rdt.append(row_dt)
        loss = (rl[0].sum() + rl[1].sum()) / (2 * b)
        acc = hit.mean()
        if need:
            demb = torch.zeros_like(emb)
            ops.call("uniir_select_normalize_bwd", q, invq, dfe[0], qi, demb, b, E)
            ops.call("uniir_select_normalize_bwd", p, invp, dfe[1], pi, demb, b, E)
            ctx.save_for_backward(demb, rdt[0].sum() + rdt[1].sum())
        if nc_m is None:
            nc_m = q_m.new_zeros(0, E)
        ctx.mark_non_differentiable(acc, q_m, p_m, nc_m)
        return loss, acc, q_m, p_m, nc_m

    @staticmethod
    def backward(ctx, dloss, _dacc, _dq, _dp, _dn):
        demb, dtemp = ctx.saved_tensors
        return demb * dloss, dtemp * dloss, None, None, None, None, None, None, None, None


class BLIPFeatureFusion(nn.Module):
    SCORE_FUSION = False

    def __init__(self, med_config="backbone/configs/med_config.json", image_size=224, vit="base", vit_grad_ckpt=False,
                 vit_ckpt_layer=0, embed_dim=768, queue_size=57600, momentum=0.995, config=None, seed=0,
                 vit_config=None):
        super().__init__()
        if isinstance(med_config, str):
            med_config = json.load(open(med_config))
        self.med_cfg = dict(MED_DEFAULT)
        self.med_cfg.update({k: v for k, v in dict(med_config).items() if k in MED_DEFAULT})
        self.vit_cfg = dict(vit_config or VIT_CONFIGS[vit])
        self.image_size = self.vit_cfg.get("img_size", image_size)
        W, D = self.med_cfg["hidden_size"], self.vit_cfg["embed_dim"]
        if W // self.med_cfg["num_attention_heads"] != 64 or D // self.vit_cfg["num_heads"] != 64 or W % 64 or D % 64:
            raise ValueError("the attention kernels are built for head_dim 64 (BLIP base / large both are)")
        sf = self.SCORE_FUSION
        if not sf and embed_dim != W:
            raise ValueError("BLIP_FF returns the BERT pooler output: embed_dim must equal the hidden size")
        self.med_cfg["encoder_width"] = D
        self.queue_size, self.momentum, self.embed_dim, self.config = queue_size, momentum, embed_dim, config
        vis = [("visual_encoder." + n, s) for n, s in vit_param_shapes(self.vit_cfg, self.image_size)]
        txt = [("text_encoder." + n, s) for n, s in bert_param_shapes(self.med_cfg, D, pooler=not sf)]
        if sf:     # blip_sf.py:35-47: no pooler, projection heads, cross-attention present but frozen -> kept at the END of
            # the flat stores so that the optimizer range stops before it (a frozen tensor must not see weight decay)
            proj = [("vision_proj.weight", (embed_dim, D)), ("vision_proj.bias", (embed_dim,)),
                    ("text_proj.weight", (embed_dim, W)), ("text_proj.bias", (embed_dim,))]
            self._frozen = {n for n, _ in txt if "crossattention" in n}
            self._shapes = vis + [t for t in txt if t[0] not in self._frozen] + proj + [t for t in txt if t[0] in self._frozen]
        else:
            self._frozen = set()
            self._shapes = vis + txt
        g = torch.Generator().manual_seed(seed)
        for n, shp in self._shapes:
            leaf = n.rsplit(".", 1)[-1]
            if "norm" in n.lower() and leaf == "weight":
                v = torch.ones(shp)
            elif leaf == "bias":
                v = torch.zeros(shp)
            else:
                v = torch.randn(shp, generator=g) * 0.02
            _attach(self, n, nn.Parameter(v, requires_grad=n not in self._frozen))
        for n, shp in self._shapes:      # momentum encoders start as copies (blip_ff.py:280-285 copy_params)
            enc, rest = n.split(".", 1)
            _attach(self, f"{enc}_m.{rest}", nn.Parameter(self.get_parameter(n).detach().clone(), requires_grad=False))
        self.text_encoder.embeddings.register_buffer(
            "position_ids", torch.arange(self.med_cfg["max_position_embeddings"]).expand((1, -1)))
        self.text_encoder_m.embeddings.register_buffer(
            "position_ids", torch.arange(self.med_cfg["max_position_embeddings"]).expand((1, -1)))
        self.register_buffer("query_queue", nn.functional.normalize(torch.randn(embed_dim, queue_size, generator=g), dim=0))
        self.register_buffer("cand_queue", nn.functional.normalize(torch.randn(embed_dim, queue_size, generator=g), dim=0))
        self.register_buffer("idx_queue", torch.full((1, queue_size), -100))
        self.register_buffer("new_ptr_queue", torch.zeros(1, dtype=torch.long))
        self.temp = nn.Parameter(0.07 * torch.ones([]))
        self._online = self._mom = None
        self._ptr_host = None
        self.check_masks = True
        # BERT on the rows up to each caption's valid length only (TextPack): same embeddings bit for bit, 47 % fewer BERT rows at
        # the bench's caption lengths.  False = the padded rows of the reference, all L positions (A/B, tests)
        self.pack_text = True
        self.last_text_rows = None        # (rows BERT ran on, M * L) of the last encode call (bench: executed FLOPs)

    # ---- reference surface ---------------------------------------------------------------------------------
    def get_img_preprocess_fn(self):
        from .blip_front import get_blip_transform
        return get_blip_transform(self.image_size, min_scale=0.5, is_train=self.training)

    def get_tokenizer(self):
        from .blip_front import init_tokenizer
        tok = init_tokenizer()
        max_len = self.config.tokenizer_max_length

        def tokenizer_wrapper(txt):
            return tok(txt, padding="max_length", truncation=True, max_length=max_len, return_tensors="pt")

        return tokenizer_wrapper

    @torch.no_grad()
    def copy_params(self):
        for n, _ in self._shapes:
            enc, rest = n.split(".", 1)
            self.get_parameter(f"{enc}_m.{rest}").data.copy_(self.get_parameter(n).data)

    # ---- flat storage --------------------------------------------------------------------------------------
    def _online_params(self):
        return [(n, self.get_parameter(n)) for n, _ in self._shapes] + [("temp", self.temp)]

    def _ensure_flat(self):
        dev = self.temp.device
        if dev.type != "cuda":
            raise RuntimeError("uniir_amd BLIPFeatureFusion runs on an MI355X only (no CPU path); move the model to cuda")
        st = self._online
        if st is not None and st.p32.device == dev and all(
                p.data_ptr() == st.p32.data_ptr() + 4 * st.off[n] for n, p in self._online_params()):
            return self._flat_dict()
        online = FlatStore(self._shapes + [("temp", ())], dev, True)
        mom = FlatStore(self._shapes, dev, False)
        for n, p in self._online_params():
            online.p(n).copy_(p.data.float())
            p.data = online.p(n)
            if n not in self._frozen:
                p.grad = online.grad_view(n)
        for n, _ in self._shapes:
            enc, rest = n.split(".", 1)
            pm = self.get_parameter(f"{enc}_m.{rest}")
            mom.p(n).copy_(pm.data.float())
            pm.data = mom.p(n)
        self._online, self._mom = online, mom
        D, P = self.vit_cfg["embed_dim"], self.vit_cfg["patch_size"]
        kpad = (3 * P * P + 63) // 64 * 64
        self._conv16 = torch.zeros(D, kpad, device=dev, dtype=torch.bfloat16)
        self._conv16_m = torch.zeros(D, kpad, device=dev, dtype=torch.bfloat16)
        self._dconv = torch.zeros(D, kpad, device=dev, dtype=torch.float32)
        self._version = -1
        self.refresh_shadow()
        return self._flat_dict()

    def _flat_dict(self):
        """what NativeAdamW reads: one weight-decay group over everything trainable (uniir_blip/train.py:193-197); frozen
        tensors (BLIP_SF's cross-attention) sit between the trainable range and temp and are skipped"""
        st = self._online
        d = dict(p32=st.p32, g32=st.g32, w16=st.w16_buf, total=st.total, split=0)
        if self._frozen:
            first = min(st.off[n] for n in self._frozen)
            d["ranges"] = [(0, first, 1), (st.off["temp"], st.total, 1)]
        return d

    def optimizer_groups(self):
        """(no-decay params, decay params) for NativeAdamW: the reference applies weight decay to every parameter"""
        return [], [p for n, p in self._online_params() if n not in self._frozen]

    def _refresh_conv(self, momentum_too=True):
        D, P = self.vit_cfg["embed_dim"], self.vit_cfg["patch_size"]
        pairs = [(self._online, self._conv16)] + ([(self._mom, self._conv16_m)] if momentum_too else [])
        for st, dst in pairs:
            ops.call("uniir_cast_pad_rows", st.p("visual_encoder.patch_embed.proj.weight"), dst, D, 3 * P * P, dst.shape[1])

    def refresh_shadow(self):
        self._online.refresh_shadow()
        self._mom.refresh_shadow()
        self._refresh_conv()
        self._version = self._param_version()

    def _param_version(self):
        # torch-side writes (load_state_dict, manual edits) bump these; the kernels' own updates (AdamW, EMA) refresh
        # the bf16 shadows themselves.  temp has no shadow, and is clamp_()ed every step: left out.
        return sum(p._version for n, p in self.named_parameters() if n != "temp")

    def _sync(self):
        self._ensure_flat()
        if self._version != self._param_version():
            self.refresh_shadow()

    def zero_grad(self, set_to_none=False):
        if self._online is not None:
            self._online.g32.zero_()
            for n, p in self._online_params():
                if n not in self._frozen and (p.grad is None or p.grad.data_ptr() != self._online.grad_view(n).data_ptr()):
                    p.grad = self._online.grad_view(n)
        else:
            super().zero_grad(set_to_none=set_to_none)

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self._ptr_host = None
        return out

    # ---- encoders ------------------------------------------------------------------------------------------
    def _drop_seeds(self):
        """train mode with a non-zero rate: a fresh seed stream for one encoder pass (else None: no dropout launches)"""
        rates = (self.med_cfg.get("hidden_dropout_prob", 0.0), self.med_cfg.get("attention_probs_dropout_prob", 0.0),
                 self.vit_cfg.get("drop_path_rate", 0.0))
        return ops.DropSeeds() if self.training and any(r > 0 for r in rates) else None

    def _text_inputs(self, txt):
        ids = txt["input_ids"] if isinstance(txt, dict) else txt.input_ids
        mask = txt["attention_mask"] if isinstance(txt, dict) else txt.attention_mask
        if self.check_masks and not bool((mask[:, :-1] >= mask[:, 1:]).all()):
            raise ValueError("attention_mask must be a prefix mask (ones, then padding): tokenizer padding='max_length'")
        return ids.to(torch.int32).contiguous(), mask.sum(1).to(torch.int32).contiguous()

    def _text_pack(self, txt, key_len, L):
        """the TextPack of this token batch (None: pack_text off, or nothing to drop).  Built once per batch -- the online and the
        momentum encoder of a step see the same tokens -- and remembered on the attention-mask tensor under its version."""
        if not self.pack_text:
            return None
        mask = txt["attention_mask"] if isinstance(txt, dict) else txt.attention_mask
        try:
            ver = mask._version
        except RuntimeError:
            ver = None
        hit = getattr(mask, "_uniir_pack", None)
        if ver is not None and hit is not None and hit[0] == ver and hit[1] == L:
            return hit[2]
        # the captions' lengths on the host: attached by the prefetcher while the mask was still there (mask._uniir_lens, tied to the
        # tensor version like CLIP's hint), else one device -> host read
        lens = getattr(mask, "_uniir_lens", None)
        if (isinstance(lens, torch.Tensor) and not lens.is_cuda and lens.dim() == 1 and lens.shape[0] == mask.shape[0]
                and not lens.is_floating_point() and ver is not None and getattr(mask, "_uniir_lens_version", None) == ver):
            pack = TextPack.build(lens, L, key_len.device)
        else:
            pack = TextPack.build(key_len, L)
        if ver is not None:
            mask._uniir_pack = (ver, L, pack)
        return pack

    def encode_multimodal_input(self, txt_dict_batched, image_batched, txt_mask=None, img_mask=None, use_momentum=False):
        """blip_ff.py:82-116: BERT(text) cross-attending to ViT(image) tokens -> pooler_output [n, embed_dim]"""
        self._sync()
        ids, key_len = self._text_inputs(txt_dict_batched)
        pack = self._text_pack(txt_dict_batched, key_len, ids.shape[1])
        self.last_text_rows = (pack.R if pack is not None else ids.numel(), ids.numel())
        if use_momentum:
            with torch.no_grad():
                drop = self._drop_seeds()
                tok, Ti, _ = vit_forward(self._mom, self._conv16_m, "visual_encoder.", self.vit_cfg, self.image_size,
                                         image_batched, False, drop=drop)
                return bert_forward(self._mom, "text_encoder.", self.med_cfg, ids, key_len, tok, Ti, False, drop=drop,
                                    pack=pack)[0]
        anchor = torch.zeros(1, device=ids.device, requires_grad=torch.is_grad_enabled())
        return _EncodeFn.apply(self, ids, key_len, image_batched, anchor, pack)

    @torch.no_grad()
    def _momentum_update(self):
        """blip_ff.py:287-292 as one fused pass over the flat encoder region (temp sits after it)"""
        self._sync()
        ops.call("uniir_ema_update", self._mom.p32, self._online.p32, self._mom.w16_buf, self._mom.total,
                 float(self.momentum))
        self._refresh_conv()

    @torch.no_grad()
    def _dequeue_and_enqueue(self, query_feats, cand_feats, idxs):
        """blip_ff.py:294-310"""
        idxs = comm.all_gather_rows(idxs.view(-1, 1))
        query_feats = comm.all_gather_rows(query_feats)
        cand_feats = comm.all_gather_rows(cand_feats)
        bsz = query_feats.shape[0]
        if self._ptr_host is None:
            self._ptr_host = int(self.new_ptr_queue)
        ptr = self._ptr_host
        assert self.queue_size % bsz == 0  # same requirement as the reference
        self.query_queue[:, ptr:ptr + bsz] = query_feats.T
        self.cand_queue[:, ptr:ptr + bsz] = cand_feats.T
        self.idx_queue[:, ptr:ptr + bsz] = idxs.T
        self._ptr_host = (ptr + bsz) % self.queue_size
        self.new_ptr_queue.fill_(self._ptr_host)

    def compute_contrastive_loss(self, batch, alpha):
        index_mapping = batch["index_mapping"]
        hard = "neg_cand_list" in index_mapping
        dev = self.temp.device
        txt, img = batch["txt_batched"], batch["image_batched"]
        ids_row = torch.as_tensor(batch["p_did_list"], device=dev).to(torch.int64).flatten()
        qi = torch.tensor(index_mapping["query"], dtype=torch.int32).flatten().to(dev)
        pi = torch.tensor(index_mapping["pos_cand"], dtype=torch.int32).flatten().to(dev)
        ni = ids_neg = None
        if hard:     # blip_ff.py:127-131: [bs, neg_num] rows of the flat batch and their candidate ids
            ni = torch.tensor(index_mapping["neg_cand_list"], dtype=torch.int32).flatten().to(dev)
            ids_neg = torch.as_tensor(batch["nc_dids_list"], device=dev).to(torch.int64).flatten()
        with torch.no_grad():
            self.temp.clamp_(0.001, 0.5)
        emb = self.encode_multimodal_input(txt, img, batch.get("txt_mask_batched"), batch.get("image_mask_batched"))
        self._momentum_update()
        emb_m = self.encode_multimodal_input(txt, img, batch.get("txt_mask_batched"), batch.get("image_mask_batched"),
                                             use_momentum=True)
        loss, acc, q_m, p_m, nc_m = _SoftTargetLossFn.apply(emb, self.temp, self, emb_m, qi, pi, ids_row, alpha, ni, ids_neg)
        if hard and not bool(torch.rand(1) < 0.5):
            # blip_ff.py:233-246: a coin flip (same host generator as the reference) enqueues each query's FIRST negative
            bs = qi.numel()
            nneg = ni.numel() // bs
            self._dequeue_and_enqueue(q_m.detach(), nc_m.detach().view(bs, nneg, -1)[:, 0, :].contiguous(),
                                      ids_neg.view(bs, nneg)[:, 0].contiguous())
        else:
            self._dequeue_and_enqueue(q_m.detach(), p_m.detach(), ids_row)
        return {"loss": loss, "accuracy": acc}

    def encode_mbeir_batch(self, batch):
        id_list = batch.get("did_list") or batch.get("qid_list")
        if id_list is None:
            raise ValueError("id_list not found in batch.")
        emb = self.encode_multimodal_input(batch["txt_batched"], batch["image_batched"], batch.get("txt_mask_batched"),
                                           batch.get("image_mask_batched"))
        assert emb.size(0) == len(id_list), "embeddings and id_batched must have the same batch size."
        return emb, id_list

    def forward(self, batch, alpha=None, encode_mbeir_batch=False):
        if encode_mbeir_batch:
            return self.encode_mbeir_batch(batch)
        return self.compute_contrastive_loss(batch, alpha)


class _EncodeSFFn(torch.autograd.Function):
    """BLIP_SF encoder (blip_sf.py:97-172): text_proj(BERT mode "text" class token) * txt_mask + vision_proj(ViT class
    token) * img_mask"""

    @staticmethod
    def forward(ctx, model, ids, key_len, images, tmask, imask, anchor, pack=None):
        save = bool(ctx.needs_input_grad[6])
        emb, stash = model._encode_sf(model._online, model._conv16, ids, key_len, images, tmask, imask, save, pack)
        ctx.model, ctx.stash = model, stash
        return emb

    @staticmethod
    def backward(ctx, demb):
        model, st, S = ctx.model, ctx.model._online, ctx.stash
        ctx.stash = None
        M, E = demb.shape
        dev = demb.device
        G = st.grad_view
        dt, di = torch.empty(M, E, device=dev), torch.empty(M, E, device=dev)
        ops.call("uniir_fuse_embeddings_bwd", demb.contiguous(), S["tmask"], S["imask"], dt, di, M, E)

        def proj_bwd(dy, x16, name):
            dy16 = torch.empty(M, E, device=dev, dtype=torch.bfloat16)
            ops.call("uniir_cast_f32_to_bf16", dy, dy16, dy.numel())
            ops.linear_wgrad(dy16, x16, G(name + ".weight"), dbias=G(name + ".bias"))
            K = x16.shape[1]
            dx = torch.empty(M, K, device=dev, dtype=torch.float32)
            ops.gemm(dy16, st.w16(name + ".weight"), dx, M, K, E, E, K, K, b_tmaj=True, epilogue=ops.EPI_F32)
            return dx

        dtfeat = proj_bwd(dt, S["tfeat16"], "text_proj")
        difeat = proj_bwd(di, S["ifeat16"], "vision_proj")
        bert_backward(st, "text_encoder.", model.med_cfg, dtfeat, S["bst"], cross=False, pool=False)
        T, D = S["T"], difeat.shape[1]
        dtok = torch.zeros(M * T, D, device=dev, dtype=torch.float32)
        ops.call("uniir_scatter_rows", difeat, None, dtok, M, T, D)
        vit_backward(st, model._dconv, "visual_encoder.", model.vit_cfg, dtok, S["vst"])
        return None, None, None, None, None, None, None, None


class BLIPScoreFusion(BLIPFeatureFusion):
    """src/models/uniir_blip/blip_scorefusion/blip_sf.py: the same momentum / queue loss as BLIP_FF on score-fused
    embeddings; state-dict keys add vision_proj{,_m}.* / text_proj{,_m}.*, the BERT has no pooler, its cross-attention
    parameters exist but are frozen (:66-69) and unused (mode "text")."""
    SCORE_FUSION = True

    def _encode_sf(self, st, conv16, ids, key_len, images, tmask, imask, save, pack=None):
        dev = ids.device
        M = ids.shape[0]
        E = self.embed_dim
        drop = self._drop_seeds()
        tok, T, vst = vit_forward(st, conv16, "visual_encoder.", self.vit_cfg, self.image_size, images, save, drop=drop)
        ifeat16 = tok.view(M, T, -1)[:, 0].contiguous()                     # class-token rows (a copy)
        tfeat, bst = bert_forward(st, "text_encoder.", self.med_cfg, ids, key_len, None, 0, save, cross=False, pool=False,
                                  drop=drop, pack=pack)
        tfeat16 = torch.empty(M, tfeat.shape[1], device=dev, dtype=torch.bfloat16)
        ops.call("uniir_cast_f32_to_bf16", tfeat, tfeat16, tfeat.numel())
        temb = ops.linear_fwd(tfeat16, st.w16("text_proj.weight"), st.p("text_proj.bias"), epilogue=ops.EPI_RESID_F32)
        iemb = ops.linear_fwd(ifeat16, st.w16("vision_proj.weight"), st.p("vision_proj.bias"), epilogue=ops.EPI_RESID_F32)
        emb = torch.empty(M, E, device=dev, dtype=torch.float32)
        ops.call("uniir_fuse_embeddings", temb, iemb, tmask, imask, emb, M, E)
        stash = dict(vst=vst, bst=bst, T=T, tfeat16=tfeat16, ifeat16=ifeat16, tmask=tmask, imask=imask) if save else None
        return emb, stash

    def encode_multimodal_input(self, txt_dict_batched, image_batched, txt_mask=None, img_mask=None, use_momentum=False):
        self._sync()
        ids, key_len = self._text_inputs(txt_dict_batched)
        pack = self._text_pack(txt_dict_batched, key_len, ids.shape[1])
        self.last_text_rows = (pack.R if pack is not None else ids.numel(), ids.numel())
        M = ids.shape[0]
        dev = ids.device
        ones = torch.ones(M, dtype=torch.int64, device=dev)
        tmask = ones if txt_mask is None else txt_mask.to(dev).to(torch.int64).contiguous()
        imask = ones if img_mask is None else img_mask.to(dev).to(torch.int64).contiguous()
        if use_momentum:
            with torch.no_grad():
                return self._encode_sf(self._mom, self._conv16_m, ids, key_len, image_batched, tmask, imask, False, pack)[0]
        anchor = torch.zeros(1, device=dev, requires_grad=torch.is_grad_enabled())
        return _EncodeSFFn.apply(self, ids, key_len, image_batched, tmask, imask, anchor, pack)


def blip_sf(pretrained="", **kwargs):
    model = BLIPScoreFusion(**kwargs)
    if pretrained:
        from .host_utils import load_checkpoint_file
        sd = load_checkpoint_file(pretrained)
        msg = model.load_state_dict(sd.get("model", sd), strict=False)
        print("missing keys:")
        print(msg.missing_keys)
    return model


def blip_ff(pretrained="", **kwargs):
    model = BLIPFeatureFusion(**kwargs)
    if pretrained:
        from .host_utils import load_checkpoint_file
        sd = load_checkpoint_file(pretrained)
        msg = model.load_state_dict(sd.get("model", sd), strict=False)
        print("missing keys:")
        print(msg.missing_keys)
    return model
