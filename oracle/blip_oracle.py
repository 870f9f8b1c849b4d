"""CPU ORACLE (test infrastructure, NOT the product path) -- plain PyTorch fp32 restatement of the BLIP_FF rows of the
hot path (a17 encode_multimodal_input, a18 contrastive loss + momentum + queues, a19 MED BERT,
a20 BLIP ViT).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

Restated from (paths relative to the UniIR reference tree):
  * src/models/uniir_blip/backbone/vit.py:24-268   VisionTransformer (pre-LN blocks, fused qkv Linear, erf GELU,
    LayerNorm eps 1e-6, final norm, ALL tokens returned); PatchEmbed is timm's (Conv2d(3, D, 16, 16, bias=True) +
    flatten(2).transpose(1, 2); third-party, restated from its published behaviour)
  * src/models/uniir_blip/backbone/med.py:52-100 (embeddings), :101-232 (self / cross attention, additive
    (1 - mask) * -10000 key mask :687-688), :235-247, :303-330 (post-LN residual sublayers), :499-511 (tanh pooler),
    BertModel.forward in mode "multimodal" (:691-829)
  * src/models/uniir_blip/blip_featurefusion/blip_ff.py:82-116 (encode), :118-257 (loss), :288-310 (momentum, queues)
PARITY PINNING: tests/golden/g6_med.npz, g7_vit.npz, g8_blipff.npz were produced by importing those reference files
here (tests/golden/make_golden_blip.py); tests/test_oracle_blip.py holds this restatement to them.
Dropout / DropPath: the fixtures run with probability 0.  The train-mode sites are restated as optional multiplicative
masks (`masks(kind, shape)`, called in forward order; vit.py:79-80 DropPath, med.py:84,175,198,212,341 dropout) so that
tests can hold the HIP path's masked forward / backward to the same arithmetic with the SAME masks; the random stream
itself is the framework's, not torch's.
"""
import math

import torch
import torch.nn.functional as F


def _ln(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


# ------------------------------------------------------------------------------------------------ BLIP ViT
def vit_forward(sd, x, cfg, prefix="", masks=None):
    """sd: state dict with timm-style keys under `prefix`; x [N,3,H,W] -> tokens [N, 1+g*g, D]"""
    one = lambda kind, shape: 1.0
    masks = masks or one
    p = prefix
    D, P, heads = cfg["embed_dim"], cfg["patch_size"], cfg["num_heads"]
    t = F.conv2d(x, sd[p + "patch_embed.proj.weight"], sd[p + "patch_embed.proj.bias"], stride=P)
    t = t.flatten(2).transpose(1, 2)
    t = torch.cat([sd[p + "cls_token"].expand(t.shape[0], -1, -1), t], dim=1)
    t = t + sd[p + "pos_embed"][:, : t.shape[1], :]
    hd = D // heads
    for i in range(cfg["depth"]):
        b = f"{p}blocks.{i}."
        h = _ln(t, sd[b + "norm1.weight"], sd[b + "norm1.bias"], 1e-6)
        N, L, _ = h.shape
        qkv = (h @ sd[b + "attn.qkv.weight"].t() + sd[b + "attn.qkv.bias"]).reshape(N, L, 3, heads, hd).permute(2, 0, 3, 1, 4)
        a = torch.softmax((qkv[0] @ qkv[1].transpose(-2, -1)) * hd ** -0.5, dim=-1)
        o = (a @ qkv[2]).transpose(1, 2).reshape(N, L, D)
        t = t + (o @ sd[b + "attn.proj.weight"].t() + sd[b + "attn.proj.bias"]) * masks("path", (N, 1, 1))
        h = _ln(t, sd[b + "norm2.weight"], sd[b + "norm2.bias"], 1e-6)
        h = F.gelu(h @ sd[b + "mlp.fc1.weight"].t() + sd[b + "mlp.fc1.bias"])
        t = t + (h @ sd[b + "mlp.fc2.weight"].t() + sd[b + "mlp.fc2.bias"]) * masks("path", (N, 1, 1))
    return _ln(t, sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6)


# ------------------------------------------------------------------------------------------------ MED BERT
def _attn(sd, pre, q_in, kv_in, heads, add_mask, pmask=1.0):
    """BertSelfAttention: q from q_in, k/v from kv_in; additive mask [N,1,1,Lk] or None"""
    N, Lq, W = q_in.shape
    hd = W // heads
    q = (q_in @ sd[pre + "query.weight"].t() + sd[pre + "query.bias"]).view(N, Lq, heads, hd).transpose(1, 2)
    k = (kv_in @ sd[pre + "key.weight"].t() + sd[pre + "key.bias"]).view(N, -1, heads, hd).transpose(1, 2)
    v = (kv_in @ sd[pre + "value.weight"].t() + sd[pre + "value.bias"]).view(N, -1, heads, hd).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    if add_mask is not None:
        s = s + add_mask
    return ((torch.softmax(s, dim=-1) * pmask) @ v).transpose(1, 2).reshape(N, Lq, W)


def bert_forward(sd, ids, mask, enc_hidden, cfg, prefix="", masks=None):
    """BertModel(mode='multimodal'): -> (last_hidden_state [N,L,W], pooler_output [N,W])"""
    p = prefix
    eps, heads = cfg["layer_norm_eps"], cfg["num_attention_heads"]
    N, L = ids.shape
    masks = masks or (lambda kind, shape: 1.0)
    h = sd[p + "embeddings.word_embeddings.weight"][ids] + sd[p + "embeddings.position_embeddings.weight"][:L]
    h = _ln(h, sd[p + "embeddings.LayerNorm.weight"], sd[p + "embeddings.LayerNorm.bias"], eps)
    h = h * masks("hidden", h.shape)
    add_mask = (1.0 - mask[:, None, None, :].to(h.dtype)) * -10000.0
    for i in range(cfg["num_hidden_layers"]):
        b = f"{p}encoder.layer.{i}."
        ctx = _attn(sd, b + "attention.self.", h, h, heads, add_mask, masks("attn", (N, heads, L, L)))
        dense = ctx @ sd[b + "attention.output.dense.weight"].t() + sd[b + "attention.output.dense.bias"]
        h = _ln(dense * masks("hidden", h.shape) + h,
                sd[b + "attention.output.LayerNorm.weight"], sd[b + "attention.output.LayerNorm.bias"], eps)
        if enc_hidden is not None:      # mode "multimodal"; mode "text" (BLIP_SF) skips the cross-attention sublayer
            ctx = _attn(sd, b + "crossattention.self.", h, enc_hidden, heads, None,   # image attention mask is all ones
                        masks("attn", (N, heads, L, enc_hidden.shape[1])))
            dense = ctx @ sd[b + "crossattention.output.dense.weight"].t() + sd[b + "crossattention.output.dense.bias"]
            h = _ln(dense * masks("hidden", h.shape) + h,
                    sd[b + "crossattention.output.LayerNorm.weight"], sd[b + "crossattention.output.LayerNorm.bias"], eps)
        f = F.gelu(h @ sd[b + "intermediate.dense.weight"].t() + sd[b + "intermediate.dense.bias"])
        h = _ln((f @ sd[b + "output.dense.weight"].t() + sd[b + "output.dense.bias"]) * masks("hidden", h.shape) + h,
                sd[b + "output.LayerNorm.weight"], sd[b + "output.LayerNorm.bias"], eps)
    if p + "pooler.dense.weight" not in sd:      # add_pooling_layer=False (BLIP_SF)
        return h, None
    pooled = torch.tanh(h[:, 0] @ sd[p + "pooler.dense.weight"].t() + sd[p + "pooler.dense.bias"])
    return h, pooled


# ------------------------------------------------------------------------------------------------ BLIP_FF
def encode_multimodal_input(sd, ids, mask, images, vit_cfg, med_cfg, momentum=False):
    """blip_ff.py:82-116: pooler_output of BERT(text) cross-attending to ViT(image) tokens (masks unused there)"""
    sfx = "_m" if momentum else ""
    img = vit_forward(sd, images, vit_cfg, prefix=f"visual_encoder{sfx}.")
    return bert_forward(sd, ids, mask, img, med_cfg, prefix=f"text_encoder{sfx}.")[1]


def encode_multimodal_input_sf(sd, ids, mask, images, txt_mask, img_mask, vit_cfg, med_cfg, momentum=False):
    """blip_scorefusion/blip_sf.py:97-172: text_proj(BERT mode "text" [:,0]) * txt_mask + vision_proj(ViT [:,0]) * img_mask"""
    sfx = "_m" if momentum else ""
    tfeat = bert_forward(sd, ids, mask, None, med_cfg, prefix=f"text_encoder{sfx}.")[0][:, 0]
    temb = tfeat @ sd[f"text_proj{sfx}.weight"].t() + sd[f"text_proj{sfx}.bias"]
    ifeat = vit_forward(sd, images, vit_cfg, prefix=f"visual_encoder{sfx}.")[:, 0]
    iemb = ifeat @ sd[f"vision_proj{sfx}.weight"].t() + sd[f"vision_proj{sfx}.bias"]
    return temb * txt_mask.unsqueeze(-1) + iemb * img_mask.unsqueeze(-1)


def momentum_update(sd, m):
    for k in list(sd.keys()):
        for enc in ("visual_encoder", "text_encoder", "vision_proj", "text_proj"):
            if k.startswith(enc + "."):
                km = enc + "_m." + k[len(enc) + 1:]
                if km in sd and sd[km].dtype.is_floating_point:
                    sd[km] = sd[km] * m + sd[k].detach() * (1.0 - m)


def contrastive_loss(sd, state, batch, alpha, vit_cfg, med_cfg, momentum):
    """blip_ff.py:118-257.  `sd`: parameters (online ones may require grad; momentum entries are replaced in place),
    `state`: dict(query_queue [E,K], cand_queue [E,K], idx_queue [1,K] int64, ptr int).  With
    index_mapping["neg_cand_list"] ([b,N]) and batch["nc_dids_list"] the hard-negative variant (:127-131,159-170,
    196-205,233-246) runs; its enqueue choice draws torch.rand(1) from the global CPU generator like the reference."""
    with torch.no_grad():
        sd["temp"].clamp_(0.001, 0.5)
    temp = sd["temp"]
    ids, mask, img = batch["ids"], batch["mask"], batch["img"]
    im = batch["index_mapping"]
    hard = "neg_cand_list" in im
    if "tmask" in batch:     # BLIP_SF: score-level fusion of separately encoded text / image (same loss)
        enc = lambda mom: encode_multimodal_input_sf(sd, ids, mask, img, batch["tmask"], batch["imask"], vit_cfg, med_cfg, mom)
    else:
        enc = lambda mom: encode_multimodal_input(sd, ids, mask, img, vit_cfg, med_cfg, mom)
    emb = enc(False)
    qi = torch.tensor(im["query"]).flatten()
    pi = torch.tensor(im["pos_cand"]).flatten()
    q = F.normalize(emb[qi], dim=-1)
    p = F.normalize(emb[pi], dim=-1)
    bs, E = q.shape
    pc_idx = batch["p_did_list"].view(-1, 1)
    if hard:
        nc_idx = torch.as_tensor(batch["nc_dids_list"]).view(-1, 1)
        hn = nc_idx.size(0)
        idx_all = torch.cat([pc_idx.t(), nc_idx.t(), state["idx_queue"].clone()[:, hn:]], dim=1)
    else:
        idx_all = torch.cat([pc_idx.t(), state["idx_queue"].clone()], dim=1)
    pos = torch.eq(pc_idx, idx_all).float()
    tgt = pos / pos.sum(1, keepdim=True)
    with torch.no_grad():
        momentum_update(sd, momentum)
        emb_m = enc(True)
        q_m = F.normalize(emb_m[qi], dim=-1)
        p_m = F.normalize(emb_m[pi], dim=-1)
        q_m_all = torch.cat([q_m.t(), state["query_queue"].clone()], dim=1)
        if hard:
            nc_m = emb_m[torch.tensor(im["neg_cand_list"])]              # [b, N, E], NOT normalised (blip_ff.py:187-190)
            p_m_all = torch.cat([p_m.t(), nc_m.view(hn, E).t(), state["cand_queue"].clone()[:, hn:]], dim=1)
        else:
            p_m_all = torch.cat([p_m.t(), state["cand_queue"].clone()], dim=1)
        t_q2p = alpha * F.softmax(q_m @ p_m_all / temp, dim=1) + (1 - alpha) * tgt
        t_p2q = alpha * F.softmax(p_m @ q_m_all / temp, dim=1) + (1 - alpha) * tgt
    sim_q2p = q @ p_m_all / temp
    sim_p2q = p @ q_m_all / temp
    loss = (-(F.log_softmax(sim_q2p, dim=1) * t_q2p).sum(1).mean() - (F.log_softmax(sim_p2q, dim=1) * t_p2q).sum(1).mean()) / 2

    def enqueue(qf, cf, idxs):   # _dequeue_and_enqueue (world size 1)
        with torch.no_grad():
            b, K, ptr = qf.shape[0], state["query_queue"].shape[1], state["ptr"]
            assert K % b == 0
            state["query_queue"][:, ptr:ptr + b] = qf.t()
            state["cand_queue"][:, ptr:ptr + b] = cf.t()
            state["idx_queue"][:, ptr:ptr + b] = idxs.view(1, -1)
            state["ptr"] = (ptr + b) % K

    if hard:
        if torch.rand(1) < 0.5:
            enqueue(q_m, p_m, pc_idx)
        else:
            enqueue(q_m, nc_m[:, 0, :].contiguous(), nc_idx.view(bs, -1)[:, 0].contiguous())
    else:
        enqueue(q_m, p_m, pc_idx)
    acc = pos.gather(1, sim_q2p.max(1)[1].unsqueeze(1)).squeeze().mean()
    return {"loss": loss, "accuracy": acc, "sim_q2p": sim_q2p}
