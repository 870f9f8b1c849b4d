"""CLIP_FF on the MI355X: the T5 fusion stack against the reference-generated golden G13 (reference
CLIPFeatureFusion.encode_multimodal_input + loss with stub encoders), and the whole model (towers without pooling ->
fusion -> loss -> backward -> AdamW with the separate T5 group) against the oracle.  bf16 GEMMs: relative L2 2e-2 forward
and 3e-2 on the gradients next to the output; deeper gradients pass through ReLU'(x) of bf16-rounded pre-activations (a
sign flip on |x| < ~1e-2 is a discrete error, ~0.4 % of the units at unit variance) and un-scaled, peaky T5 softmaxes, so
they are held to direction (cosine > 0.985, measured 0.987-0.999) plus relative L2 < 0.2."""
import json
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "uniir_amd", "src"))
G = os.path.join(ROOT, "tests", "golden")


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu().flatten(), torch.as_tensor(b).detach().double().cpu().flatten()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def cos(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu().flatten(), torch.as_tensor(b).detach().double().cpu().flatten()
    return torch.nn.functional.cosine_similarity(a, b, dim=0).item()


def deep_ok(a, b, what):
    assert cos(a, b) > 0.985 and rel(a, b) < 0.2, (what, cos(a, b), rel(a, b))


def _model(cfg, t5_cfg, seed=0, dropout_rate=0.0):
    from oracle import clip_oracle as O
    from models.uniir_clip.clip_featurefusion.clip_ff import CLIPFeatureFusion
    from uniir_amd import clip_model
    clip_model.CLIP_CONFIGS["tiny-ff"] = cfg
    config = SimpleNamespace(model=SimpleNamespace(gather_embeddings=False), data_config=SimpleNamespace(in_batch_neg_num=0))
    m = CLIPFeatureFusion("tiny-ff", device="cuda", config=config,
                          t5_config=dict(d_model=t5_cfg["d_model"], num_heads=t5_cfg["num_heads"], d_ff=t5_cfg["d_ff"],
                                         num_layers=t5_cfg["num_layers"], dropout_rate=dropout_rate))
    sd = O.init_state_dict(cfg, seed=seed)
    sd.pop("text_projection")
    m.clip_model.load_state_dict(sd, strict=True)
    return m, sd


def test_g13_fusion_stack_matches_reference_golden():
    from oracle import clip_oracle as O
    from uniir_amd import clipff_model as FM
    from uniir_amd.losses import InBatchNCEFn
    z = np.load(os.path.join(G, "g13_clipff.npz"))
    t5_cfg = json.loads(str(z["cfg"]))
    cfg = O.tiny_config(embed_dim=t5_cfg["d_model"], transformer_width=t5_cfg["d_model"], transformer_heads=2)
    m, _ = _model(cfg, t5_cfg)
    m.t5_layers.load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}, strict=True)
    st = m._ensure_t5()
    m.zero_grad()
    txt, img = torch.from_numpy(z["txt_feat"]).cuda(), torch.from_numpy(z["img_feat"]).cuda()
    M, Tt, D = txt.shape
    Ti = img.shape[1]
    x = torch.cat([txt, img], dim=1).view(M * (Tt + Ti), D).contiguous()
    pooled, stash = FM.t5_forward(st, "", x, M, Tt + Ti, m.t5_heads, m.t5_layers_n, True)
    assert rel(pooled, z["emb"]) < 2e-2, rel(pooled, z["emb"])
    emb = pooled.detach().requires_grad_(True)
    b = M // 2
    iq = torch.tensor([2 * i for i in range(b)], dtype=torch.int32, device="cuda")
    ip = torch.tensor([2 * i + 1 for i in range(b)], dtype=torch.int32, device="cuda")
    loss, acc, _ = InBatchNCEFn.apply(emb, iq, ip, torch.tensor(1 / 0.07, device="cuda"), False)
    loss.backward()
    assert abs(loss.item() - float(z["loss"])) < 3e-2 and acc.item() == float(z["acc"])
    dx = FM.t5_backward(st, "", emb.grad, stash, m.t5_heads, m.t5_layers_n).view(M, Tt + Ti, D)
    deep_ok(dx[:, :Tt], z["dtxt"], "dtxt")
    deep_ok(dx[:, Tt:], z["dimg"], "dimg")
    deep_ok(st.grad_view("block.0.layer.0.SelfAttention.q.weight"), z["g_q0"], "q0")
    deep_ok(st.grad_view("block.0.layer.0.SelfAttention.relative_attention_bias.weight"), z["g_rel"], "rel bias")


def test_clipff_model_matches_oracle_and_trains():
    from oracle import clip_oracle as O
    from oracle import clipff_oracle as FF
    from uniir_amd.trainer import NativeAdamW
    t5_cfg = dict(d_model=128, num_heads=2, d_ff=256, num_layers=2, d_kv=64)
    cfg = O.tiny_config(embed_dim=128, transformer_width=128, transformer_heads=2)
    m, sd = _model(cfg, t5_cfg, seed=5)
    pairs = 4
    batch = O.synthetic_batch(cfg, pairs, seed=23)
    dbatch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    # oracle on the same weights
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    t5o = {n: p.detach().cpu().clone().requires_grad_(True) for n, p in m.t5_layers.named_parameters()}
    emb_o = FF.encode_multimodal_input(sdo, t5o, cfg, t5_cfg, batch["txt_batched"], batch["image_batched"])
    out_o = O.inbatch_contrastive_loss(emb_o, batch["index_mapping"], sdo["logit_scale"].exp())
    out_o["loss"].backward()
    # device
    opt = NativeAdamW(m.clip_model, lr=1e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.2, allreduce=False,
                      extra=[m.t5_optimizer_group(lr=1e-3)])
    m.train()
    opt.zero_grad()
    with torch.no_grad():
        emb_d = m.encode_multimodal_input(dbatch["txt_batched"], dbatch["image_batched"])
    assert rel(emb_d, emb_o) < 2e-2, rel(emb_d, emb_o)
    out_d = m(dbatch)
    out_d["loss"].backward()
    assert abs(out_d["loss"].item() - out_o["loss"].item()) < 3e-2 * max(1.0, abs(out_o["loss"].item()))
    for name in ("block.0.layer.0.SelfAttention.q.weight", "block.1.layer.1.DenseReluDense.wo.weight",
                 "block.0.layer.0.SelfAttention.relative_attention_bias.weight", "final_layer_norm.weight"):
        deep_ok(m.t5_layers.get_parameter(name).grad, t5o[name].grad, name)
    for name in ("visual.proj", "visual.conv1.weight", "token_embedding.weight", "ln_final.weight",
                 "visual.transformer.resblocks.0.attn.in_proj_weight"):
        deep_ok(m.clip_model.get_parameter(name).grad, sdo[name].grad, name)
    w0 = m.t5_layers.get_parameter("block.0.layer.0.SelfAttention.q.weight").detach().clone()
    c0 = m.clip_model.visual.proj.detach().clone()
    opt.step()
    assert (m.t5_layers.get_parameter("block.0.layer.0.SelfAttention.q.weight") - w0).abs().max().item() > 1e-4
    assert (m.clip_model.visual.proj - c0).abs().max().item() > 1e-6
    with torch.no_grad():
        dbatch["did_list"] = list(range(2 * pairs))
        emb, ids = m(dbatch, encode_mbeir_batch=True)
    assert emb.shape == (2 * pairs, 128) and torch.isfinite(emb).all()


def test_t5_train_mode_dropout_matches_masked_oracle():
    """the six T5 dropout sites with exported counter-based masks fed to the oracle's hooks: forward, input gradient and
    weight gradients; plus the module switch (eval deterministic and mask-free, train mode stochastic)"""
    from oracle import clip_oracle as O
    from oracle import clipff_oracle as FF
    from uniir_amd import clipff_model as FM
    from uniir_amd import ops
    z = np.load(os.path.join(G, "g13_clipff.npz"))
    t5_cfg = json.loads(str(z["cfg"]))
    cfg = O.tiny_config(embed_dim=t5_cfg["d_model"], transformer_width=t5_cfg["d_model"], transformer_heads=2)
    m, _ = _model(cfg, t5_cfg, dropout_rate=0.1)
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    m.t5_layers.load_state_dict(sd, strict=True)
    st = m._ensure_t5()
    m.zero_grad()
    txt, img = torch.from_numpy(z["txt_feat"]), torch.from_numpy(z["img_feat"])
    M, Tt, D = txt.shape
    T = Tt + img.shape[1]
    x = torch.cat([txt, img], dim=1).view(M * T, D).contiguous().cuda()
    p = 0.1
    torch.manual_seed(11)
    pooled, stash = FM.t5_forward(st, "", x.clone(), M, T, m.t5_heads, m.t5_layers_n, True, drop=ops.DropSeeds(), p=p)
    w = torch.randn(pooled.shape, generator=torch.Generator().manual_seed(2))
    dx = FM.t5_backward(st, "", w.cuda(), stash, m.t5_heads, m.t5_layers_n).view(M, T, D)
    torch.manual_seed(11)
    seeds = ops.DropSeeds()

    def masks(kind, shape):
        buf = torch.empty(int(np.prod(shape)), device="cuda")
        ops.call("uniir_dropout_mask", buf, buf.numel(), p, seeds.next())
        return buf.view(*shape).cpu()

    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xo = torch.cat([txt, img], dim=1).clone().requires_grad_(True)
    ref = FF.t5_stack(sdg, xo, t5_cfg, masks=masks).mean(dim=1)
    assert rel(pooled, ref) < 2e-2, rel(pooled, ref)
    assert rel(pooled, z["emb"]) > 5e-2
    (ref * w).sum().backward()
    deep_ok(dx, xo.grad, "dx")
    for name in ("block.0.layer.0.SelfAttention.q.weight", "block.0.layer.0.SelfAttention.o.weight",
                 "block.1.layer.1.DenseReluDense.wi.weight", "block.1.layer.1.DenseReluDense.wo.weight",
                 "block.0.layer.0.SelfAttention.relative_attention_bias.weight", "final_layer_norm.weight"):
        deep_ok(st.grad_view(name), sdg[name].grad, name)
    # module switch
    batch = O.synthetic_batch(cfg, 2, seed=3)
    t, im = batch["txt_batched"].cuda(), batch["image_batched"].cuda()
    with torch.no_grad():
        m.eval()
        e0, e1 = m.encode_multimodal_input(t, im), m.encode_multimodal_input(t, im)
        m.train()
        torch.manual_seed(1)
        d0 = m.encode_multimodal_input(t, im)
        torch.manual_seed(1)
        d1 = m.encode_multimodal_input(t, im)
        d2 = m.encode_multimodal_input(t, im)
    assert torch.equal(e0, e1) and torch.equal(d0, d1) and not torch.equal(d0, d2) and rel(d0, e0) > 1e-2


def test_clipff_save_resume_then_step(tmp_path):
    """a CLIP_FF checkpoint maps to the CPU on load (host_utils.load_checkpoint_file); the train.py mirror moves
    only top-level tensors of the optimizer state back, so the T5 group's moments ("extra": a list of (m, v) tuples) arrive on
    the CPU -- the first optimizer step after a resume must move them, and the resumed run must continue exactly like the
    uninterrupted one"""
    from oracle import clip_oracle as O
    from uniir_amd.host_utils import load_checkpoint_file
    from uniir_amd.trainer import NativeAdamW
    t5_cfg = dict(d_model=128, num_heads=2, d_ff=256, num_layers=2, d_kv=64)
    cfg = O.tiny_config(embed_dim=128, transformer_width=128, transformer_heads=2)
    batch = O.synthetic_batch(cfg, 4, seed=23)
    dbatch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}

    def make():
        m, _ = _model(cfg, t5_cfg, seed=5)
        m.eval()                                    # no dropout: both runs must see identical arithmetic
        opt = NativeAdamW(m.clip_model, lr=1e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.2, allreduce=False,
                          extra=[m.t5_optimizer_group(lr=1e-3)])
        return m, opt

    def step(m, opt):
        opt.zero_grad()
        out = m(dbatch)
        out["loss"].backward()
        opt.step()
        return out["loss"].item()

    m1, o1 = make()
    step(m1, o1)
    path = str(tmp_path / "clip_ff_epoch_0.pth")
    torch.save({"model": m1.state_dict(), "optimizer": o1.state_dict()}, path)
    l_ref = step(m1, o1)                            # the uninterrupted second step
    ck = load_checkpoint_file(path)                 # everything on the CPU now
    assert all(not t.is_cuda for mv in ck["optimizer"]["extra"] for t in mv)
    m2, o2 = make()
    m2.load_state_dict(ck["model"])
    o2.load_state_dict({k: (v.to("cuda") if isinstance(v, torch.Tensor) else v) for k, v in ck["optimizer"].items()})
    l_res = step(m2, o2)                            # used to raise "uniir_amd ops need device tensors"
    assert abs(l_res - l_ref) < 1e-5 * max(1.0, abs(l_ref))
    q1 = m1.t5_layers.get_parameter("block.0.layer.0.SelfAttention.q.weight")
    q2 = m2.t5_layers.get_parameter("block.0.layer.0.SelfAttention.q.weight")
    # same state, same batch -> the same update up to the order of the fp32 atomics in the weight-gradient GEMMs
    assert (q1 - q2).abs().max().item() < 1e-5 and (m1.clip_model.visual.proj - m2.clip_model.visual.proj).abs().max().item() < 1e-5


def test_clipff_vit_l14_two_pairs_against_the_oracle():
    """CLIP_FF at its large architecture (clip_ff.py:161-192): ViT-L/14 towers without pooling (257 image + 77 text tokens, all
    projected to 768), 2-layer T5 fusion with 12 heads and relative position bias over the 334 tokens, mean pooling, InfoNCE --
    2 pairs, eval mode (no dropout), against oracle/clipff_oracle.py: embeddings, loss, accuracy, gradients of both stacks"""
    from oracle import clip_oracle as O
    from oracle import clipff_oracle as FF
    from models.uniir_clip.clip_featurefusion.clip_ff import CLIPFeatureFusion
    from uniir_amd.clip_model import CLIP_CONFIGS
    cfg = CLIP_CONFIGS["ViT-L/14"]
    t5_cfg = dict(d_model=768, num_heads=12, d_ff=2048, num_layers=2, d_kv=64)
    config = SimpleNamespace(model=SimpleNamespace(gather_embeddings=False), data_config=SimpleNamespace(in_batch_neg_num=0))
    m = CLIPFeatureFusion("ViT-L/14", device="cuda", config=config)
    sd = O.init_state_dict(cfg, seed=11)
    sd.pop("text_projection")
    m.clip_model.load_state_dict(sd, strict=True)
    m.eval()
    pairs = 2
    batch = O.synthetic_batch(cfg, pairs, seed=31)
    dbatch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    t5o = {n: p.detach().cpu().clone().requires_grad_(True) for n, p in m.t5_layers.named_parameters()}
    emb_o = FF.encode_multimodal_input(sdo, t5o, cfg, t5_cfg, batch["txt_batched"], batch["image_batched"])
    out_o = O.inbatch_contrastive_loss(emb_o, batch["index_mapping"], sdo["logit_scale"].exp())
    out_o["loss"].backward()
    m.clip_model._ensure_flat()
    m._ensure_t5()
    m.zero_grad()
    with torch.no_grad():
        emb_d = m.encode_multimodal_input(dbatch["txt_batched"], dbatch["image_batched"])
    print("OBS clipff-L emb rel", rel(emb_d, emb_o))
    assert rel(emb_d, emb_o) < 1.5e-2, rel(emb_d, emb_o)         # observed 7.7e-3
    out_d = m(dbatch)
    out_d["loss"].backward()
    print("OBS clipff-L loss", out_d["loss"].item(), out_o["loss"].item())
    assert abs(out_d["loss"].item() - out_o["loss"].item()) < 4e-3 * max(1.0, abs(out_o["loss"].item()))     # observed 7.5e-4
    assert out_d["accuracy"].item() == out_o["accuracy"].item()
    for name in ("block.0.layer.0.SelfAttention.q.weight", "block.1.layer.1.DenseReluDense.wo.weight",
                 "block.0.layer.0.SelfAttention.relative_attention_bias.weight", "final_layer_norm.weight"):
        deep_ok(m.t5_layers.get_parameter(name).grad, t5o[name].grad, name)
    for name in ("visual.proj", "visual.conv1.weight", "token_embedding.weight", "ln_final.weight",
                 "visual.transformer.resblocks.0.attn.in_proj_weight", "visual.transformer.resblocks.23.mlp.c_proj.weight",
                 "transformer.resblocks.11.attn.out_proj.weight"):
        deep_ok(m.clip_model.get_parameter(name).grad, sdo[name].grad, name)
