"""BLIP_FF on the MI355X against the reference-generated goldens (G6 MED BERT, G7 BLIP ViT, G8 two BLIP_FF training
steps) and the oracle.  The device path computes its GEMMs in bf16 (fp32 accumulate), so the comparisons are relative
L2 errors: 2e-2 for forward tensors, 5e-2 for gradients (the reference's own fp16-autocast run is no closer to fp32);
index / bookkeeping results (queue ids, pointer, accuracy) are exact."""
import json
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu().flatten(), torch.as_tensor(b).double().cpu().flatten()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def tiny_model(med_cfg, vit_cfg, queue_size=16, momentum=0.9):
    from uniir_amd.blip_model import BLIPFeatureFusion
    return BLIPFeatureFusion(med_config=med_cfg, vit_config=vit_cfg, embed_dim=med_cfg["hidden_size"],
                             queue_size=queue_size, momentum=momentum,
                             config=types.SimpleNamespace(tokenizer_max_length=20))


def load_sub(model, z, tag, prefix):
    sd = {prefix + k[len(tag):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(tag)}
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys, missing.unexpected_keys
    return sd


def test_g7_vit_forward_backward():
    from uniir_amd import blip_model as bm
    z = np.load(os.path.join(G, "g7_vit.npz"))
    vit_cfg = json.loads(str(z["cfg"]))
    med_cfg = dict(hidden_size=128, intermediate_size=256, num_attention_heads=2, num_hidden_layers=1, vocab_size=64,
                   max_position_embeddings=32)
    model = tiny_model(med_cfg, vit_cfg).cuda()
    load_sub(model, z, "sd::", "visual_encoder.")
    model._sync()
    model.zero_grad()
    st = model._online
    x = torch.from_numpy(z["x"]).cuda()
    tok, T, stash = bm.vit_forward(st, model._conv16, "visual_encoder.", model.vit_cfg, model.image_size, x, True)
    y = torch.from_numpy(z["y"])
    assert rel(tok.float().view(y.shape), y) < 2e-2
    w = torch.from_numpy(z["w"]).cuda().view(-1, y.shape[-1]).contiguous()
    bm.vit_backward(st, model._dconv, "visual_encoder.", model.vit_cfg, w, stash)
    for k in z.files:
        if k.startswith("grad::"):
            g = st.grad_view("visual_encoder." + k[6:])
            assert rel(g, z[k]) < 5e-2, (k, rel(g, z[k]))


def test_g6_med_bert_forward_backward():
    from uniir_amd import blip_model as bm
    z = np.load(os.path.join(G, "g6_med.npz"))
    med_cfg = json.loads(str(z["cfg"]))
    vit_cfg = dict(img_size=32, patch_size=16, embed_dim=med_cfg["encoder_width"], depth=1, num_heads=2)
    model = tiny_model(med_cfg, vit_cfg).cuda()
    load_sub(model, z, "sd::", "text_encoder.")
    model._sync()
    model.zero_grad()
    st = model._online
    ids = torch.from_numpy(z["ids"]).to(torch.int32).cuda()
    key_len = torch.from_numpy(z["mask"]).sum(1).to(torch.int32).cuda()
    img = torch.from_numpy(z["img"])
    n, Ti, Ew = img.shape
    img16 = img.cuda().to(torch.bfloat16).view(n * Ti, Ew).contiguous()
    pooled, stash = bm.bert_forward(st, "text_encoder.", model.med_cfg, ids, key_len, img16, Ti, True)
    assert rel(pooled, z["pooler_output"]) < 2e-2
    dimg = bm.bert_backward(st, "text_encoder.", model.med_cfg, torch.from_numpy(z["w"]).cuda(), stash)
    assert rel(dimg.view(n, Ti, Ew), z["dimg"]) < 5e-2
    for k in z.files:
        if k.startswith("grad::"):
            g = st.grad_view("text_encoder." + k[6:])
            assert rel(g, z[k]) < 5e-2, (k, rel(g, z[k]))


def test_g8_blip_ff_two_training_steps():
    z = np.load(os.path.join(G, "g8_blipff.npz"))
    med_cfg, vit_cfg = json.loads(str(z["med_cfg"])), json.loads(str(z["vit_cfg"]))
    model = tiny_model(med_cfg, vit_cfg, queue_size=int(z["queue_size"]), momentum=float(z["momentum"]))
    load_sub(model, z, "sd0::", "")
    model.copy_params()
    model = model.cuda()
    model.train()
    for step in range(2):
        b = len(z[f"s{step}_pdid"])
        batch = {
            "txt_batched": types.SimpleNamespace(input_ids=torch.from_numpy(z[f"s{step}_ids"]).cuda(),
                                                 attention_mask=torch.from_numpy(z[f"s{step}_mask"]).cuda()),
            "image_batched": torch.from_numpy(z[f"s{step}_img"]).cuda(),
            "txt_mask_batched": torch.ones(2 * b, dtype=torch.long).cuda(),
            "image_mask_batched": torch.ones(2 * b, dtype=torch.long).cuda(),
            "p_did_list": torch.from_numpy(z[f"s{step}_pdid"]),
            "index_mapping": {"query": [[2 * i] for i in range(b)], "pos_cand": [[2 * i + 1] for i in range(b)]},
        }
        model.zero_grad()
        out = model(batch, alpha=float(z[f"s{step}_alpha"]))
        out["loss"].backward()
        assert abs(out["loss"].item() - float(z[f"s{step}_loss"])) < 2e-2, (out["loss"].item(), float(z[f"s{step}_loss"]))
        assert out["accuracy"].item() == float(z[f"s{step}_acc"])
        assert rel(model.query_queue, z[f"s{step}_query_queue"]) < 2e-2
        assert rel(model.cand_queue, z[f"s{step}_cand_queue"]) < 2e-2
        assert np.array_equal(model.idx_queue.cpu().numpy(), z[f"s{step}_idx_queue"])
        assert int(model.new_ptr_queue.item()) == int(z[f"s{step}_ptr"].item())
        assert rel(model.get_parameter("visual_encoder_m.blocks.0.attn.qkv.weight"), z[f"s{step}_m_vit_qkv0"]) < 1e-5
        assert rel(model.temp.grad, z[f"s{step}_dtemp"]) < 8e-2
        for name, key in (("visual_encoder.blocks.0.attn.qkv.weight", "g_vit_qkv0"),
                          ("text_encoder.encoder.layer.0.attention.self.query.weight", "g_txt_q0"),
                          ("text_encoder.pooler.dense.weight", "g_pool")):
            r = rel(model.get_parameter(name).grad, z[f"s{step}_{key}"])
            assert r < 8e-2, (step, name, r)
        with torch.no_grad():   # the fixture's SGD nudge between its two steps
            for n, p in model._online_params():
                if n != "temp":
                    p.add_(-0.05 * p.grad)


def test_blip_ff_native_adamw_and_embedding_path():
    """optimizer step through NativeAdamW (one weight-decay group), then the embedding extraction entry point"""
    from uniir_amd.trainer import NativeAdamW
    z = np.load(os.path.join(G, "g8_blipff.npz"))
    med_cfg, vit_cfg = json.loads(str(z["med_cfg"])), json.loads(str(z["vit_cfg"]))
    model = tiny_model(med_cfg, vit_cfg).cuda()
    opt = NativeAdamW(model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05, allreduce=False)
    b = 4
    batch = {
        "txt_batched": types.SimpleNamespace(input_ids=torch.from_numpy(z["s0_ids"]).cuda(),
                                             attention_mask=torch.from_numpy(z["s0_mask"]).cuda()),
        "image_batched": torch.from_numpy(z["s0_img"]).cuda(),
        "p_did_list": torch.from_numpy(z["s0_pdid"]),
        "index_mapping": {"query": [[2 * i] for i in range(b)], "pos_cand": [[2 * i + 1] for i in range(b)]},
        "did_list": list(range(2 * b)),
    }
    w0 = model.get_parameter("text_encoder.pooler.dense.weight").detach().clone()
    losses = []
    for _ in range(3):
        opt.zero_grad()
        out = model(batch, alpha=0.4)
        out["loss"].backward()
        g = model.get_parameter("text_encoder.pooler.dense.weight").grad.clone()
        opt.step()
        losses.append(out["loss"].item())
    assert all(np.isfinite(losses))
    w1 = model.get_parameter("text_encoder.pooler.dense.weight").detach()
    assert (w1 - w0).abs().max().item() > 1e-4
    assert g.abs().max().item() > 0
    with torch.no_grad():
        emb, ids = model(batch, encode_mbeir_batch=True)
    assert emb.shape == (2 * b, med_cfg["hidden_size"]) and ids == list(range(2 * b))
    assert torch.isfinite(emb).all() and emb.abs().max().item() <= 1.0


def test_g8h_blip_ff_hard_negatives():
    """two training steps with one hard negative per query against the reference golden (G8h): loss, accuracy, the
    [p | negatives | queue] id row, the coin-flip enqueue (same host generator seed as the fixture)"""
    z = np.load(os.path.join(G, "g8h_blipff_hardneg.npz"))
    med_cfg, vit_cfg = json.loads(str(z["med_cfg"])), json.loads(str(z["vit_cfg"]))
    model = tiny_model(med_cfg, vit_cfg, queue_size=int(z["queue_size"]), momentum=float(z["momentum"]))
    load_sub(model, z, "sd0::", "")
    model.copy_params()
    model = model.cuda()
    model.train()
    for step in range(2):
        b = len(z[f"s{step}_pdid"])
        M = 3 * b
        batch = {
            "txt_batched": types.SimpleNamespace(input_ids=torch.from_numpy(z[f"s{step}_ids"]).cuda(),
                                                 attention_mask=torch.from_numpy(z[f"s{step}_mask"]).cuda()),
            "image_batched": torch.from_numpy(z[f"s{step}_img"]).cuda(),
            "txt_mask_batched": torch.ones(M, dtype=torch.long).cuda(),
            "image_mask_batched": torch.ones(M, dtype=torch.long).cuda(),
            "p_did_list": torch.from_numpy(z[f"s{step}_pdid"]),
            "nc_dids_list": torch.from_numpy(z[f"s{step}_ncdid"]),
            "index_mapping": {"query": [[3 * i] for i in range(b)], "pos_cand": [[3 * i + 1] for i in range(b)],
                              "neg_cand_list": [[3 * i + 2] for i in range(b)]},
        }
        model.zero_grad()
        torch.manual_seed(1000 + step)
        out = model(batch, alpha=float(z[f"s{step}_alpha"]))
        out["loss"].backward()
        want = float(z[f"s{step}_loss"])
        assert abs(out["loss"].item() - want) < 2e-2 * max(1.0, abs(want)), (out["loss"].item(), want)
        assert out["accuracy"].item() == float(z[f"s{step}_acc"])
        assert rel(model.query_queue, z[f"s{step}_query_queue"]) < 2e-2
        assert rel(model.cand_queue, z[f"s{step}_cand_queue"]) < 2e-2
        assert np.array_equal(model.idx_queue.cpu().numpy(), z[f"s{step}_idx_queue"])
        assert int(model.new_ptr_queue.item()) == int(z[f"s{step}_ptr"].item())
        r = rel(model.get_parameter("text_encoder.pooler.dense.weight").grad, z[f"s{step}_g_pool"])
        assert r < 8e-2, (step, r)
        with torch.no_grad():
            for n, p in model._online_params():
                if n != "temp":
                    p.add_(-0.05 * p.grad)


def test_g8s_blip_sf_two_training_steps():
    """BLIPScoreFusion against the reference golden G8s: projections, masked sum (one image-only and one text-only item),
    momentum of the projection heads, frozen cross-attention untouched by AdamW"""
    from uniir_amd.blip_model import BLIPScoreFusion
    from uniir_amd.trainer import NativeAdamW
    z = np.load(os.path.join(G, "g8s_blipsf.npz"))
    med_cfg, vit_cfg = json.loads(str(z["med_cfg"])), json.loads(str(z["vit_cfg"]))
    model = BLIPScoreFusion(med_config=med_cfg, vit_config=vit_cfg, embed_dim=int(z["embed_dim"]), queue_size=int(z["queue_size"]),
                            momentum=float(z["momentum"]), config=types.SimpleNamespace(tokenizer_max_length=20))
    load_sub(model, z, "sd0::", "")
    model.copy_params()
    model = model.cuda()
    model.train()
    frozen = "text_encoder.encoder.layer.0.crossattention.self.query.weight"
    assert not model.get_parameter(frozen).requires_grad
    for step in range(2):
        b = len(z[f"s{step}_pdid"])
        batch = {
            "txt_batched": types.SimpleNamespace(input_ids=torch.from_numpy(z[f"s{step}_ids"]).cuda(),
                                                 attention_mask=torch.from_numpy(z[f"s{step}_mask"]).cuda()),
            "image_batched": torch.from_numpy(z[f"s{step}_img"]).cuda(),
            "txt_mask_batched": torch.from_numpy(z[f"s{step}_tmask"]).cuda(),
            "image_mask_batched": torch.from_numpy(z[f"s{step}_imask"]).cuda(),
            "p_did_list": torch.from_numpy(z[f"s{step}_pdid"]),
            "index_mapping": {"query": [[2 * i] for i in range(b)], "pos_cand": [[2 * i + 1] for i in range(b)]},
        }
        model.zero_grad()
        out = model(batch, alpha=float(z[f"s{step}_alpha"]))
        out["loss"].backward()
        assert abs(out["loss"].item() - float(z[f"s{step}_loss"])) < 2e-2, (out["loss"].item(), float(z[f"s{step}_loss"]))
        assert out["accuracy"].item() == float(z[f"s{step}_acc"])
        assert rel(model.query_queue, z[f"s{step}_query_queue"]) < 2e-2
        assert np.array_equal(model.idx_queue.cpu().numpy(), z[f"s{step}_idx_queue"])
        # (step 1 averages weights that were nudged with the device gradients: bf16-level differences)
        assert rel(model.get_parameter("vision_proj_m.weight"), z[f"s{step}_m_vproj"]) < 2e-4
        for name, key in (("visual_encoder.blocks.0.attn.qkv.weight", "g_vit_qkv0"),
                          ("text_encoder.encoder.layer.0.attention.self.query.weight", "g_txt_q0"),
                          ("vision_proj.weight", "g_vproj"), ("text_proj.bias", "g_tprojb")):
            r = rel(model.get_parameter(name).grad, z[f"s{step}_{key}"])
            assert r < 8e-2, (step, name, r)
        with torch.no_grad():   # the fixture nudges every parameter that received a gradient (temp included)
            for n, p in model._online_params():
                if n not in model._frozen:
                    p.add_(-0.05 * p.grad)
    # the optimizer must leave the frozen cross-attention (and its weight decay) alone
    opt = NativeAdamW(model, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.5, allreduce=False)
    w_frozen = model.get_parameter(frozen).detach().clone()
    w_proj = model.get_parameter("text_proj.weight").detach().clone()
    t0 = model.temp.detach().clone()
    opt.step()
    assert torch.equal(model.get_parameter(frozen).detach(), w_frozen)
    assert (model.get_parameter("text_proj.weight").detach() - w_proj).abs().max().item() > 1e-4
    assert model.temp.item() != t0.item()


def test_blip_ff_train_mode_dropout_matches_masked_oracle():
    """train mode with BERT hidden / attention dropout and ViT DropPath: the counter-based masks are exported
    (uniir_dropout_mask, same seeds) and fed to the oracle's mask hooks -- forward and gradients must agree like the
    dropout-free runs do; eval mode is deterministic and mask-free"""
    from oracle import blip_oracle as bo
    from uniir_amd import ops
    z = np.load(os.path.join(G, "g8_blipff.npz"))
    med_cfg, vit_cfg = json.loads(str(z["med_cfg"])), json.loads(str(z["vit_cfg"]))
    med_cfg.update(hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.2)
    vit_cfg["drop_path_rate"] = 0.5
    model = tiny_model(med_cfg, vit_cfg)
    sd = load_sub(model, z, "sd0::", "")
    model = model.cuda()
    txt = types.SimpleNamespace(input_ids=torch.from_numpy(z["s0_ids"]).cuda(), attention_mask=torch.from_numpy(z["s0_mask"]).cuda())
    img = torch.from_numpy(z["s0_img"]).cuda()
    M = img.shape[0]
    model.eval()
    with torch.no_grad():
        e0, e1 = model.encode_multimodal_input(txt, img), model.encode_multimodal_input(txt, img)
    assert torch.equal(e0, e1)
    ref_eval = bo.encode_multimodal_input({k: v for k, v in sd.items()}, torch.from_numpy(z["s0_ids"]),
                                          torch.from_numpy(z["s0_mask"]), torch.from_numpy(z["s0_img"]), vit_cfg, med_cfg)
    assert rel(e0, ref_eval) < 2e-2
    model.train()
    model.zero_grad()
    torch.manual_seed(5)
    emb = model.encode_multimodal_input(txt, img)
    assert rel(emb, ref_eval) > 5e-2          # the masks did something
    w = torch.randn(emb.shape, generator=torch.Generator().manual_seed(1))
    (emb * w.cuda()).sum().backward()
    with torch.no_grad():
        torch.manual_seed(6)
        assert not torch.equal(model.encode_multimodal_input(txt, img), emb.detach())
    # the same draws, replayed for the oracle
    torch.manual_seed(5)
    seeds = ops.DropSeeds()
    depth = vit_cfg["depth"]
    keep = 1.0 - torch.linspace(0, 0.5, depth).view(depth, 1, 1)
    rowscale = torch.floor(keep + torch.rand(depth, 2, M)) / keep
    path_iter = iter([rowscale[i, j] for i in range(depth) for j in range(2)])

    def masks(kind, shape):
        if kind == "path":
            return next(path_iter).view(shape)
        p = med_cfg["hidden_dropout_prob"] if kind == "hidden" else med_cfg["attention_probs_dropout_prob"]
        buf = torch.empty(int(np.prod(shape)), device="cuda")
        ops.call("uniir_dropout_mask", buf, buf.numel(), p, seeds.next())
        return buf.view(*shape).cpu()

    sdg = {k: v.clone().requires_grad_(v.dtype == torch.float32) for k, v in sd.items()}
    tok = bo.vit_forward(sdg, torch.from_numpy(z["s0_img"]), vit_cfg, prefix="visual_encoder.", masks=masks)
    ref = bo.bert_forward(sdg, torch.from_numpy(z["s0_ids"]), torch.from_numpy(z["s0_mask"]), tok, med_cfg,
                          prefix="text_encoder.", masks=masks)[1]
    assert rel(emb, ref) < 2e-2, rel(emb, ref)
    (ref * w).sum().backward()
    for name in ("visual_encoder.blocks.0.attn.qkv.weight", "visual_encoder.blocks.0.attn.proj.bias",
                 "visual_encoder.blocks.1.mlp.fc2.bias", "visual_encoder.pos_embed",
                 "text_encoder.embeddings.word_embeddings.weight", "text_encoder.encoder.layer.0.attention.self.query.weight",
                 "text_encoder.encoder.layer.0.crossattention.self.key.weight",
                 "text_encoder.encoder.layer.0.attention.output.dense.bias",
                 "text_encoder.encoder.layer.1.output.dense.weight", "text_encoder.pooler.dense.weight"):
        r = rel(model.get_parameter(name).grad, sdg[name].grad)
        assert r < 8e-2, (name, r)


def test_blip_ff_large_two_pairs_against_the_oracle():
    """BASELINE configs[4] at its real architecture (blip_ff.py:82-257): ViT-L/16 @224 (197 tokens, 16 heads, width 1024),
    MED BERT-base with 100 text tokens cross-attending to the 1024-wide image tokens (K / V projections 1024 -> 768), padding
    masks of different lengths, momentum encoders, queue -- 2 pairs, eval mode (no dropout), against oracle/blip_oracle.py:
    embeddings, loss, accuracy, d temp, parameter gradients"""
    from oracle import blip_oracle as bo
    from uniir_amd.blip_model import BLIPFeatureFusion
    torch.manual_seed(0)
    L, pairs, K = 100, 2, 16
    model = BLIPFeatureFusion(med_config={}, vit="large", queue_size=K, momentum=0.995,
                              config=types.SimpleNamespace(tokenizer_max_length=L), seed=3).cuda()
    model.eval()
    M = 2 * pairs
    g = torch.Generator().manual_seed(7)
    ids = torch.randint(1000, 30000, (M, L), generator=g)
    ids[:, 0] = 101
    valid = torch.tensor([L, 37, 5, 64])
    mask = (torch.arange(L).unsqueeze(0) < valid.unsqueeze(1)).long()
    ids = ids * mask
    img = torch.randn(M, 3, 224, 224, generator=g)
    pdid = torch.tensor([501, 502])
    # non-trivial queues so that the queue block of the logits matters
    with torch.no_grad():
        model.query_queue.copy_(torch.nn.functional.normalize(torch.randn(768, K, generator=g), dim=0).cuda())
        model.cand_queue.copy_(torch.nn.functional.normalize(torch.randn(768, K, generator=g), dim=0).cuda())
        model.idx_queue.copy_(torch.arange(900, 900 + K).view(1, K).cuda())
    sd = {n: p.detach().cpu().clone() for n, p in model.named_parameters()}
    for n in sd:
        if sd[n].dtype == torch.float32 and "_m." not in n:
            sd[n].requires_grad_(True)
    state = {"query_queue": model.query_queue.detach().cpu().clone(), "cand_queue": model.cand_queue.detach().cpu().clone(),
             "idx_queue": model.idx_queue.detach().cpu().clone(), "ptr": int(model.new_ptr_queue.item())}
    im = {"query": [[2 * i] for i in range(pairs)], "pos_cand": [[2 * i + 1] for i in range(pairs)]}
    out_o = bo.contrastive_loss(sd, state, {"ids": ids, "mask": mask, "img": img, "index_mapping": im, "p_did_list": pdid},
                                0.4, model.vit_cfg, model.med_cfg, 0.995)
    out_o["loss"].backward()
    batch = {"txt_batched": types.SimpleNamespace(input_ids=ids.cuda(), attention_mask=mask.cuda()), "image_batched": img.cuda(),
             "p_did_list": pdid, "index_mapping": im}
    model.zero_grad()
    out_d = model(batch, alpha=0.4)
    out_d["loss"].backward()
    print("OBS blip-large loss", out_d["loss"].item(), out_o["loss"].item())
    assert abs(out_d["loss"].item() - out_o["loss"].item()) < 2e-2 * max(1.0, abs(out_o["loss"].item()))
    assert out_d["accuracy"].item() == out_o["accuracy"].item()
    assert np.array_equal(model.idx_queue.cpu().numpy(), state["idx_queue"].numpy())
    assert rel(model.query_queue, state["query_queue"]) < 2e-2
    errs = {}
    for n in ("visual_encoder.blocks.0.attn.qkv.weight", "visual_encoder.blocks.23.mlp.fc2.weight", "visual_encoder.pos_embed",
              "visual_encoder.patch_embed.proj.weight", "text_encoder.embeddings.word_embeddings.weight",
              "text_encoder.encoder.layer.0.attention.self.query.weight",
              "text_encoder.encoder.layer.0.crossattention.self.key.weight",
              "text_encoder.encoder.layer.11.crossattention.self.value.weight",
              "text_encoder.encoder.layer.11.output.dense.weight", "text_encoder.pooler.dense.weight"):
        errs[n] = rel(model.get_parameter(n).grad, sd[n].grad)
    print("OBS blip-large grad rel", {k[-40:]: round(v, 4) for k, v in errs.items()})
    assert max(errs.values()) < 8e-2, errs
    assert rel(model.temp.grad, sd["temp"].grad) < 8e-2
    # embedding entry point (forward only)
    with torch.no_grad():
        emb = model.encode_multimodal_input(batch["txt_batched"], batch["image_batched"])
        ref = bo.encode_multimodal_input({k: v.detach() for k, v in sd.items()}, ids, mask, img, model.vit_cfg, model.med_cfg)
    print("OBS blip-large emb rel", rel(emb, ref))
    assert rel(emb, ref) < 2e-2


@pytest.mark.parametrize("score_fusion", [False, True])
def test_packed_bert_rows_equal_the_padded_rows(score_fusion):
    """BERT on the rows up to each caption's valid length only (blip_model.TextPack; reference: padded rows with
    masked keys, med.py:160-232,687-688, only token 0 pooled, blip_ff.py:82-116).  Against pack_text = False on the same weights,
    tokens, images and dropout seeds: the embedding is BITWISE equal in eval mode and in train mode (hidden dropout, attention
    dropout -- masks drawn at the dense coordinates -- and DropPath), every parameter gradient equal up to the order of the fp32
    additions in the weight-gradient reductions (the reduction runs over fewer rows, so the split-K chunks differ).  BERT-base width
    (768, 12 heads, 100 positions, 2 layers) over a small ViT; lengths include a full caption, a one-token caption and ragged ones;
    BLIP_SF: mode "text" (no cross-attention), class-token output."""
    from uniir_amd.blip_model import BLIPFeatureFusion, BLIPScoreFusion
    L, M = 100, 8
    med = dict(hidden_size=768, intermediate_size=3072, num_attention_heads=12, num_hidden_layers=2, vocab_size=30524,
               max_position_embeddings=512, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    vit = dict(img_size=64, patch_size=16, embed_dim=256, depth=2, num_heads=4, drop_path_rate=0.1)
    cls = BLIPScoreFusion if score_fusion else BLIPFeatureFusion
    model = cls(med_config=med, vit_config=vit, embed_dim=768 if not score_fusion else 256, queue_size=16,
                config=types.SimpleNamespace(tokenizer_max_length=L), seed=11).cuda()
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(1000, 30000, (M, L), generator=g)
    ids[:, 0] = 101
    valid = torch.tensor([L, 1, 37, 5, 64, 99, 2, 50])
    mask = (torch.arange(L).unsqueeze(0) < valid.unsqueeze(1)).long()
    ids = ids * mask
    img = torch.randn(M, 3, 64, 64, generator=g).cuda()
    w = torch.randn(M, model.embed_dim, generator=g).cuda()

    def run(pack, train):
        model.pack_text = pack
        model.train(train)
        model.zero_grad()
        txt = types.SimpleNamespace(input_ids=ids.cuda(), attention_mask=mask.cuda())     # fresh tensors: no remembered pack
        torch.manual_seed(21)                                                              # the dropout seeds of this pass
        emb = model.encode_multimodal_input(txt, img)
        rows = model.last_text_rows
        (emb * w).sum().backward()
        grads = {n: p.grad.detach().clone() for n, p in model._online_params() if n not in model._frozen and n != "temp"}
        with torch.no_grad():
            torch.manual_seed(22)
            emb_m = model.encode_multimodal_input(txt, img, use_momentum=True)
        return emb.detach().clone(), emb_m.clone(), grads, rows

    for train in (False, True):
        e_d, m_d, g_d, rows_d = run(False, train)
        e_p, m_p, g_p, rows_p = run(True, train)
        assert rows_d == (M * L, M * L) and rows_p == (int(valid.sum()), M * L)
        assert torch.equal(e_p, e_d), (train, (e_p - e_d).abs().max().item())
        assert torch.equal(m_p, m_d), train                                     # the momentum encoder (no grad) packs the same way
        worst = max((float((g_p[n] - g_d[n]).abs().max()) / (float(g_d[n].abs().max()) + 1e-12), n) for n in g_d)
        assert worst[0] <= 2e-5, (train, worst)
        live = [n for n in g_d if float(g_d[n].abs().max()) > 0]
        assert len(live) >= len(g_d) - (4 if score_fusion else 0)               # (BLIP_SF: nothing else is dead)
    # a batch of full-length captions has nothing to drop: no pack is built
    full = types.SimpleNamespace(input_ids=ids.cuda(), attention_mask=torch.ones(M, L, dtype=torch.long).cuda())
    model.pack_text = True
    model.eval()
    with torch.no_grad():
        model.encode_multimodal_input(full, img)
    assert model.last_text_rows == (M * L, M * L)


def test_blip_ff_training_step_is_reproducible_bit_for_bit():
    """Round 6 (uniir_reduce_scratch): BLIP_FF's gradients go through the same reductions as CLIP's -- LayerNorm weight / bias sums,
    bias gradients out of the weight-gradient GEMMs and the dgrad epilogues, the word-embedding scatter ([CLS] / [PAD] ids shared by
    every caption) -- so two runs of the same two training steps (train mode: hidden / attention dropout and DropPath on, seeded
    through torch's CPU generator) from the same state give the same loss, gradients, momentum weights and queues, bit for bit."""
    z = np.load(os.path.join(G, "g8_blipff.npz"))
    med_cfg, vit_cfg = json.loads(str(z["med_cfg"])), json.loads(str(z["vit_cfg"]))
    med_cfg.update(hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    vit_cfg["drop_path_rate"] = 0.1
    runs = []
    for run in range(2):
        model = tiny_model(med_cfg, vit_cfg, queue_size=int(z["queue_size"]), momentum=float(z["momentum"]))
        load_sub(model, z, "sd0::", "")
        model.copy_params()
        model = model.cuda()
        model.train()
        torch.manual_seed(77)
        rec = []
        for step in range(2):
            b = len(z[f"s{step}_pdid"])
            batch = {
                "txt_batched": types.SimpleNamespace(input_ids=torch.from_numpy(z[f"s{step}_ids"]).cuda(),
                                                     attention_mask=torch.from_numpy(z[f"s{step}_mask"]).cuda()),
                "image_batched": torch.from_numpy(z[f"s{step}_img"]).cuda(),
                "p_did_list": torch.from_numpy(z[f"s{step}_pdid"]),
                "index_mapping": {"query": [[2 * i] for i in range(b)], "pos_cand": [[2 * i + 1] for i in range(b)]},
            }
            model.zero_grad()
            out = model(batch, alpha=float(z[f"s{step}_alpha"]))
            out["loss"].backward()
            torch.cuda.synchronize()
            rec.append((out["loss"].detach().clone(), model._online.g32.clone(), model._mom.p32.clone(), model.query_queue.clone(),
                        model.cand_queue.clone()))
            with torch.no_grad():
                for n, p in model._online_params():
                    if n not in model._frozen:
                        p.add_(-0.05 * p.grad)
        runs.append(rec)
    for step in range(2):
        for k, (a, b) in enumerate(zip(runs[0][step], runs[1][step])):
            assert torch.equal(a, b), (step, k, int((a != b).sum()), float((a.float() - b.float()).abs().max()))
