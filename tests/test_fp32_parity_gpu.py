"""The north-star parity bar, end to end: "results match the reference PyTorch path on the same inputs (fp32 logits within
1e-3, identical top-k candidate ids)" (BASELINE.json).  The bf16 MFMA towers cannot meet 1e-3 on logits of magnitude ~14
(bf16 has 8 mantissa bits), so the bar is demonstrated on the fp32 forward path of the same module surface
(`clip_model.precision = "fp32"`: exact-fp32 MFMA GEMMs, fp32 attention and LayerNorm, csrc/fp32_path.hip) -- what the
reference computes after model.float() -- at BASELINE configs[0] AS WRITTEN: CLIP_SF ViT-B/32, batch 32, against the
oracle (pinned to the reference by goldens G1-G5).  Reference entry point: clip_sf.py:53-63,88-97,134-144."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "uniir_amd", "src"))


def test_fp32_kernels_against_torch():
    from uniir_amd import ops
    torch.manual_seed(3)
    # attention: self (causal and not), cross with key lengths, sequence > 256 rows per pass
    for (b, tq, tk, h, causal, klen) in [(2, 50, 50, 2, 0, None), (2, 77, 77, 3, 1, None), (1, 300, 300, 1, 0, None),
                                         (3, 20, 45, 2, 0, [45, 7, 1])]:
        W = h * 64
        q = torch.randn(b * tq, W, device=DEV)
        kv = torch.randn(b * tk, 2 * W, device=DEV)
        out = torch.empty(b * tq, W, device=DEV)
        kl = None if klen is None else torch.tensor(klen, device=DEV, dtype=torch.int32)
        ops.call("uniir_attention_f32_fwd", q, W, kv, kv[:, W:], 2 * W, out, W, kl, b, tq, tk, h, causal, 0.125)
        qq = q.double().view(b, tq, h, 64).transpose(1, 2)
        kk = kv[:, :W].double().reshape(b, tk, h, 64).transpose(1, 2)
        vv = kv[:, W:].double().reshape(b, tk, h, 64).transpose(1, 2)
        s = qq @ kk.transpose(-1, -2) * 0.125
        if causal:
            s = s + torch.full((tq, tk), float("-inf"), device=DEV, dtype=torch.float64).triu_(1)
        if klen is not None:
            dead = torch.arange(tk, device=DEV)[None, :] >= kl[:, None].long()
            s = s.masked_fill(dead[:, None, None, :], float("-inf"))
        ref = (torch.softmax(s, -1) @ vv).transpose(1, 2).reshape(b * tq, W)
        assert (out.double() - ref).abs().max().item() < 2e-6
    # bias + activation + residual
    y = torch.randn(37, 96, device=DEV)
    bias, resid = torch.randn(96, device=DEV), torch.randn(37, 96, device=DEV)
    for act, fn in ((-1, lambda t: t), (ops.ACT_QUICKGELU, lambda t: t * torch.sigmoid(1.702 * t)),
                    (ops.ACT_GELU_ERF, torch.nn.functional.gelu), (ops.ACT_RELU, torch.relu)):
        z = y.clone()
        ops.call("uniir_bias_act_f32", z, bias, resid, 37, 96, act)
        assert (z - (resid + fn(y + bias))).abs().max().item() < 2e-6
    z = y.clone()
    ops.call("uniir_bias_act_f32", z, None, None, 37, 96, ops.ACT_QUICKGELU)
    assert (z - y * torch.sigmoid(1.702 * y)).abs().max().item() < 2e-6


def _build(name, seed):
    from oracle import clip_oracle as O
    from models.uniir_clip.clip_scorefusion.clip_sf import CLIPScoreFusion
    from uniir_amd.clip_model import CLIP_CONFIGS
    cfg = CLIP_CONFIGS[name]
    sd = O.init_state_dict(cfg, seed=seed)
    config = SimpleNamespace(model=SimpleNamespace(gather_embeddings=False), data_config=SimpleNamespace(in_batch_neg_num=0))
    model = CLIPScoreFusion(name, device=DEV, config=config)
    model.float()
    model.clip_model.load_state_dict(sd, strict=True)
    return O, cfg, sd, model


def test_config1_as_written_fp32_logits_within_1e_3_and_identical_top10():
    from oracle import c_oracle
    from uniir_amd import retrieval
    from uniir_amd.losses import InBatchNCEFn
    O, cfg, sd, model = _build("ViT-B/32", seed=1)
    pairs = 32                                                              # BASELINE configs[0]: batch 32
    batch = O.synthetic_batch(cfg, pairs, seed=2023)
    dbatch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    oracle = O.OracleCLIP(cfg, sd)
    with torch.no_grad():
        emb_o = O.encode_multimodal_input(oracle.sd(), cfg, batch["txt_batched"], batch["image_batched"],
                                          batch["txt_mask_batched"], batch["image_mask_batched"])
        out_o = O.inbatch_contrastive_loss(emb_o, batch["index_mapping"], oracle.logit_scale.exp())
    model.eval()
    model.clip_model.precision = "fp32"
    with torch.no_grad():
        emb_d = model.encode_multimodal_input(dbatch["txt_batched"], dbatch["image_batched"], dbatch["txt_mask_batched"],
                                              dbatch["image_mask_batched"])
        idx_q = torch.arange(0, 2 * pairs, 2, device=DEV, dtype=torch.int32)
        idx_p = idx_q + 1
        loss_d, acc_d, score_d = InBatchNCEFn.apply(emb_d, idx_q, idx_p, model.get_logit_scale(), False)
        out_d = model(dbatch)                                               # the module's own entry point, same numbers
    scale = float(emb_o.abs().max())
    emb_err = (emb_d.cpu() - emb_o).abs().max().item()
    assert emb_err < 2e-5 * max(1.0, scale), (emb_err, scale)
    logit_err = (score_d.cpu() - out_o["score"]).abs().max().item()
    assert out_o["score"].abs().max().item() > 1.0                          # logits are O(10): 1e-3 is a real constraint
    assert logit_err < 1e-3, logit_err                                      # the north-star tolerance
    assert abs(loss_d.item() - out_o["loss"].item()) < 1e-4 and acc_d.item() == out_o["accuracy"].item()
    assert abs(out_d["loss"].item() - loss_d.item()) < 1e-6
    # identical top-10 candidate ids: the 64 item embeddings as the pool (fp16, as the embedder stores them), the 32
    # queries searched on the device path vs the C oracle on the oracle's embeddings
    ids = (np.arange(2 * pairs) * 7 + 9_000_001).astype(np.int64)
    pool_o = emb_o.numpy().astype(np.float16)
    want_s, want_i = c_oracle.topk(pool_o, ids, pool_o[0::2], 10)
    pool_d = emb_d.half()
    got_s, got_i = retrieval.search_shard(retrieval.PoolShard(pool_d, torch.from_numpy(ids)), pool_d[0::2].contiguous(), 10)
    # ids must be identical wherever the oracle's own ranking is not a near tie (neighbouring scores > 1e-5 apart: the two
    # embedding sets differ by ~1e-6 before the fp16 rounding of the stored pool)
    gap = np.abs(np.diff(want_s, axis=1))
    clear = np.ones_like(want_i, dtype=bool)
    clear[:, 1:] &= gap > 1e-5
    clear[:, :-1] &= gap > 1e-5
    got_i_np = got_i.cpu().numpy()
    assert clear.mean() > 0.9 and np.array_equal(got_i_np[clear], want_i[clear])
    assert all(set(a) == set(b) for a, b, c in zip(got_i_np, want_i, clear) if c.all())
    assert np.abs(got_s.cpu().numpy() - want_s).max() < 1e-3
    # The 16-bit MFMA towers on the same batch.  fp16 (round 5: the embedder's precision, mbeir_embedder.py:52-56): embeddings within
    # 1.5e-3 of the fp32 oracle; bf16 (the training precision): 1e-2.  "Identical top-10 ids" for either is asserted where the
    # oracle's own ranking is clear at THAT precision's score error (neighbouring scores further apart than twice the largest
    # difference between the two 32 x 64 cosine matrices: a swap of two candidates needs a gap below that), which is most positions
    # for fp16 and a minority for bf16 -- bf16 embeddings do reorder near ties.
    def cosines(pool16):
        x = pool16.astype(np.float32)
        x = x / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-30)
        return x[0::2] @ x.T

    cos_o = cosines(pool_o)
    top11 = -np.sort(-cos_o, axis=1)[:, :11]
    def sixteen_bit(precision, emb_tol, logit_tol, min_clear):
        model.clip_model.precision = precision
        with torch.no_grad():
            emb_x = model.encode_multimodal_input(dbatch["txt_batched"], dbatch["image_batched"], dbatch["txt_mask_batched"],
                                                  dbatch["image_mask_batched"])
            _l, _a, score_x = InBatchNCEFn.apply(emb_x, idx_q, idx_p, model.get_logit_scale(), False)
        assert (score_x.cpu() - out_o["score"]).abs().max().item() < logit_tol
        rel = ((emb_x.cpu() - emb_o).norm() / emb_o.norm()).item()
        assert rel < emb_tol, (precision, rel)
        pool_x = emb_x.half()
        xs, xi = retrieval.search_shard(retrieval.PoolShard(pool_x, torch.from_numpy(ids)), pool_x[0::2].contiguous(), 10)
        serr = float(np.abs(cosines(pool_x.cpu().numpy()) - cos_o).max())
        ok = np.ones_like(want_i, dtype=bool)
        ok[:, 1:] &= gap > 2 * serr + 1e-5
        ok[:, :-1] &= gap > 2 * serr + 1e-5
        ok[:, -1] &= (top11[:, 9] - top11[:, 10]) > 2 * serr + 1e-5          # the 10th entry against the first one NOT returned
        assert ok.mean() > min_clear, (precision, ok.mean(), serr)
        assert np.array_equal(xi.cpu().numpy()[ok], want_i[ok]), precision
        return rel, serr

    rel_h, serr_h = sixteen_bit("fp16", 1.5e-3, 0.05, 0.5)
    rel_b, serr_b = sixteen_bit("bf16", 1e-2, 0.25, 0.1)
    assert rel_h < 0.35 * rel_b                       # three more mantissa bits


def test_fp16_forward_refuses_backward_and_matches_the_oracle_at_vit_l14():
    """precision = "fp16" (round 5): the reference embedder's autocast(fp16) forward (mbeir_embedder.py:52-56) on the MFMA towers --
    fp16 weights, activations and attention operands, fp32 residual stream / LayerNorm / accumulation -- forward only.  ViT-L/14,
    4 items incl. the packed text tower: embeddings within 1.5e-3 (relative, per item) of the fp32 oracle; training mode raises."""
    O, cfg, sd, model = _build("ViT-L/14", seed=3)
    batch = O.synthetic_batch(cfg, 2, seed=11)
    dbatch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    with torch.no_grad():
        emb_o = O.encode_multimodal_input(sd, cfg, batch["txt_batched"], batch["image_batched"], batch["txt_mask_batched"],
                                          batch["image_mask_batched"])
    model.eval()
    model.clip_model.precision = "fp16"
    with torch.no_grad():
        emb_h = model.encode_multimodal_input(dbatch["txt_batched"], dbatch["image_batched"], dbatch["txt_mask_batched"],
                                              dbatch["image_mask_batched"]).cpu()
        t_h = model.encode_text(dbatch["txt_batched"]).cpu()
        model.clip_model.pack_text = False
        t_dense = model.encode_text(dbatch["txt_batched"]).cpu()
        model.clip_model.pack_text = True
    for r in range(emb_o.shape[0]):
        rel = ((emb_h[r] - emb_o[r]).norm() / emb_o[r].norm()).item()
        assert rel < 1.5e-3, (r, rel)
    assert torch.equal(t_h, t_dense)                  # packed == dense text tower, bitwise, in fp16 too
    model.train()
    with pytest.raises(RuntimeError):
        model(dbatch)
    model.clip_model.precision = "bf16"               # and training is unaffected afterwards
    out = model(dbatch)
    out["loss"].backward()


def test_fp32_precision_refuses_backward():
    O, cfg, sd, model = _build("ViT-B/32", seed=2)
    model.clip_model.precision = "fp32"
    batch = O.synthetic_batch(cfg, 2, seed=5)
    dbatch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    model.train()
    with pytest.raises(RuntimeError):
        model(dbatch)


def test_vit_l14_two_pairs_bf16_against_the_oracle():
    """the headline architecture itself (ViT-L/14: patch K 588 -> 640 padding, 257-token attention inside the tower, 16
    heads, width-1024 LayerNorm) against the fp32 oracle on 2 pairs: embeddings, loss, accuracy, every parameter
    gradient, gradient cosine -- gates at ~2x the observed bf16 error"""
    O, cfg, sd, model = _build("ViT-L/14", seed=3)
    oracle = O.OracleCLIP(cfg, sd)
    pairs = 2
    batch = O.synthetic_batch(cfg, pairs, seed=77)
    dbatch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    emb_o = O.encode_multimodal_input(oracle.sd(), cfg, batch["txt_batched"], batch["image_batched"],
                                      batch["txt_mask_batched"], batch["image_mask_batched"])
    out_o = O.inbatch_contrastive_loss(emb_o, batch["index_mapping"], oracle.logit_scale.exp())
    out_o["loss"].backward()

    def rel(a, b):
        a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
        return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()

    model.train()
    model.zero_grad()
    emb_d = model.encode_multimodal_input(dbatch["txt_batched"], dbatch["image_batched"], dbatch["txt_mask_batched"],
                                          dbatch["image_mask_batched"])
    assert rel(emb_d, emb_o) < 1.2e-2, rel(emb_d, emb_o)          # observed 6.2e-3
    out_d = model(dbatch)
    assert abs(out_d["loss"].item() - out_o["loss"].item()) < 1e-2 * max(1.0, abs(out_o["loss"].item()))
    assert out_d["accuracy"].item() == out_o["accuracy"].item()
    out_d["loss"].backward()
    errs, gd, go = {}, [], []
    for n, p in model.clip_model.named_parameters():
        g = getattr(oracle, n.replace(".", "__")).grad
        if g is None or g.abs().max() == 0:
            continue
        errs[n] = rel(p.grad, g)
        gd.append(p.grad.flatten().cpu())
        go.append(g.flatten())
    worst = max(errs.values())
    print(f"ViT-L/14 2 pairs: emb rel {rel(emb_d, emb_o):.2e}, worst grad rel {worst:.2e}")
    big = {n: e for n, e in errs.items() if e > 1e-1}           # observed worst 5.9e-2 (bias / LayerNorm gradients of 4 items)
    assert not big, big
    assert torch.nn.functional.cosine_similarity(torch.cat(gd), torch.cat(go), dim=0).item() > 0.9995
    # Is a 6 % bias-gradient error rounding noise or a bug?  The oracle that rounds to bf16 where the device does
    # (oracle/clip_oracle.py rounding("bf16")) answers per parameter: its own distance from the fp32 oracle is the noise this batch
    # gives that parameter under 16-bit storage (gradients summed over a handful of items cancel, so a 4e-3 rounding of the terms is
    # several per cent of the sum); the device, which rounds at the same points but sums in another order, must stay within a small
    # multiple of that noise -- a kernel bug would not
    oracle16 = O.OracleCLIP(cfg, sd)
    with O.rounding("bf16"):
        emb_r = O.encode_multimodal_input(oracle16.sd(), cfg, batch["txt_batched"], batch["image_batched"],
                                          batch["txt_mask_batched"], batch["image_mask_batched"])
        out_r = O.inbatch_contrastive_loss(emb_r, batch["index_mapping"], oracle16.logit_scale.exp())
        out_r["loss"].backward()
    assert rel(emb_d, emb_r) < 8e-3, rel(emb_d, emb_r)              # observed 3.7e-3 (vs the fp32 oracle: 5.9e-3)
    ratio = {}
    for n, p in model.clip_model.named_parameters():
        g32, g16 = getattr(oracle, n.replace(".", "__")).grad, getattr(oracle16, n.replace(".", "__")).grad
        if g32 is None or g32.abs().max() == 0:
            continue
        noise = rel(g16, g32)                                       # what bf16 storage does to this parameter's gradient
        ratio[n] = (errs[n], noise, errs[n] / (noise + 2e-3))
    worst16 = max(ratio, key=lambda n: ratio[n][2])
    print("OBS L14 device error / bf16-oracle noise: worst", worst16, ratio[worst16], "median",
          sorted(r[2] for r in ratio.values())[len(ratio) // 2])
    # observed: median 0.87-0.90 at ViT-B/32 and ViT-L/14 (the device IS the bf16-rounded computation), worst 1.07 (L/14) / 2.8
    # (B/32, the scalar logit_scale: one number, no averaging)
    bad = {n: r for n, r in ratio.items() if r[2] > 4.0}
    assert not bad, bad
    assert sorted(r[2] for r in ratio.values())[len(ratio) // 2] < 1.5
    # and the fp32 forward of the same architecture: 257-token fp32 attention, K = 588 patch GEMM
    model.eval()
    model.clip_model.precision = "fp32"
    with torch.no_grad():
        emb_f = model.encode_multimodal_input(dbatch["txt_batched"], dbatch["image_batched"], dbatch["txt_mask_batched"],
                                              dbatch["image_mask_batched"])
    assert (emb_f.cpu() - emb_o.detach()).abs().max().item() < 2e-5 * max(1.0, float(emb_o.abs().max()))
