"""CPU tests: the oracle (oracle/clip_oracle.py, oracle/oracle.c) against the golden fixtures captured from the
reference itself (tests/golden/make_golden.py).  These pin the checker before any HIP result is trusted."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle
from oracle import clip_oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def _emb(d, pre):
    txt, img = torch.tensor(d[f"{pre}txt"]), torch.tensor(d[f"{pre}img"])
    tm = torch.tensor(d[f"{pre}tmask"]) if f"{pre}tmask" in d else torch.ones(txt.shape[0], dtype=torch.long)
    im = torch.tensor(d[f"{pre}imask"]) if f"{pre}imask" in d else torch.ones(txt.shape[0], dtype=torch.long)
    txt.requires_grad_(True)
    img.requires_grad_(True)
    return txt, img, img * im.unsqueeze(-1) + txt * tm.unsqueeze(-1)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_g1_infonce_w1(tag):
    d = load("g1_infonce_w1.npz")
    txt, img, emb = _emb(d, f"{tag}_")
    b = emb.shape[0] // 2
    im = {"query": [[2 * i] for i in range(b)], "pos_cand": [[2 * i + 1] for i in range(b)]}
    ls = torch.tensor(float(np.log(1 / 0.07)), requires_grad=True)
    out = O.inbatch_contrastive_loss(emb, im, ls.exp())
    out["loss"].backward()
    assert abs(out["loss"].item() - float(d[f"{tag}_loss"])) < 1e-6
    assert out["accuracy"].item() == float(d[f"{tag}_acc"])
    assert np.abs(out["score"].detach().numpy() - d[f"{tag}_score"]).max() < 1e-5
    assert np.abs(txt.grad.numpy() - d[f"{tag}_dtxt"]).max() < 1e-6
    assert np.abs(img.grad.numpy() - d[f"{tag}_dimg"]).max() < 1e-6
    assert abs(ls.grad.item() - float(d[f"{tag}_dscale"])) < 1e-5
    # C restatement (exact fma-chain logits) against the reference's recorded logits: north-star 1e-3
    with torch.no_grad():
        q = torch.nn.functional.normalize(emb[0::2], dim=-1).numpy()
        p = torch.nn.functional.normalize(emb[1::2], dim=-1).numpy()
    sc = c_oracle.infonce_scores(q, p, float(np.exp(np.log(1 / 0.07))))
    assert np.abs(sc - d[f"{tag}_score"]).max() < 1e-4
    loss, acc, _ = c_oracle.infonce_loss(sc)
    assert abs(loss - float(d[f"{tag}_loss"])) < 1e-5 and acc == float(d[f"{tag}_acc"])


def test_g2_infonce_two_ranks():
    """the gather branch (clip_sf.py:102-103,134-136) restated without a process group: all_p = rank-major concat;
    d p_r = sum over ranks of their d all_p slice r (the autograd all-gather's backward)."""
    d = load("g2_infonce_w2.npz")
    W = 2
    embs, leaves = [], []
    for r in range(W):
        txt, img, emb = _emb(d, f"r{r}_")
        embs.append(emb)
        leaves.append((txt, img))
    b = embs[0].shape[0] // 2
    im = {"query": [[2 * i] for i in range(b)], "pos_cand": [[2 * i + 1] for i in range(b)]}
    ls = torch.tensor(float(np.log(1 / 0.07)))
    p_all = [torch.nn.functional.normalize(e[1::2], dim=-1) for e in embs]
    total = 0
    outs = []
    for r in range(W):
        gather = lambda p, r=r: torch.cat([p if j == r else p_all[j] for j in range(W)], dim=0)
        out = O.inbatch_contrastive_loss(embs[r], im, ls.exp(), gather=gather, rank=r)
        outs.append(out)
        total = total + out["loss"]          # every rank back-propagates its own loss; all-gather bwd sums them
    total.backward()
    for r in range(W):
        assert abs(outs[r]["loss"].item() - float(d[f"r{r}_loss"])) < 1e-6
        assert outs[r]["accuracy"].item() == float(d[f"r{r}_acc"])
        assert np.abs(outs[r]["score"].detach().numpy() - d[f"r{r}_score"]).max() < 1e-5
        assert np.array_equal(d[f"r{r}_targets"], r * b + np.arange(b))
        assert np.abs(leaves[r][0].grad.numpy() - d[f"r{r}_dtxt"]).max() < 1e-6


@pytest.mark.parametrize("tag,ibn", [("n0", 0), ("n2", 2)])
def test_g3_hard_negative_branch(tag, ibn):
    d = load("g3_hardneg.npz")
    txt, img, emb = _emb(d, f"{tag}_")
    b, nneg = 4, 2
    im = {"query": [], "pos_cand": [], "neg_cand_list": []}
    c = 0
    for i in range(b):
        im["query"].append([c]); c += 1
        im["pos_cand"].append([c]); c += 1
        im["neg_cand_list"].append(list(range(c, c + nneg))); c += nneg
    out = O.inbatch_contrastive_loss(emb, im, torch.tensor(1 / 0.07), in_batch_neg_num=ibn)
    out["loss"].backward()
    assert abs(out["loss"].item() - float(d[f"{tag}_loss"])) < 1e-5
    assert out["accuracy"].item() == float(d[f"{tag}_acc"])
    assert np.abs(txt.grad.numpy() - d[f"{tag}_dtxt"]).max() < 1e-5


def test_g4_mask_semantics():
    d = load("g4_masks.npz")
    cfg = O.tiny_config()
    m = O.OracleCLIP(cfg, seed=7)
    batch = O.synthetic_batch(cfg, 3, seed=21)
    with torch.no_grad():
        emb = O.encode_multimodal_input(m.sd(), cfg, batch["txt_batched"], batch["image_batched"],
                                        torch.tensor(d["tmask"]), torch.tensor(d["imask"]))
        out = O.inbatch_contrastive_loss(emb, batch["index_mapping"], m.logit_scale.exp())
    assert np.abs(emb.numpy() - d["emb"]).max() < 1e-5
    assert abs(out["loss"].item() - float(d["loss"])) < 1e-5


def test_g5_towers_match_independent_implementation():
    """encoder arithmetic (third-party in the reference, unpinned there) against transformers.CLIPModel outputs."""
    d = load("g5_hf_clip.npz")
    cfg = json.loads(str(d["cfg"]))
    sd = {k[4:]: torch.tensor(d[k]) for k in d.files if k.startswith("sd::")}
    with torch.no_grad():
        t = O.encode_text(sd, torch.tensor(d["txt"]), cfg)
        i = O.encode_image(sd, torch.tensor(d["img"]), cfg)
    assert np.abs(t.numpy() - d["text_features"]).max() < 2e-5
    assert np.abs(i.numpy() - d["image_features"]).max() < 2e-5


def test_g10_training_trajectory():
    """reference engine.train_one_epoch (accumulation 2, AdamW groups, cosine LR) vs the oracle's restatement."""
    d = load("g10_train.npz")
    cfg = json.loads(str(d["cfg"]))
    sd0 = {k[5:]: torch.tensor(d[k]) for k in d.files if k.startswith("sd0::")}
    model = O.OracleCLIP(cfg, sd0)
    nd, dec = O.weight_decay_groups(model.named_parameters())
    assert len(nd) == int(d["n_nodecay"]) and len(dec) == int(d["n_decay"])
    lr, accum, T = float(d["lr"]), int(d["accum"]), int(d["t_total"])
    opt = torch.optim.AdamW([{"params": [p for _, p in nd], "weight_decay": 0.0},
                             {"params": [p for _, p in dec], "weight_decay": 0.2}], lr=lr, betas=(0.9, 0.98), eps=1e-6)
    losses, lrs, steps = [], [], 0
    opt.zero_grad()
    for i in range(4):
        batch = O.synthetic_batch(cfg, 4, seed=40 + i)
        emb = O.encode_multimodal_input(model.sd(), cfg, batch["txt_batched"], batch["image_batched"],
                                        batch["txt_mask_batched"], batch["image_mask_batched"])
        out = O.inbatch_contrastive_loss(emb, batch["index_mapping"], model.logit_scale.exp())
        (out["loss"] / accum).backward()
        if (i + 1) % accum == 0:
            for gp in opt.param_groups:
                gp["lr"] = O.cosine_lr(lr, steps, T)
            opt.step()
            opt.zero_grad()
            steps += 1
        losses.append(out["loss"].item())
        lrs.append(O.cosine_lr(lr, steps, T))       # engine.py:49 logs lr AFTER scheduler.step()
    assert np.abs(np.array(losses) - d["losses"]).max() < 2e-4
    assert np.abs(np.array(lrs) - d["lrs"]).max() < 1e-9
    for k in ["visual.proj", "transformer.resblocks.1.mlp.c_fc.weight", "logit_scale", "visual.ln_pre.weight"]:
        got = getattr(model, k.replace(".", "__")).detach().numpy()
        assert np.abs(got - d[f"sd1::{k}"]).max() < 5e-5, k


def test_g11_embedder_semantics():
    """mbeir_embedder.py:44-57: per-batch forward, .half() per batch, concat, ids extended in order."""
    d = load("g11_embedder.npz")
    cfg = O.tiny_config()
    m = O.OracleCLIP(cfg, seed=12)
    embs, ids = [], []
    with torch.no_grad():
        for i in range(2):
            b = O.synthetic_batch(cfg, 2 + i, seed=60 + i)
            e = O.encode_multimodal_input(m.sd(), cfg, b["txt_batched"], b["image_batched"], b["txt_mask_batched"],
                                          b["image_mask_batched"])
            embs.append(e.half())
            ids += [1000 * (i + 1) + j for j in range(e.shape[0])]
    got = torch.cat(embs).numpy()
    assert got.dtype == np.float16 and np.array_equal(np.array(ids), d["ids"])
    assert np.abs(got.astype(np.float32) - d["emb"].astype(np.float32)).max() < 2e-3


def test_c_topk_oracle_known_answers():
    rng = np.random.default_rng(0)
    n, d, k = 500, 64, 10
    pool = rng.standard_normal((n, d)).astype(np.float16)
    ids = (rng.permutation(n) * 3 + 7).astype(np.int64)
    q = rng.standard_normal((6, d)).astype(np.float16)
    q[0] = (pool[42].astype(np.float32) * 3).astype(np.float16)      # planted neighbour
    pool[7] = 0                                                      # zero row stays zero (FAISS semantics)
    pool[100] = pool[50]; pool[101] = pool[50]                       # duplicates -> id tie-break
    q[1] = pool[50]
    s, i = c_oracle.topk(pool, ids, q, k)
    assert i[0, 0] == ids[42] and abs(s[0, 0] - 1.0) < 1e-3
    assert sorted(i[1, :3].tolist()) == sorted(ids[[50, 100, 101]].tolist()) and list(i[1, :3]) == sorted(i[1, :3])
    pn = pool.astype(np.float64)
    nrm = np.linalg.norm(pn, axis=1, keepdims=True); nrm[nrm == 0] = 1
    qn = q.astype(np.float64) / np.linalg.norm(q.astype(np.float64), axis=1, keepdims=True)
    ref = qn @ (pn / nrm).T
    top = np.sort(ref, axis=1)[:, ::-1][:, :k]
    assert np.abs(s - top).max() < 2e-6
    assert (np.diff(s, axis=1) <= 0).all()
    s2, i2 = c_oracle.topk(pool[:4], ids[:4], q, k)                  # fewer rows than k: (-inf, -1) padding
    assert (i2[:, 4:] == -1).all() and np.isinf(s2[:, 4:]).all()


def test_bf16_rounding_mode_is_a_small_perturbation_of_the_fp32_restatement():
    """oracle.clip_oracle.rounding("bf16"): the same restatement with the device's 16-bit rounding points -- forward and every
    parameter gradient stay within bf16 noise of the fp32 run (and differ from it), and nothing changes outside the context"""
    from oracle import clip_oracle as O
    cfg = O.tiny_config(vision_width=128, vision_layers=2, transformer_width=64, transformer_heads=1, transformer_layers=2)
    sd = O.init_state_dict(cfg, seed=4)
    batch = O.synthetic_batch(cfg, 6, seed=8)

    def run(mode):
        m = O.OracleCLIP(cfg, sd)
        with O.rounding(mode):
            emb = O.encode_multimodal_input(m.sd(), cfg, batch["txt_batched"], batch["image_batched"], batch["txt_mask_batched"],
                                            batch["image_mask_batched"])
            out = O.inbatch_contrastive_loss(emb, batch["index_mapping"], m.logit_scale.exp())
            out["loss"].backward()
        return emb.detach(), out["loss"].item(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}

    e32, l32, g32 = run(None)
    e16, l16, g16 = run("bf16")
    e32b, l32b, _ = run(None)
    assert torch.equal(e32, e32b) and l32 == l32b                       # the context leaves nothing behind
    rel = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-12)).item()
    assert 1e-4 < rel(e16, e32) < 2e-2
    assert abs(l16 - l32) < 2e-2 * max(1.0, abs(l32))
    worst = max(rel(g16[n], g32[n]) for n in g32 if g32[n].abs().max() > 0)
    assert 1e-4 < worst < 1.5e-1, worst
    flat16 = torch.cat([g16[n].flatten() for n in g32]); flat32 = torch.cat([g32[n].flatten() for n in g32])
    assert torch.nn.functional.cosine_similarity(flat16, flat32, dim=0).item() > 0.999
