"""GPU parity against the C oracle where the arithmetic order is pinned: InfoNCE logits (bit-exact: both are
k-ordered fmaf chains) and brute-force top-k (bit-exact distances, identical ids), plus the golden G1/G2 vectors
captured from the reference itself."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _nce(q, ap, scale, toff):
    from uniir_amd import ops
    b, E = q.shape
    B = ap.shape[0]
    score = torch.empty(b, B, device=DEV)
    stats = torch.empty(3 * b, device=DEV)
    loss, acc = torch.empty(1, device=DEV), torch.empty(1, device=DEV)
    ops.call("uniir_infonce_fwd", q, ap, torch.tensor([scale], device=DEV), b, B, E, toff, score, stats, loss, acc)
    return score, loss, acc


@pytest.mark.parametrize("b,B,E,toff", [(5, 5, 24, 0), (32, 96, 512, 32), (130, 260, 768, 130)])
def test_infonce_logits_bit_exact_vs_c_oracle(b, B, E, toff):
    from oracle import c_oracle
    rng = np.random.default_rng(b)
    q = rng.standard_normal((b, E)).astype(np.float32)
    p = rng.standard_normal((B, E)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    p /= np.linalg.norm(p, axis=1, keepdims=True)
    scale = float(np.float32(1 / 0.07))
    score, loss, acc = _nce(torch.tensor(q, device=DEV), torch.tensor(p, device=DEV), scale, toff)
    want = c_oracle.infonce_scores(q, p, scale)
    assert np.array_equal(score.cpu().numpy(), want)          # bit-exact logits
    l, a, _ = c_oracle.infonce_loss(want, toff)
    assert abs(loss.item() - l) < 2e-6 * max(1, abs(l)) and acc.item() == a


def test_infonce_golden_reference_vectors():
    """G1 (reference CLIPScoreFusion, W=1) and G2 (reference under 2 gloo ranks): fp32 logits within 1e-3."""
    from uniir_amd import ops
    d = np.load(os.path.join(G, "g1_infonce_w1.npz"))
    for tag in ("a", "b"):
        emb = torch.tensor((d[f"{tag}_img"] * d[f"{tag}_imask"][:, None].astype(np.float32)
                            + d[f"{tag}_txt"] * d[f"{tag}_tmask"][:, None].astype(np.float32)).astype(np.float32), device=DEV)
        b, E = emb.shape[0] // 2, emb.shape[1]
        q, p = torch.empty(b, E, device=DEV), torch.empty(b, E, device=DEV)
        iq = torch.arange(0, 2 * b, 2, device=DEV, dtype=torch.int32)
        ip = iq + 1
        ops.call("uniir_select_normalize", emb, iq, q, None, b, E)
        ops.call("uniir_select_normalize", emb, ip, p, None, b, E)
        score, loss, acc = _nce(q, p, float(np.exp(np.log(1 / 0.07))), 0)
        assert np.abs(score.cpu().numpy() - d[f"{tag}_score"]).max() < 1e-3
        assert abs(loss.item() - float(d[f"{tag}_loss"])) < 1e-4
        assert acc.item() == float(d[f"{tag}_acc"])
    d = np.load(os.path.join(G, "g2_infonce_w2.npz"))
    ps = []
    for r in range(2):
        emb = torch.tensor(d[f"r{r}_img"] + d[f"r{r}_txt"], device=DEV)
        ps.append(torch.nn.functional.normalize(emb[1::2], dim=-1))
    allp = torch.cat(ps)
    for r in range(2):
        emb = torch.tensor(d[f"r{r}_img"] + d[f"r{r}_txt"], device=DEV)
        q = torch.nn.functional.normalize(emb[0::2], dim=-1).contiguous()
        b = q.shape[0]
        score, loss, acc = _nce(q, allp.contiguous(), float(np.exp(np.log(1 / 0.07))), r * b)
        assert np.abs(score.cpu().numpy() - d[f"r{r}_score"]).max() < 1e-3
        assert abs(loss.item() - float(d[f"r{r}_loss"])) < 1e-4 and acc.item() == float(d[f"r{r}_acc"])


@pytest.mark.parametrize("n,nq,d,k", [(3000, 9, 64, 10), (20000, 40, 768, 10), (9000, 17, 512, 50), (7, 4, 64, 10)])
def test_topk_bit_exact_vs_c_oracle(n, nq, d, k):
    from oracle import c_oracle
    from uniir_amd import retrieval
    rng = np.random.default_rng(n + nq)
    pool = rng.standard_normal((n, d)).astype(np.float16)
    qs = rng.standard_normal((nq, d)).astype(np.float16)
    ids = (rng.permutation(n).astype(np.int64) * 5 + 11)
    if n > 200:
        pool[5] = 0
        pool[150:153] = pool[77]
        qs[0] = pool[77]
        qs[1] = (pool[33].astype(np.float32) * 2).astype(np.float16)
    want_s, want_i = c_oracle.topk(pool, ids, qs, k)
    shard = retrieval.PoolShard(torch.tensor(pool, device=DEV), torch.tensor(ids, device=DEV))
    s, i = retrieval.search_shard(shard, torch.tensor(qs, device=DEV), k)
    assert np.array_equal(i.cpu().numpy(), want_i)
    assert np.array_equal(s.cpu().numpy(), want_s)            # bit-exact distances


@pytest.mark.parametrize("tag,ibn", [("n0", 0), ("n2", 2)])
def test_hard_negative_branch_matches_reference_golden(tag, ibn):
    """G3 was produced by the reference's own compute_inbatch_contrastive_loss (hard-negative branch) on CPU"""
    from uniir_amd.losses import HardNegNCEFn
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "g3_hardneg.npz"))
    emb = torch.tensor(d[f"{tag}_txt"] + d[f"{tag}_img"], device="cuda", requires_grad=True)
    b, nneg = 4, 2
    iq, ip, ineg, c = [], [], [], 0
    for i in range(b):
        iq.append(c); c += 1
        ip.append(c); c += 1
        ineg += list(range(c, c + nneg)); c += nneg
    t = lambda x: torch.tensor(x, dtype=torch.int32, device="cuda")
    scale = torch.tensor(1 / 0.07, device="cuda", requires_grad=True)
    loss, acc = HardNegNCEFn.apply(emb, t(iq), t(ip), t(ineg), scale, ibn)
    loss.backward()
    assert abs(loss.item() - float(d[f"{tag}_loss"])) < 1e-5
    assert acc.item() == float(d[f"{tag}_acc"])
    assert np.abs(emb.grad.cpu().numpy() - d[f"{tag}_dtxt"]).max() < 1e-5
    # the golden holds d loss / d logit_scale (the log-domain parameter): d/d exp(.) times exp(.)
    want = float(d[f"{tag}_dscale"])
    assert abs(scale.grad.item() / 0.07 - want) < 1e-4 * max(1.0, abs(want))


def test_hard_negative_branch_random_vs_oracle():
    from oracle import clip_oracle as O
    from uniir_amd.losses import HardNegNCEFn
    torch.manual_seed(5)
    b, nneg, E, ibn = 24, 3, 512, 7
    M = b * (2 + nneg)
    emb_c = torch.randn(M, E, requires_grad=True)
    im = {"query": [], "pos_cand": [], "neg_cand_list": []}
    c = 0
    for i in range(b):
        im["query"].append([c]); c += 1
        im["pos_cand"].append([c]); c += 1
        im["neg_cand_list"].append(list(range(c, c + nneg))); c += nneg
    ref = O.inbatch_contrastive_loss(emb_c, im, torch.tensor(14.0), in_batch_neg_num=ibn)
    ref["loss"].backward()
    emb = emb_c.detach().cuda().requires_grad_(True)
    t = lambda x: torch.tensor(x, dtype=torch.int32, device="cuda").flatten()
    loss, acc = HardNegNCEFn.apply(emb, t(im["query"]), t(im["pos_cand"]), t(im["neg_cand_list"]),
                                   torch.tensor(14.0, device="cuda"), ibn)
    (2.0 * loss).backward()      # upstream gradient other than 1
    assert abs(loss.item() - ref["loss"].item()) < 2e-5
    assert acc.item() == ref["accuracy"].item()
    assert (emb.grad.cpu() - 2.0 * emb_c.grad).abs().max().item() < 2e-5
