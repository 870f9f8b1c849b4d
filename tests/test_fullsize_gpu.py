"""BASELINE.json's full sizes on the MI355X, checked through properties that do not need a CPU run of the whole problem
(SURVEY.md section 8c / the task's parity bar):
  * InfoNCE at config 2's per-rank shape (512 queries x 4096 gathered candidates x 768): the C oracle still finishes in
    seconds here, so the logits are compared bit for bit;
  * brute-force top-10 over one GPU's 700 k x 768 fp16 shard: order, id validity, planted neighbours at rank 1, every
    returned score re-derived bit-exactly by the C oracle on just the returned rows, shard-split + merge == single search;
    the same over the WHOLE 5.6 M x 768 pool resident on one GPU (logical sub-shards below the 2-GiB buffer bound);
  * the CLIP_SF ViT-L/14 encoder at 1024 items: row independence (a permuted batch gives the permuted embeddings, a
    64-item slice gives the same rows), unit norms after the loss's normalisation, and gradient linearity in the loss."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_infonce_full_rank_shape_bit_exact():
    from oracle import c_oracle
    from uniir_amd import ops
    b, B, E, toff = 512, 4096, 768, 1024          # rank 2 of 8: targets offset by rank * b
    rng = np.random.default_rng(7)
    q = rng.standard_normal((b, E)).astype(np.float32)
    p = rng.standard_normal((B, E)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    p /= np.linalg.norm(p, axis=1, keepdims=True)
    scale = float(np.float32(1 / 0.07))
    score = torch.empty(b, B, device=DEV)
    stats = torch.empty(3 * b, device=DEV)
    loss, acc = torch.empty(1, device=DEV), torch.empty(1, device=DEV)
    ops.call("uniir_infonce_fwd", torch.tensor(q, device=DEV), torch.tensor(p, device=DEV), torch.tensor([scale], device=DEV),
             b, B, E, toff, score, stats, loss, acc)
    want = c_oracle.infonce_scores(q, p, scale)
    assert np.array_equal(score.cpu().numpy(), want)
    l, a, _ = c_oracle.infonce_loss(want, toff)
    assert abs(loss.item() - l) < 2e-6 * max(1.0, abs(l)) and acc.item() == a


@pytest.mark.parametrize("nq,n", [(64, 700_000), (1024, 700_000), (128, 700_000), (256, 700_000), (200, 1_398_000), (256, 32_768),
                                  (130, 32_769)])
def test_topk_full_shard_properties(nq, n):
    """size-independent properties at BASELINE's shard size and at the limits of the streaming scans: 1 398 000 rows = the largest
    shard the 31-bit buffer bound addresses (2 147 328 000 bytes), 32 768 rows = the 2 048 groups they need at least (one more row:
    a ragged last tile); 64 queries = private rings, 128 / 130 / 200 / 256 = shared rings, 1024 = four 256-query sweeps"""
    from oracle import c_oracle
    from uniir_amd import retrieval
    d, k = 768, 10
    g = torch.Generator(device=DEV).manual_seed(11 + nq)
    pool = torch.randn(n, d, device=DEV, generator=g).half()
    queries = torch.randn(nq, d, device=DEV, generator=g).half()
    where = torch.randperm(n, device=DEV, generator=g)[:nq]
    pool[where] = (queries.float() * 3.0).half()              # planted: same direction, another norm -> cosine 1
    ids = torch.randperm(n, device=DEV, generator=g).to(torch.int64) + 5_000_000
    shard = retrieval.PoolShard(pool, ids)
    s, i = retrieval.search_shard(shard, queries, k)
    sc, ic = s.cpu().numpy(), i.cpu().numpy()
    assert (np.diff(sc, axis=1) <= 0).all()                                        # sorted, best first
    assert (ic >= 5_000_000).all() and all(len(set(r)) == k for r in ic.tolist())  # valid, unique
    assert np.array_equal(ic[:, 0], ids[where].cpu().numpy())                      # the planted row wins
    assert np.abs(sc[:, 0] - 1.0).max() < 1e-3
    # every returned (query, row) score is the oracle's exact fp32 value: re-run the oracle on the returned rows only
    row_of = torch.empty(n + 5_000_000, dtype=torch.int32, device=DEV)
    row_of[ids] = torch.arange(n, dtype=torch.int32, device=DEV)
    for qi in range(0, nq, max(1, nq // 16)):
        rows = row_of[i[qi]].long()
        ws, wi = c_oracle.topk(pool[rows].cpu().numpy(), ids[rows].cpu().numpy(), queries[qi:qi + 1].cpu().numpy(), k)
        assert np.array_equal(wi[0], ic[qi]) and np.array_equal(ws[0], sc[qi])
    # two half shards + merge == the single search, bit for bit
    h = n // 2
    a = retrieval.search_shard(retrieval.PoolShard(pool[:h], ids[:h]), queries, k)
    b = retrieval.search_shard(retrieval.PoolShard(pool[h:], ids[h:]), queries, k)
    ms, mi = retrieval.merge_shards(torch.stack([a[0], b[0]]), torch.stack([a[1], b[1]]))
    assert torch.equal(ms, s) and torch.equal(mi, i)


@pytest.mark.parametrize("nq", [64, 1024])
def test_topk_whole_mbeir_pool_on_one_gpu(nq):
    """BASELINE configs[3] as written on ONE GPU (mbeir_retriever.py:196-206 with a single visible device; README.md:126: 5.6 M
    candidates): a 5.6 M x 768 fp16 pool (8.6 GB) resident as ONE PoolShard.  retrieval.search_shard searches it as 8 logical
    sub-shards of 700 000 rows (below the 2-GiB buffer bound and within the fused tail's rows: round 5) + uniir_topk_merge; checked
    by the size-independent properties (order, valid unique ids, planted neighbours at rank 1, the C oracle's exact scores on the
    returned rows) and against the merge of 5 x 1.12 M-row searches (another partitioning of the same pool, which runs the UNFUSED
    tail: separate selection and re-score kernels), bit for bit"""
    from oracle import c_oracle
    from uniir_amd import retrieval
    n, d, k = 5_600_000, 768, 10
    g = torch.Generator(device=DEV).manual_seed(50 + nq)
    pool = torch.empty(n, d, device=DEV, dtype=torch.float16)
    for lo in range(0, n, 700_000):                           # generated in slices: no 17-GB fp32 temporary
        pool[lo:lo + 700_000] = torch.randn(700_000, d, device=DEV, generator=g).half()
    queries = torch.randn(nq, d, device=DEV, generator=g).half()
    where = torch.randperm(n, device=DEV, generator=g)[:nq]
    pool[where] = (queries.float() * 3.0).half()              # planted: same direction, another norm -> cosine 1
    ids = torch.arange(n, device=DEV, dtype=torch.int64) * 3 + 7
    shard = retrieval.PoolShard(pool, ids)
    bounds = retrieval.subshard_bounds(n, d)
    assert len(bounds) == 8 and all((hi - lo) * d * 2 < 2 ** 31 and lo % 32 == 0 for lo, hi in bounds) and bounds[-1][1] == n
    s, i = retrieval.search_shard(shard, queries, k)
    sc, ic = s.cpu().numpy(), i.cpu().numpy()
    assert (np.diff(sc, axis=1) <= 0).all()
    assert ((ic - 7) % 3 == 0).all() and (ic >= 7).all() and (ic < 3 * n + 7).all() and all(len(set(r)) == k for r in ic.tolist())
    assert np.array_equal(ic[:, 0], ids[where].cpu().numpy())
    assert np.abs(sc[:, 0] - 1.0).max() < 1e-3
    for qi in range(0, nq, max(1, nq // 8)):                  # the oracle's exact fp32 scores on the returned rows
        rows = (i[qi] - 7) // 3
        ws, wi = c_oracle.topk(pool[rows].cpu().numpy(), ids[rows].cpu().numpy(), queries[qi:qi + 1].cpu().numpy(), k)
        assert np.array_equal(wi[0], ic[qi]) and np.array_equal(ws[0], sc[qi])
    # no better row was missed: an independent fp32 scan of ALL 5.6 M rows (torch matmul on the device, slice by slice; not the oracle,
    # a second opinion that can afford the whole pool) -- the 11 best cosines per query; the returned k-th score must not lie below
    # the best score of any row that was NOT returned (up to the fp32 noise of two different summation orders)
    qs = min(nq, 64)
    qf = queries[:qs].float()
    qf = qf / qf.norm(dim=1, keepdim=True).clamp_min(1e-30)
    best_s = torch.full((qs, 0), 0.0, device=DEV)
    best_r = torch.zeros(qs, 0, dtype=torch.int64, device=DEV)
    for lo in range(0, n, 700_000):
        blk = pool[lo:lo + 700_000].float()
        blk = blk / blk.norm(dim=1, keepdim=True).clamp_min(1e-30)
        ts, tr = torch.topk(qf @ blk.t(), k + 1, dim=1)
        best_s, sel = torch.topk(torch.cat([best_s, ts], 1), k + 1, dim=1)
        best_r = torch.gather(torch.cat([best_r, tr + lo], 1), 1, sel)
        del blk
    ret_rows = (i[:qs] - 7) // 3
    for qi in range(qs):
        missed = [(float(v), int(r)) for v, r in zip(best_s[qi], best_r[qi]) if int(r) not in set(ret_rows[qi].tolist())]
        assert missed and missed[0][0] <= float(s[qi, k - 1]) + 2e-6, (qi, missed[0], float(s[qi, k - 1]))
        assert abs(float(best_s[qi, 0]) - float(s[qi, 0])) < 2e-6
    parts = [retrieval.search_shard(retrieval._shard_view(shard, lo, lo + 1_120_000), queries, k) for lo in range(0, n, 1_120_000)]
    ms, mi = retrieval.merge_shards(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
    assert torch.equal(ms, s) and torch.equal(mi, i)


def test_topk_700k_pool_of_clip_base_width():
    """a 700 k x 512 shard (the CLIP base models' embeddings, clip_sf.py with ViT-B/32: embed_dim 512) at BASELINE's shard size:
    64 queries on the streaming scan, checked by the size-independent properties, the C oracle on the returned rows, the merge of
    two half-shard searches, and the same queries inside a 200-query search (another scan kernel): identical rows"""
    from oracle import c_oracle
    from uniir_amd import retrieval
    n, d, k, nq = 700_000, 512, 10, 64
    g = torch.Generator(device=DEV).manual_seed(77)
    pool = torch.randn(n, d, device=DEV, generator=g).half()
    queries = torch.randn(200, d, device=DEV, generator=g).half()
    where = torch.randperm(n, device=DEV, generator=g)[:200]
    pool[where] = (queries.float() * 2.0).half()
    ids = torch.arange(n, device=DEV, dtype=torch.int64) * 3 + 7
    shard = retrieval.PoolShard(pool, ids)
    s, i = retrieval.search_shard(shard, queries[:nq], k)
    sc, ic = s.cpu().numpy(), i.cpu().numpy()
    assert (np.diff(sc, axis=1) <= 0).all() and all(len(set(r)) == k for r in ic.tolist())
    assert np.array_equal(ic[:, 0], ids[where[:nq]].cpu().numpy()) and np.abs(sc[:, 0] - 1.0).max() < 1e-3
    for qi in range(0, nq, 8):
        rows = (i[qi] - 7) // 3
        ws, wi = c_oracle.topk(pool[rows].cpu().numpy(), ids[rows].cpu().numpy(), queries[qi:qi + 1].cpu().numpy(), k)
        assert np.array_equal(wi[0], ic[qi]) and np.array_equal(ws[0], sc[qi])
    half = 350_000
    a = retrieval.search_shard(retrieval._shard_view(shard, 0, half), queries[:nq], k)
    b = retrieval.search_shard(retrieval._shard_view(shard, half, n), queries[:nq], k)
    ms, mi = retrieval.merge_shards(torch.stack([a[0], b[0]]), torch.stack([a[1], b[1]]))
    assert torch.equal(ms, s) and torch.equal(mi, i)
    s2, i2 = retrieval.search_shard(shard, queries, k)
    assert torch.equal(s2[:nq], s) and torch.equal(i2[:nq], i)


def test_topk_c_abi_refuses_shards_of_2_gib():
    """uniir_topk_ip addresses a shard through 31-bit buffer offsets: rows * dim * 2 >= 2 GiB is UNIIR_ESHAPE at the C ABI (never
    a silent slower path); 1 398 096 rows x 768 (the largest single shard retrieval.subshard_bounds leaves whole) is accepted"""
    from uniir_amd import _lib, retrieval
    lib = _lib.load()
    d, k, nq = 768, 10, 8
    n_ok = (2 ** 31 - 1) // (d * 2) // 16 * 16
    assert n_ok == 1_398_096 and retrieval.subshard_bounds(n_ok, d) == [(0, n_ok)]
    assert len(retrieval.subshard_bounds(n_ok + 16, d)) == 2
    n_bad = 1_500_000
    pool = torch.zeros(n_bad, d, device=DEV, dtype=torch.float16)
    ids = torch.arange(n_bad, device=DEV)
    inv = torch.ones(n_bad, device=DEV)
    q = torch.randn(nq, d, device=DEV).half()
    out_s, out_i = torch.empty(nq, k, device=DEV), torch.empty(nq, k, device=DEV, dtype=torch.int64)
    ws = torch.empty(lib.uniir_topk_ip_workspace_bytes(nq, k, n_bad), device=DEV, dtype=torch.uint8)
    rc = lib.uniir_topk_ip(pool.data_ptr(), inv.data_ptr(), ids.data_ptr(), n_bad, d, q.data_ptr(), nq, k, out_s.data_ptr(),
                           out_i.data_ptr(), ws.data_ptr(), ws.numel(), None)
    assert rc == -2                                           # UNIIR_ESHAPE
    rc = lib.uniir_topk_ip(pool.data_ptr(), inv.data_ptr(), ids.data_ptr(), n_ok, d, q.data_ptr(), nq, k, out_s.data_ptr(),
                           out_i.data_ptr(), ws.data_ptr(), ws.numel(), None)
    assert rc == 0
    torch.cuda.synchronize()


def test_vit_l14_encoder_row_independence_and_gradient_linearity():
    import os
    import sys
    from types import SimpleNamespace
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "uniir_amd", "src"))
    from models.uniir_clip.clip_scorefusion.clip_sf import CLIPScoreFusion
    from uniir_amd.clip_model import CLIP_CONFIGS
    cfg = CLIP_CONFIGS["ViT-L/14"]
    config = SimpleNamespace(model=SimpleNamespace(gather_embeddings=False), data_config=SimpleNamespace(in_batch_neg_num=0))
    torch.manual_seed(5)
    model = CLIPScoreFusion(model_name="ViT-L/14", device=DEV, config=config)
    model.float()
    M = 1024
    g = torch.Generator().manual_seed(9)
    txt = torch.randint(1, cfg["vocab_size"] - 2, (M, 77), generator=g, dtype=torch.int32)
    eot = torch.randint(3, 77, (M,), generator=g)
    txt[torch.arange(M), eot] = cfg["vocab_size"] - 1                      # EOT = arg-max token id
    for r in range(M):
        txt[r, eot[r] + 1:] = 0
    img = torch.randn(M, 3, 224, 224, generator=torch.Generator(device=DEV).manual_seed(9), device=DEV)
    txt = txt.to(DEV)
    ones = torch.ones(M, dtype=torch.long, device=DEV)
    with torch.no_grad():
        emb = model.encode_multimodal_input(txt, img, ones, ones)
        assert emb.shape == (M, 768) and torch.isfinite(emb).all()
        perm = torch.randperm(M, generator=g).to(DEV)
        emb_p = model.encode_multimodal_input(txt[perm], img[perm], ones, ones)
        assert torch.equal(emb_p, emb[perm])                               # rows do not see each other
        emb_s = model.encode_multimodal_input(txt[:64], img[:64], ones[:64], ones[:64])
        assert (emb_s - emb[:64]).abs().max().item() <= 1e-6 * emb.abs().max().item()
    # gradient linearity: d(2 L) = 2 dL (powers of two commute with every rounding; fp32 atomics reorder -> 1e-5)
    batch = {"txt_batched": txt[:128], "image_batched": img[:128], "txt_mask_batched": ones[:128], "image_mask_batched": ones[:128],
             "index_mapping": {"query": [[2 * j] for j in range(64)], "pos_cand": [[2 * j + 1] for j in range(64)]}}
    grads = []
    for scale in (1.0, 2.0):
        model.zero_grad()
        out = model(batch)
        (out["loss"] * scale).backward()
        grads.append({n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    for name in ("clip_model.visual.proj", "clip_model.visual.transformer.resblocks.0.attn.in_proj_weight",
                 "clip_model.transformer.resblocks.11.mlp.c_fc.weight", "clip_model.token_embedding.weight",
                 "clip_model.visual.transformer.resblocks.23.ln_2.weight"):
        a, b = grads[0][name], grads[1][name]
        assert a.abs().max().item() > 0 and (b - 2 * a).abs().max().item() <= 1e-5 * b.abs().max().item(), name
    # round 6, at the real shapes (24 x 1024-wide blocks of 257 tokens, 12 x 768-wide blocks of packed captions, 256 items = one
    # 256-row GEMM panel of pooled rows): the last block on the pooled rows (clip_model.pool_last_block) against every row through
    # every sublayer -- embeddings and loss bitwise, every parameter gradient up to the order of the fp32 additions; and two runs of
    # the same backward give the same bits (reproducible reductions)
    clip = model.clip_model
    assert clip.pool_last_block
    big = {"txt_batched": txt[:256], "image_batched": img[:256], "txt_mask_batched": ones[:256], "image_mask_batched": ones[:256],
           "index_mapping": {"query": [[2 * j] for j in range(128)], "pos_cand": [[2 * j + 1] for j in range(128)]}}
    res = {}
    for tag, pooled in (("pooled", True), ("pooled again", True), ("full", False)):
        clip.pool_last_block = pooled
        model.zero_grad()
        with torch.no_grad():
            e = model.encode_multimodal_input(txt[:256], img[:256], ones[:256], ones[:256])
        out = model(big)
        out["loss"].backward()
        res[tag] = (e.clone(), out["loss"].detach().clone(), clip._flat["g32"].clone())
    clip.pool_last_block = True
    assert torch.equal(res["pooled"][0], emb[:256]) or (res["pooled"][0] - emb[:256]).abs().max().item() <= 1e-6 * emb.abs().max().item()
    assert torch.equal(res["pooled"][0], res["full"][0]) and torch.equal(res["pooled"][1], res["full"][1])
    assert torch.equal(res["pooled"][2], res["pooled again"][2])
    gp, gf = res["pooled"][2], res["full"][2]
    # (at 257 tokens the full block's attention backward is the persistent pair-tile kernel, the pooled block's one-query backward the
    # general kernel: two valid bf16 kernels whose dK / dV differ in the last bf16 bits, which every earlier layer's gradient
    # inherits -- 2e-3 relative observed; the last block's own parameters, which see no attention gradient from that layer, 1e-5.
    # The tiny configuration of tests/test_clip_model_gpu.py runs the general kernel in both forms and holds 1e-5 everywhere.)
    fl = clip._flat
    worst = (0.0, "")
    for n, off in fl["off"].items():
        k = 1
        for dmn in fl["shapes"][n]:
            k *= dmn
        a, b = gp[off:off + k], gf[off:off + k]
        den = float(b.norm())
        r = 0.0 if den == 0.0 else float((a - b).norm()) / den
        worst = max(worst, (r, n))
        own = any(t in n for t in ("resblocks.23.mlp", "resblocks.23.ln_2", "resblocks.23.attn.out_proj", "visual.proj", "ln_post",
                                   "transformer.resblocks.11.mlp", "text_projection", "ln_final")) and "visual.transformer.resblocks.11" not in n
        assert r <= (1e-5 if own else 1e-2), (n, r)
    print("OBS pooled vs full last block, worst parameter gradient", worst)


def test_config1_vit_b32_step_against_the_oracle():
    """BASELINE.json configs[0] (the reference's own CPU-runnable case): real CLIP ViT-B/32 dimensions, a few pairs so
    that the fp32 oracle's forward + backward finishes in seconds on the host; embeddings, loss and every parameter
    gradient against the oracle, gates at ~2x the observed bf16 error (1.2e-2 / 5e-3 / 1e-1 relative L2, gradient
    cosine > 0.9995); the same configuration at batch 32 in fp32 meets the north-star 1e-3 in tests/test_fp32_parity_gpu.py"""
    import os
    import sys
    from types import SimpleNamespace
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "uniir_amd", "src"))
    from oracle import clip_oracle as O
    from models.uniir_clip.clip_scorefusion.clip_sf import CLIPScoreFusion
    from uniir_amd.clip_model import CLIP_CONFIGS
    cfg = CLIP_CONFIGS["ViT-B/32"]
    sd = O.init_state_dict(cfg, seed=1)
    config = SimpleNamespace(model=SimpleNamespace(gather_embeddings=False), data_config=SimpleNamespace(in_batch_neg_num=0))
    model = CLIPScoreFusion("ViT-B/32", device=DEV, config=config)
    model.float()
    model.clip_model.load_state_dict(sd, strict=True)
    oracle = O.OracleCLIP(cfg, sd)
    pairs = 4
    batch = O.synthetic_batch(cfg, pairs, seed=21)
    dbatch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    emb_o = O.encode_multimodal_input(oracle.sd(), cfg, batch["txt_batched"], batch["image_batched"],
                                      batch["txt_mask_batched"], batch["image_mask_batched"])
    out_o = O.inbatch_contrastive_loss(emb_o, batch["index_mapping"], oracle.logit_scale.exp())
    out_o["loss"].backward()
    model.train()
    model.zero_grad()
    emb_d = model.encode_multimodal_input(dbatch["txt_batched"], dbatch["image_batched"], dbatch["txt_mask_batched"],
                                          dbatch["image_mask_batched"])

    def rel(a, b):
        a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
        return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()

    print("OBS b32 emb rel", rel(emb_d, emb_o))
    assert rel(emb_d, emb_o) < 1.2e-2, rel(emb_d, emb_o)          # observed 5.9e-3
    out_d = model(dbatch)
    print("OBS b32 loss diff", abs(out_d["loss"].item() - out_o["loss"].item()))
    assert abs(out_d["loss"].item() - out_o["loss"].item()) < 5e-3 * max(1.0, abs(out_o["loss"].item()))    # observed 1.0e-3
    assert out_d["accuracy"].item() == out_o["accuracy"].item()
    out_d["loss"].backward()
    errs, gd, go = {}, [], []
    for n, p in model.clip_model.named_parameters():
        g = getattr(oracle, n.replace(".", "__")).grad
        if g is None:
            continue
        errs[n] = rel(p.grad, g)
        gd.append(p.grad.flatten().cpu())
        go.append(g.flatten())
    print("OBS b32 worst grad", max(errs.values()), "cos", torch.nn.functional.cosine_similarity(torch.cat(gd), torch.cat(go), dim=0).item())
    big = {n: e for n, e in errs.items() if e > 1e-1}           # observed worst 6.5e-2 (a bias gradient summed over 8 items)
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
    print("OBS b32 device error / bf16-oracle noise: worst", worst16, ratio[worst16], "median",
          sorted(r[2] for r in ratio.values())[len(ratio) // 2])
    # observed: median 0.87-0.90 at ViT-B/32 and ViT-L/14 (the device IS the bf16-rounded computation), worst 1.07 (L/14) / 2.8
    # (B/32, the scalar logit_scale: one number, no averaging)
    bad = {n: r for n, r in ratio.items() if r[2] > 4.0}
    assert not bad, bad
    assert sorted(r[2] for r in ratio.values())[len(ratio) // 2] < 1.5


@pytest.mark.parametrize("n,d", [(1_500_032, 768), (1_500_016, 768), (2_200_000, 512)])
def test_topk_ip_multi_equals_the_merge_of_sub_shard_searches(n, d):
    """uniir_topk_ip_multi (round 5): one resident shard above the 2-GiB buffer bound is searched in ONE C call -- a scan per logical
    sub-shard, then one batched tail, one sort, one merge launch -- and must equal the merge of stand-alone searches of the same row
    ranges bit for bit; 64 queries (interactive scan) and 300 (two 256-query sweeps).  n = 1 500 016 leaves a last sub-shard with an
    odd number of 16-row groups, which the batched tail does not take: the entry point's per-sub-shard fallback loop runs instead.
    2.2 M x 512 (the CLIP base models' width, 3 sub-shards): batched at 64 queries, the fallback loop at 300 (no shared-ring scan at
    this width)."""
    from uniir_amd import _lib, retrieval
    k = 10
    lib = _lib.load()
    per = int(lib.uniir_topk_subshard_rows(n, d))
    bounds = retrieval.subshard_bounds(n, d)
    assert len(bounds) == (2 if d == 768 else 3) and bounds[0] == (0, per) and per % 32 == 0
    if d == 768:
        assert ((n - per + 15) // 16) % 2 == (0 if n == 1_500_032 else 1)
    assert int(lib.uniir_topk_subshard_rows(5_600_000, 768)) == 700_000 and int(lib.uniir_topk_subshard_rows(700_000, 768)) == 700_000
    g = torch.Generator(device=DEV).manual_seed(n % 1000)
    pool = torch.empty(n, d, device=DEV, dtype=torch.float16)
    for lo in range(0, n, 500_000):
        m = min(500_000, n - lo)
        pool[lo:lo + m] = torch.randn(m, d, device=DEV, generator=g).half()
    ids = torch.randperm(n, device=DEV, generator=g).to(torch.int64) * 5 + 1
    shard = retrieval.PoolShard(pool, ids)
    for nq in (64, 300):
        queries = torch.randn(nq, d, device=DEV, generator=g).half()
        where = torch.randperm(n, device=DEV, generator=g)[:nq]
        pool[where] = (queries.float() * 2.0).half()
        shard = retrieval.PoolShard(pool, ids)
        s, i = retrieval.search_shard(shard, queries, k)
        parts = [retrieval.search_shard(retrieval._shard_view(shard, lo, hi), queries, k) for lo, hi in bounds]
        ms, mi = retrieval.merge_shards(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
        assert torch.equal(s, ms) and torch.equal(i, mi), nq
        assert torch.equal(i[:, 0], ids[where])
