"""GPU parity of the brute-force top-k against a torch fp32 restatement of FAISS IDMap,Flat IP semantics
(normalize rows, exact inner product, top-k descending).  Ids must be identical wherever the reference's own
score gap exceeds fp32 noise; planted neighbours make that explicit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ref(pool, ids, queries, k):
    p = torch.nn.functional.normalize(pool.double(), dim=-1)
    q = torch.nn.functional.normalize(queries.double(), dim=-1)
    s = q @ p.t()
    v, i = torch.topk(s, min(k, pool.shape[0]), dim=-1)
    return v, ids[i]


@pytest.mark.parametrize("n,nq,d,k", [(1000, 7, 64, 10), (5000, 130, 512, 10), (70000, 33, 768, 10), (20000, 300, 768, 50), (5, 3, 64, 10), (3000, 2300, 256, 10)])
def test_topk_matches_reference(n, nq, d, k):
    from uniir_amd import retrieval
    torch.manual_seed(0)
    pool = torch.randn(n, d, device=DEV).half()
    queries = torch.randn(nq, d, device=DEV).half()
    if n > 100:
        # planted neighbours: query j ~ pool row 10*j+3
        for j in range(min(nq, 20)):
            queries[j] = (pool[10 * j + 3].float() * 2.0 + 0.05 * torch.randn(d, device=DEV)).half()
        pool[17] = 0  # zero row must score 0 and not break anything
    ids = (torch.randperm(n, device=DEV) * 7 + 1000).long()
    shard = retrieval.PoolShard(pool, ids)
    s, i = retrieval.search_shard(shard, queries, k)
    v, ri = _ref(pool, ids, queries, k)
    kk = v.shape[1]
    assert (s[:, :kk].double() - v).abs().max() < 2e-6
    # ids identical except where the double-precision reference itself has a near-tie (< 1e-6 gap)
    gap_ok = torch.ones_like(ri, dtype=torch.bool)
    gap_ok[:, 1:] &= (v[:, :-1] - v[:, 1:]) > 1e-6
    gap_ok[:, :-1] &= (v[:, :-1] - v[:, 1:]) > 1e-6
    mism = (i[:, :kk] != ri) & gap_ok
    assert not mism.any(), mism.nonzero()[:5]
    if kk < k:
        assert (i[:, kk:] == -1).all()
    # descending
    assert (s[:, :kk - 1] >= s[:, 1:kk]).all()


def test_topk_sharded_equals_unsharded():
    from uniir_amd import retrieval
    torch.manual_seed(1)
    n, nq, d, k = 40000, 64, 768, 10
    pool = torch.randn(n, d, device=DEV).half()
    queries = torch.randn(nq, d, device=DEV).half()
    ids = torch.arange(n, device=DEV) + 5
    full = retrieval.search_shard(retrieval.PoolShard(pool, ids), queries, k)
    parts = []
    bounds = [0, 13000, 13000 + 9001, n]
    for a, b in zip(bounds[:-1], bounds[1:]):
        parts.append(retrieval.search_shard(retrieval.PoolShard(pool[a:b], ids[a:b]), queries, k))
    ms, mi = retrieval.merge_shards(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
    assert torch.equal(mi, full[1])
    assert torch.equal(ms, full[0])


def test_topk_duplicates_tiebreak():
    from uniir_amd import retrieval
    torch.manual_seed(2)
    n, d, k = 3000, 64, 10
    pool = torch.randn(n, d, device=DEV).half()
    pool[100:106] = pool[50]  # 6 exact duplicates of row 50
    ids = torch.arange(n, device=DEV).flip(0).contiguous()  # descending ids: tie-break must use ids, not rows
    q = pool[50:51].clone()
    s, i = retrieval.search_shard(retrieval.PoolShard(pool, ids), q, k)
    dup_ids = sorted(ids[[50, 100, 101, 102, 103, 104, 105]].tolist())
    assert i[0, :7].tolist() == dup_ids
    assert (s[0, :7] == s[0, 0]).all()


def test_embed_resident_then_search_without_leaving_hbm():
    """embedder <-> retriever fusion on one rank: the embedder loop keeps fp16 rows + hashed ids on the device, the shard is
    searched in place; equals the C oracle on the same fp16 rows"""
    from oracle import c_oracle
    from uniir_amd import retrieval
    g = torch.Generator().manual_seed(3)
    table = torch.randn(500, 128, generator=g)

    class Fake(torch.nn.Module):
        def forward(self, batch, encode_mbeir_batch=False):
            assert encode_mbeir_batch and batch["x"].is_cuda
            return batch["x"] * 1.5, batch["did_list"]

    loader = [{"x": table[lo:lo + 64], "did_list": list(range(9_000_001 + lo, 9_000_001 + min(500, lo + 64)))}
              for lo in range(0, 500, 64)]
    emb, ids = retrieval.embed_resident(Fake(), loader, torch.device("cuda"))
    assert emb.dtype == torch.float16 and emb.is_cuda and emb.shape == (500, 128) and ids.dtype == torch.int64
    assert ids.tolist() == list(range(9_000_001, 9_000_501))
    shard = retrieval.PoolShard(emb, ids)
    queries = torch.randn(21, 128, generator=g).half().cuda()
    s, i = retrieval.search_resident(shard, queries, 10)
    want_s, want_i = c_oracle.topk(emb.cpu().numpy(), ids.cpu().numpy(), queries.cpu().numpy(), 10)
    assert np.array_equal(i.cpu().numpy(), want_i) and np.array_equal(s.cpu().numpy(), want_s)


def _two_call_search(shard, queries, k):
    """the round-1 call sequence (uniir_topk_coarse -> gsel / rescore_coalesced / final_sort kernels) through the C ABI"""
    from uniir_amd import _lib, ops, retrieval
    nq, dev = queries.shape[0], queries.device
    kc = k + retrieval.COARSE_MARGIN
    ws = torch.empty(_lib.load().uniir_topk_workspace_bytes(nq, kc, shard.n), device=dev, dtype=torch.uint8)
    ncand = _lib.load().uniir_topk_ncand(nq, kc)
    cand = torch.empty(nq, ncand, device=dev, dtype=torch.int32)
    cs = torch.empty(nq, kc, device=dev)
    ops.call("uniir_topk_coarse", shard.emb, shard.inv_norm, shard.n, shard.dim, queries, nq, kc, cand, cs, ws, ws.numel())
    exact = torch.empty(nq, ncand, device=dev)
    out_s, out_i = torch.empty(nq, k, device=dev), torch.empty(nq, k, device=dev, dtype=torch.int64)
    ops.call("uniir_topk_rescore", shard.emb, shard.inv_norm, shard.ids, shard.n, shard.dim, queries,
             retrieval.query_inv_norms(queries), nq, cand, ncand, k, exact, out_s, out_i)
    return out_s, out_i


@pytest.mark.parametrize("n,nq,d,k", [(70001, 64, 768, 10), (30000, 7, 512, 50), (9000, 300, 256, 10), (40, 5, 64, 10),
                                      (70000, 1100, 768, 10), (280030, 64, 768, 10), (262144, 9, 768, 20)])
def test_single_call_search_equals_the_op_level_sequence_bit_for_bit(n, nq, d, k):
    """uniir_topk_ip (one call: query norms + per-chunk scan, selection, re-score, sort) == the op-level entry points called
    one by one, scores and ids, incl. zero rows, duplicate ties, a ragged last group, -1 padding and > 1024 queries"""
    from uniir_amd import retrieval
    torch.manual_seed(11)
    pool = torch.randn(n, d, device=DEV).half()
    pool[3] = 0
    if n > 200:
        pool[100:104] = pool[50]
    queries = torch.randn(nq, d, device=DEV).half()
    queries[0] = pool[50] if n > 200 else queries[0]
    queries[1] = 0
    ids = (torch.randperm(n, device=DEV) * 3 + 17).long()
    shard = retrieval.PoolShard(pool, ids)
    s1, i1 = retrieval.search_shard(shard, queries, k)
    s2, i2 = _two_call_search(shard, queries[:1024].contiguous(), k)
    assert torch.equal(i1[:1024], i2) and torch.equal(s1[:1024], s2)


@pytest.mark.parametrize("k", [57, 100, 200])
def test_large_k_is_exact_also_when_one_slice_holds_most_of_the_top_k(k):
    """k > 56 (FAISS accepts up to 2048): assembled from row slices with the hide-check; a planted cluster of 150
    near-duplicates of query 0 inside one slice forces the refinement loop.  Bit-exact against the C oracle."""
    from oracle import c_oracle
    from uniir_amd import retrieval
    g = np.random.default_rng(3)
    n, d, nq = 6000, 128, 9
    pool = g.standard_normal((n, d)).astype(np.float16)
    queries = g.standard_normal((nq, d)).astype(np.float16)
    pool[1000:1150] = (queries[0].astype(np.float32) * 1.5 + 0.05 * g.standard_normal((150, d))).astype(np.float16)
    ids = (g.permutation(n) + 10_000).astype(np.int64)
    want_s, want_i = c_oracle.topk(pool, ids, queries, k)
    shard = retrieval.PoolShard(torch.from_numpy(pool).to(DEV), torch.from_numpy(ids))
    s, i = retrieval.search_shard(shard, torch.from_numpy(queries).to(DEV), k)
    assert np.array_equal(i.cpu().numpy(), want_i) and np.array_equal(s.cpu().numpy(), want_s)


def test_large_k_beyond_the_pool_pads_like_faiss():
    from uniir_amd import retrieval
    torch.manual_seed(4)
    pool = torch.randn(70, 64, device=DEV).half()
    s, i = retrieval.search_shard(retrieval.PoolShard(pool, torch.arange(70)), torch.randn(3, 64, device=DEV).half(), 90)
    assert (i[:, :70] >= 0).all() and (i[:, 70:] == -1).all() and torch.isinf(s[:, 70:]).all()
    assert all(sorted(r.tolist()) == list(range(70)) for r in i[:, :70])


def _oracle_topk_mt(pool, ids, queries, k):
    """oracle.c top-k with the queries split over host threads (ctypes releases the GIL)"""
    import os
    from concurrent.futures import ThreadPoolExecutor
    from oracle import c_oracle
    p, i, q = pool.cpu().numpy(), ids.cpu().numpy(), queries.cpu().numpy()
    nthr = max(1, min(32, os.cpu_count() or 1, q.shape[0]))
    parts = np.array_split(np.arange(q.shape[0]), nthr)
    with ThreadPoolExecutor(nthr) as ex:
        outs = list(ex.map(lambda idx: c_oracle.topk(p, i, q[idx], k), parts))
    return np.concatenate([o[0] for o in outs]), np.concatenate([o[1] for o in outs])


@pytest.mark.parametrize("n,nq,k", [(40030, 1, 10), (40030, 37, 10), (65536, 64, 10), (40003, 64, 10), (40030, 64, 50),
                                    (40030, 100, 10), (40030, 128, 10), (40030, 200, 10), (40030, 300, 50), (40030, 700, 10),
                                    (280030, 64, 10), (280030, 5, 10), (262144, 33, 24),
                                    (40003, 65, 10), (40030, 129, 10), (40030, 256, 10), (280030, 192, 10), (280030, 128, 24)])
def test_round3_scan_and_fused_tail_equal_the_c_oracle(n, nq, k):
    """round 3 kernels against oracle.c (reference mbeir_retriever.py:188-232), scores bit-exact and ids identical:
    <= 64 queries: topk_stream2_kernel (queries in registers, pool by LDS-DMA; needs >= 2048 groups) incl. a ragged last tile
    (40030 = 2501 x 16 + 14) and an odd group count (40003 -> 2501 groups: the round-2 tail behind the new scan);
    65..256 queries: topk_stream5_kernel (the pool ring shared by 2 / 4 waves of 64 register-resident queries each, rolling
    fragment registers, MFMAs with AGPR operands; partly filled last wave: 65, 100, 129, 192, 200 queries);
    > 256: the ping-pong GEMM scan; behind all of them the fused tail
    (selection + query norm + exact re-score in one launch with 4 / 2 / 1 workgroups per query, rank-count sort in the second),
    also at k = 50 (116 groups = 1 856 re-score slots per query, several 512-slot rounds per workgroup)."""
    from uniir_amd import retrieval
    g = torch.Generator(device=DEV).manual_seed(1000 + n + nq)
    pool = torch.randn(n, 768, device=DEV, generator=g).half()
    pool[17] = 0                                  # zero row: inverse norm 0, score 0
    pool[n - 1] = pool[5]                         # duplicates incl. the very last row (ragged tile): ties broken by id
    pool[2000:2003] = pool[5]
    if n > 100000:       # 16 more copies of one row, spread over the whole shard: one query's top hits all tie (21 tied groups: more
        where = torch.randperm(n, device=DEV, generator=g)[:16]      # than kc = 18, fewer than the 36 the selection keeps)
        pool[where] = pool[5].clone().repeat(16, 1)
    queries = torch.randn(nq, 768, device=DEV, generator=g).half()
    queries[0] = pool[5]
    ids = torch.randperm(n, device=DEV, generator=g).to(torch.int64) * 5 + 123
    s, i = retrieval.search_shard(retrieval.PoolShard(pool, ids), queries, k)
    ws, wi = _oracle_topk_mt(pool, ids, queries, k)
    assert np.array_equal(i.cpu().numpy(), wi), np.argwhere(i.cpu().numpy() != wi)[:5]
    assert np.array_equal(s.cpu().numpy(), ws)


@pytest.mark.parametrize("n,nq,k", [(40030, 1, 10), (40003, 64, 10), (65536, 33, 24), (280030, 64, 10), (40030, 64, 50), (40030, 100, 10)])
def test_clip_base_width_pools_on_the_streaming_scan_equal_the_c_oracle(n, nq, k):
    """dim 512 (the CLIP base models' embed_dim: BASELINE configs[0] / clip_sf.py with ViT-B/32) on topk_stream2_kernel<16> (round 4:
    the register-resident streaming scan templated on dim / 32; before, 512-wide pools took the first-generation scan): scores
    bit-exact and ids identical to oracle.c, with a zero row, duplicates incl. the last row of a ragged tile, an odd group count,
    k = 50, and -- 100 queries -- the ping-pong scan the wider sweeps of such a pool still take"""
    from uniir_amd import retrieval
    g = torch.Generator(device=DEV).manual_seed(4000 + n + nq)
    pool = torch.randn(n, 512, device=DEV, generator=g).half()
    pool[17] = 0
    pool[n - 1] = pool[5]
    pool[2000:2003] = pool[5]
    if n > 100000:
        where = torch.randperm(n, device=DEV, generator=g)[:16]
        pool[where] = pool[5].clone().repeat(16, 1)
    queries = torch.randn(nq, 512, device=DEV, generator=g).half()
    queries[0] = pool[5]
    ids = torch.randperm(n, device=DEV, generator=g).to(torch.int64) * 5 + 123
    s, i = retrieval.search_shard(retrieval.PoolShard(pool, ids), queries, k)
    ws, wi = _oracle_topk_mt(pool, ids, queries, k)
    assert np.array_equal(i.cpu().numpy(), wi), np.argwhere(i.cpu().numpy() != wi)[:5]
    assert np.array_equal(s.cpu().numpy(), ws)



@pytest.mark.parametrize("nq,spread", [(8, 0.0), (64, 3e-6), (64, 3e-5), (200, 2e-4), (8, 2e-3)])
def test_topk_candidate_bound_against_adversarial_near_ties(nq, spread):
    """The fused tail keeps every group whose maximum lies within the PROVEN rounding bound of the k-th best group maximum
    (topk_select.h GselBound: 5e-4 |q| in the scan's units) instead of a fixed k + 8 groups.  Adversarial pools for it: 30 rows, each
    alone in a different 16-row group, whose cosines with a query differ by less than the scan's rounding error (spread 0 ... 3e-5:
    the approximate ranking among them is noise, all of them must be re-scored -- more than the old k + 8 = 18), around the bound
    (2e-4) and clearly apart (2e-3: the rule may now drop most of them).  Distances and ids must equal the C oracle's, bit for bit."""
    from oracle import c_oracle
    from uniir_amd import retrieval
    n, d, k = 40_000, 768, 10
    g = torch.Generator(device=DEV).manual_seed(int(spread * 1e7) + nq)
    pool = torch.randn(n, d, device=DEV, generator=g) * 0.3
    queries = torch.randn(nq, d, device=DEV, generator=g)
    qn = queries / queries.norm(dim=1, keepdim=True)
    groups = torch.randperm(n // 16 - 1, device=DEV, generator=g)[:30] * 16 + 3          # 30 rows, each in its own group
    for qi in range(min(nq, 8)):                       # near-tie clusters for the first queries
        rows = (groups + qi) % n
        noise = torch.randn(30, d, device=DEV, generator=g)
        noise = noise - (noise @ qn[qi])[:, None] * qn[qi][None, :]
        noise = noise / noise.norm(dim=1, keepdim=True)
        cos = 0.9 - spread * torch.arange(30, device=DEV).float()              # cosines 0.9, 0.9 - spread, ...
        pool[rows] = (cos[:, None] * qn[qi][None, :] + (1 - cos * cos).sqrt()[:, None] * noise) * (1.0 + 0.1 * qi)
    pool16, q16 = pool.half(), queries.half()
    ids = torch.randperm(n, device=DEV, generator=g).to(torch.int64) + 11
    s, i = retrieval.search_shard(retrieval.PoolShard(pool16, ids), q16, k)
    ws, wi = c_oracle.topk(pool16.cpu().numpy(), ids.cpu().numpy(), q16.cpu().numpy(), k)
    assert np.array_equal(i.cpu().numpy(), wi) and np.array_equal(s.cpu().numpy(), ws)
