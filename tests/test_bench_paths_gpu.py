"""Checkers for paths that so far only ran inside bench.py ("what's weak"):
  (i)   the BLIP_FF [b, b+K] queue logits and their backward at the REAL queue size K = 57 344, b = 256 -- the strided two-block
        uniir_sgemm / uniir_sgemm_acc calls of blip_model._SoftTargetLossFn (reference blip_ff.py:219-229) take the 128-tile
        sgemm128_kernel there (<A k-contiguous, B n-contiguous> forward, <k-contiguous, k-contiguous> backward); every other
        BLIP test uses K = 16 (64-tile kernel).  Bit-exact against oracle.c's fmaf chains.
  (ii)  retrieval.search_shard's multi-sweep query loop (uniir_topk_ip chunks of TKI_CHUNK queries): forced to 64-query sweeps,
        1 000 queries x 20 000 rows == the C oracle (reference mbeir_retriever.py:188-232).
  (iii) the modality masks of encode_multimodal_input (clip_sf.py:53-63: image-only / text-only / pair = BASELINE config 3's three
        cases) on the HEADLINE architecture, ViT-L/14, forward-only (the embedding-extraction path: no stash, UNIIR_EPI_ACT_ONLY)."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "uniir_amd", "src"))


def _oracle_chains(q, p):
    """oracle.c fmaf chains score[i][j] = sum_k q[i][k] p[j][k] (k ascending), rows split over host threads (ctypes drops the GIL)"""
    from oracle import c_oracle
    q, p = np.ascontiguousarray(q, np.float32), np.ascontiguousarray(p, np.float32)
    nthr = min(32, os.cpu_count() or 1, q.shape[0])
    parts = np.array_split(np.arange(q.shape[0]), nthr)
    with ThreadPoolExecutor(nthr) as ex:
        outs = list(ex.map(lambda idx: c_oracle.infonce_scores(q[idx], p, 1.0), parts))
    return np.concatenate(outs, axis=0)


@pytest.mark.parametrize("hn", [0, 512])
def test_blip_queue_logits_and_backward_at_the_real_queue_size(hn):
    """sims(a, blocks) / dfeat(dsim, blocks) of blip_model._SoftTargetLossFn restated call for call at K = 57 344, b = 256,
    E = 768; hn > 0 is the hard-negative variant whose queue block starts at column hn (t[:, hn:], blip_ff.py:219-229 with the
    hard negatives' columns taken out of the queue)."""
    from uniir_amd import ops
    b, E, K = 256, 768, 57_344
    g = torch.Generator(device=DEV).manual_seed(5 + hn)
    a = torch.nn.functional.normalize(torch.randn(b, E, device=DEV, generator=g), dim=1)
    rows = torch.nn.functional.normalize(torch.randn(b, E, device=DEV, generator=g), dim=1)
    queue = torch.nn.functional.normalize(torch.randn(E, K, device=DEV, generator=g), dim=0)        # the reference's [E, K] buffer
    negs = torch.randn(hn, E, device=DEV, generator=g) if hn else None
    n = b + K                                       # [rows | (negatives) | queue[:, hn:]] is always b + K columns wide
    blocks = [("rows", rows, b)] + ([("rows", negs, hn)] if hn else []) + [("queue", queue, hn)]
    out = torch.full((b, n), float("nan"), device=DEV)
    col = 0
    for kind, t, arg in blocks:
        if kind == "rows":
            ops.call("uniir_sgemm", a, E, 1, t, 1, E, out[:, col:], n, b, arg, E, 1.0)
            col += arg
        else:
            ops.call("uniir_sgemm", a, E, 1, t[:, arg:], K, 1, out[:, col:], n, b, K - arg, E, 1.0)
            col += K - arg
    assert col == n
    cols = [rows] + ([negs] if hn else []) + [queue[:, hn:].t().contiguous()]
    want = _oracle_chains(a.cpu().numpy(), torch.cat(cols, 0).cpu().numpy())
    got = out.cpu().numpy()
    assert np.array_equal(got, want), float(np.abs(got - want).max())
    # backward: d a = sum over blocks of dsim[:, cols] @ block, first block overwrites, the others accumulate.  The queue block
    # (a reduction over 57 344 columns into 2 x 6 output tiles) runs as the deterministic split-K form in blip_model; both forms
    # are checked: the un-split k-ordered chain bit for bit, the split form against the chain restated per slice in slice order
    from uniir_amd import _lib
    lib = _lib.load()
    dsim = torch.randn(b, n, device=DEV, generator=g) * 1e-2
    dn = dsim.cpu().numpy()
    for split in (False, True):
        da = torch.full((b, E), float("nan"), device=DEV)
        col, first = 0, True
        for kind, t, arg in blocks:
            fn = "uniir_sgemm" if first else "uniir_sgemm_acc"
            if kind == "rows":
                ops.call(fn, dsim[:, col:], n, 1, t, E, 1, da, E, b, E, arg, 1.0)
                col += arg
            elif not split:
                ops.call(fn, dsim[:, col:], n, 1, t[:, arg:], 1, K, da, E, b, E, K - arg, 1.0)
                col += K - arg
            else:
                ws = torch.empty(int(lib.uniir_sgemm_splitk_workspace_bytes(b, E, K - arg)), device=DEV, dtype=torch.uint8)
                ops.call("uniir_sgemm_splitk", dsim[:, col:], n, 1, t[:, arg:], 1, K, da, E, b, E, K - arg, 1.0,
                         0 if first else 1, ws, ws.numel())
                col += K - arg
            first = False
        want_da, col = None, 0
        for bi, mat in enumerate(cols):              # chain over the block's columns, blocks added in fp32 in call order
            w = mat.shape[0]
            mt = mat.t().contiguous().cpu().numpy()
            if split and bi == len(cols) - 1:        # the library's slicing: ~1024 workgroups, whole 16-column K steps per slice
                tiles = -(-b // 128) * -(-E // 128)
                splits = max(1, min(-(-1024 // tiles), w // (16 * 8)))
                kslice = -(-(w // 16) // splits) * 16
                part = None
                for k0 in range(0, w, kslice):
                    sl = _oracle_chains(dn[:, col + k0:col + min(w, k0 + kslice)], mt[:, k0:min(w, k0 + kslice)])
                    part = sl if part is None else (part + sl).astype(np.float32)
            else:
                part = _oracle_chains(dn[:, col:col + w], mt)
            want_da = part if want_da is None else (want_da + part).astype(np.float32)
            col += w
        got_da = da.cpu().numpy()
        assert np.array_equal(got_da, want_da), (split, float(np.abs(got_da - want_da).max()))


def test_search_shard_multi_sweep_loop_equals_the_oracle(monkeypatch):
    """1 000 queries in sweeps of 64 (15 full sweeps + one of 40) over a 20 000-row shard: scores bit-exact, ids identical"""
    from oracle import c_oracle
    from uniir_amd import _lib, retrieval
    lib = _lib.load()
    assert lib.uniir_topk_set_chunk(64) == 0
    try:
        n, d, nq, k = 20_000, 768, 1_000, 10
        g = torch.Generator(device=DEV).manual_seed(123)
        pool = torch.randn(n, d, device=DEV, generator=g).half()
        pool[17] = 0                                                    # a zero row (inverse norm 0)
        pool[4000:4003] = pool[77]                                      # exact duplicates: ties broken by id
        queries = torch.randn(nq, d, device=DEV, generator=g).half()
        queries[5] = pool[77]
        ids = torch.randperm(n, device=DEV, generator=g).to(torch.int64) + 9_000_000
        s, i = retrieval.search_shard(retrieval.PoolShard(pool, ids), queries, k)
        ws, wi = c_oracle.topk(pool.cpu().numpy(), ids.cpu().numpy(), queries.cpu().numpy(), k)
        assert np.array_equal(i.cpu().numpy(), wi)
        assert np.array_equal(s.cpu().numpy(), ws)
    finally:
        assert lib.uniir_topk_set_chunk(0) == 0


def test_sweep_size_is_automatic_and_both_bulk_scans_equal_the_oracle():
    """uniir_topk_ip picks 256-query sweeps where the streaming scan applies (dim 768, >= 2048 groups) and 1024 elsewhere; the same
    700-query search run as 256 + 256 + 188 (streaming scan) and, forced, as one 700-query sweep (GEMM-shaped scan) gives the
    oracle's scores bit for bit and its ids either way"""
    from oracle import c_oracle
    from uniir_amd import _lib, retrieval
    lib = _lib.load()
    assert retrieval.sweep_queries(768, 700_000) == 256 and retrieval.sweep_queries(768, 40_030) == 256
    assert retrieval.sweep_queries(512, 700_000) == 1024 and retrieval.sweep_queries(768, 20_000) == 1024
    n, d, nq, k = 40_030, 768, 700, 10
    g = torch.Generator(device=DEV).manual_seed(321)
    pool = torch.randn(n, d, device=DEV, generator=g).half()
    pool[n - 1] = pool[5]
    queries = torch.randn(nq, d, device=DEV, generator=g).half()
    queries[300] = pool[5]
    ids = torch.randperm(n, device=DEV, generator=g).to(torch.int64) * 2 + 11
    ws, wi = c_oracle.topk(pool.cpu().numpy(), ids.cpu().numpy(), queries.cpu().numpy(), k)
    shard = retrieval.PoolShard(pool, ids)
    try:
        for chunk in (0, 1024):
            assert lib.uniir_topk_set_chunk(chunk) == 0
            assert retrieval.sweep_queries(768, n) == (256 if chunk == 0 else 1024)
            s, i = retrieval.search_shard(shard, queries, k)
            assert np.array_equal(i.cpu().numpy(), wi), chunk
            assert np.array_equal(s.cpu().numpy(), ws), chunk
    finally:
        assert lib.uniir_topk_set_chunk(0) == 0


def test_vit_l14_forward_only_modality_masks_against_the_oracle():
    """6 items through the no-grad towers of the headline model: 2 image-only, 2 text-only, 2 pairs (clip_sf.py:61-62: the
    masked tower's output is multiplied by 0, so a masked-out modality must contribute exactly nothing)"""
    from oracle import clip_oracle as O
    from models.uniir_clip.clip_scorefusion.clip_sf import CLIPScoreFusion
    from uniir_amd.clip_model import CLIP_CONFIGS
    cfg = CLIP_CONFIGS["ViT-L/14"]
    sd = O.init_state_dict(cfg, seed=4)
    config = SimpleNamespace(model=SimpleNamespace(gather_embeddings=False), data_config=SimpleNamespace(in_batch_neg_num=0))
    os.environ["UNIIR_ALLOW_RANDOM_INIT"] = "1"
    model = CLIPScoreFusion("ViT-L/14", device=DEV, config=config)
    model.clip_model.load_state_dict(sd, strict=True)
    model.eval()
    batch = O.synthetic_batch(cfg, 3, seed=91)                        # 6 items
    tmask = torch.tensor([0, 0, 1, 1, 1, 1])
    imask = torch.tensor([1, 1, 0, 0, 1, 1])
    with torch.no_grad():
        emb_o = O.encode_multimodal_input(sd, cfg, batch["txt_batched"], batch["image_batched"], tmask, imask)
        emb_d = model.encode_multimodal_input(batch["txt_batched"].to(DEV), batch["image_batched"].to(DEV), tmask.to(DEV),
                                              imask.to(DEV)).cpu()
        # the same items with garbage in the masked-out modality: bitwise the same embeddings
        txt2, img2 = batch["txt_batched"].clone(), batch["image_batched"].clone()
        txt2[:2] = txt2[4:6]
        img2[2:4] = img2[4:6] * 3.0
        emb_g = model.encode_multimodal_input(txt2.to(DEV), img2.to(DEV), tmask.to(DEV), imask.to(DEV)).cpu()
    for r in range(6):
        rel = ((emb_d[r] - emb_o[r]).norm() / emb_o[r].norm()).item()
        assert rel < 1.2e-2, (r, rel)                                  # bf16 towers vs the fp32 oracle (observed ~6e-3 on pairs)
    assert torch.equal(emb_g, emb_d)


def test_mask_compacted_towers_equal_the_dense_towers_forward_and_backward():
    """Round 5: CLIP_SF runs each tower on the rows whose modality mask is 1 only (clip_sf.encode_multimodal_input,
    compact_masked = True; the reference runs both towers on every item and multiplies by the masks, clip_sf.py:53-63).
    Embeddings and the loss are BITWISE those of the dense path (x * 0 + y == y); every parameter gradient equals the dense one up to
    the order of fp32 additions (the dense path adds exact zeros for the dead rows in between).  ViT-B/32, 16 items: image-only,
    text-only, pairs, and one row with both masks 0 (sized so that the dense and the compacted towers run the same kernel class: below
    256 rows uniir_gemm sums bias gradients from the ROUNDED bf16 result in a separate pass, which is a 3e-3 difference of its own); plus a batch in which NO row has an image (empty image tower launch), and the
    prefetcher's host-side mask / caption-length hints against the device -> host fallback."""
    from oracle import clip_oracle as O
    from models.uniir_clip.clip_scorefusion.clip_sf import CLIPScoreFusion
    from uniir_amd.clip_model import CLIP_CONFIGS
    cfg = CLIP_CONFIGS["ViT-B/32"]
    sd = O.init_state_dict(cfg, seed=5)
    config = SimpleNamespace(model=SimpleNamespace(gather_embeddings=False), data_config=SimpleNamespace(in_batch_neg_num=0))
    os.environ["UNIIR_ALLOW_RANDOM_INIT"] = "1"
    model = CLIPScoreFusion("ViT-B/32", device=DEV, config=config)
    model.clip_model.load_state_dict(sd, strict=True)
    model.train()
    b = O.synthetic_batch(cfg, 8, seed=17)                            # 16 items = 8 (query, candidate) pairs
    txt, img = b["txt_batched"].to(DEV), b["image_batched"].to(DEV)
    tmask = torch.tensor([1, 0, 1, 1, 0, 1, 1, 0, 1, 1, 1, 0, 1, 1, 1, 1])
    imask = torch.tensor([0, 1, 1, 0, 1, 1, 0, 0, 1, 1, 0, 1, 1, 0, 1, 1])   # item 7: neither modality (an all-zero embedding)

    def run(compact, hints):
        model.compact_masked = compact
        model.zero_grad()
        t, tm, im = txt.clone(), tmask.to(DEV), imask.to(DEV)
        if hints:                       # what host_utils.DevicePrefetcher attaches
            t._uniir_lens = (b["txt_batched"].argmax(dim=-1) + 1).to(torch.int32)
            t._uniir_lens_version = t._version
            tm._uniir_host, im._uniir_host = tmask.clone(), imask.clone()
            tm._uniir_host_version, im._uniir_host_version = tm._version, im._version
        batch = {"txt_batched": t, "image_batched": img, "txt_mask_batched": tm, "image_mask_batched": im,
                 "index_mapping": b["index_mapping"]}
        emb = model.encode_multimodal_input(t, img, tm, im)
        out = model(batch)
        out["loss"].backward()
        grads = {n: p.grad.detach().clone() for n, p in model.clip_model.named_parameters()}
        return emb.detach().clone(), out["loss"].detach().clone(), grads

    emb_d, loss_d, g_d = run(False, False)
    for hints in (False, True):
        emb_c, loss_c, g_c = run(True, hints)
        assert torch.equal(emb_c, emb_d) and torch.equal(loss_c, loss_d), hints
        assert bool((emb_c[7] == 0).all())
        worst = max(((float((g_c[n] - g_d[n]).abs().max()) / (float(g_d[n].abs().max()) + 1e-12)), n) for n in g_d)
        assert worst[0] <= 2e-5, (worst, hints)
    # no image anywhere: the image tower is launched on zero rows
    model.eval()
    with torch.no_grad():
        tm0, im0 = torch.ones(16, dtype=torch.int64, device=DEV), torch.zeros(16, dtype=torch.int64, device=DEV)
        model.compact_masked = False
        e_dense = model.encode_multimodal_input(txt, img, tm0, im0)
        model.compact_masked = True
        e_comp = model.encode_multimodal_input(txt, img, tm0, im0)
    assert torch.equal(e_dense, e_comp)
    # a stale caption-length hint (tokens edited in place after the prefetch) is ignored, not trusted
    from uniir_amd.clip_model import text_row_offsets
    t = txt.clone()
    t._uniir_lens = torch.full((16,), 77, dtype=torch.int32)
    t._uniir_lens_version = t._version
    t[0, 0] += 0                                                       # in-place write: bumps the version
    assert text_row_offsets(t)[1] == int((txt.argmax(dim=-1) + 1).sum())
    t2 = txt.clone()
    t2._uniir_lens = torch.zeros(16, dtype=torch.int32)               # out of range: ignored
    t2._uniir_lens_version = t2._version
    assert text_row_offsets(t2)[1] == int((txt.argmax(dim=-1) + 1).sum())
    # a stale host copy of a modality mask (device mask edited in place after the hand-over) is ignored as well: the rows the towers
    # run on are the rows FuseFn's DEVICE mask keeps
    tm = tmask.to(DEV)
    tm._uniir_host, tm._uniir_host_version = tmask.clone(), tm._version
    assert torch.equal(CLIPScoreFusion._live_rows(tm), torch.nonzero(tmask).flatten())
    tm[1] = 1                                                          # now live on the device; the host copy still says dead
    want = tmask.clone()
    want[1] = 1
    assert torch.equal(CLIPScoreFusion._live_rows(tm), torch.nonzero(want).flatten())
    tm2 = tmask.to(DEV)
    tm2._uniir_host = tmask.clone()                                    # a hint without a version is not trusted either
    lt, li = CLIPScoreFusion._live_rows(tm2, imask.to(DEV))            # both read back in one transfer
    assert torch.equal(lt, torch.nonzero(tmask).flatten()) and torch.equal(li, torch.nonzero(imask).flatten())
