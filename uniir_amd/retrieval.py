"""Device-side brute-force retrieval (the FAISS "IDMap,Flat" inner-product replacement).

Mirrors what UniIR's src/common/mbeir_retriever.py does with faiss (create_index :69-103: fp16 -> fp32,
normalize_L2, add_with_ids; search_index :188-232: normalize queries, exact top-k) but keeps the pool in its
stored fp16 form resident in HBM: a pool shard is (fp16 [n,d] rows, fp32 inverse norms, int64 ids).
"""
import torch

from . import ops

COARSE_MARGIN = 8  # over-fetch so that fp16-MFMA vs exact-fp32 rounding can never change the final top-k


class PoolShard:
    """One GPU's slice of the candidate pool, resident in HBM."""

    def __init__(self, embeddings_f16: torch.Tensor, ids: torch.Tensor):
        if embeddings_f16.dtype != torch.float16 or not embeddings_f16.is_cuda:
            raise RuntimeError("PoolShard needs a CUDA fp16 [n, d] tensor")
        self.emb = embeddings_f16.contiguous()
        self.ids = ids.to(device=self.emb.device, dtype=torch.int64).contiguous()
        self.n, self.dim = self.emb.shape
        if self.dim % 64:
            raise RuntimeError("embedding dim must be a multiple of 64")
        self.inv_norm = torch.empty(max(self.n, 1), device=self.emb.device, dtype=torch.float32)
        if self.n:
            ops.call("uniir_pool_inv_norms", self.emb, self.n, self.dim, self.inv_norm)


def query_inv_norms(queries_f16: torch.Tensor) -> torch.Tensor:
    inv = torch.empty(queries_f16.shape[0], device=queries_f16.device, dtype=torch.float32)
    ops.call("uniir_pool_inv_norms", queries_f16, queries_f16.shape[0], queries_f16.shape[1], inv)
    return inv


QUERY_CHUNK = 1024   # the most queries one sweep of the shard takes inside uniir_topk_ip


def sweep_queries(dim, rows):
    """queries per sweep uniir_topk_ip uses on a shard of this shape (256 where the streaming scan applies, else 1024)"""
    from . import _lib
    return int(_lib.load().uniir_topk_ip_sweep_queries(int(dim), int(rows)))

MAX_K_DIRECT = 64 - COARSE_MARGIN     # 56: the scan keeps k + 8 <= 64 groups per query


def _shard_view(shard, lo, hi):
    """rows [lo, hi) of a resident shard as a shard of its own (views, nothing is copied or recomputed)"""
    v = object.__new__(PoolShard)
    v.emb, v.ids, v.inv_norm = shard.emb[lo:hi], shard.ids[lo:hi], shard.inv_norm[lo:hi]
    v.n, v.dim = hi - lo, shard.dim
    return v


SUBSHARD_BYTES = 1 << 31     # the streaming scans and the fused tail address a shard's rows through 31-bit buffer offsets
FUSED_TAIL_ROWS = 1024 * 2 * 24 * 16     # csrc/topk_select.h: the register-resident selection of the fused tail holds 1024 x 2 x
                                         # TK_SELREG group maxima = 49 152 groups of 16 rows = 786 432 rows


def subshard_bounds(n, dim):
    """Row ranges of the LOGICAL sub-shards a resident shard is searched in: one range while n * dim * 2 < 2 GiB, else equal ranges
    (on 32-row boundaries: aligned inverse norms, an even number of whole scan groups) that each stay below it AND within the fused
    tail's 786 432 rows.  Round 5: the ranges used to be as large as the 2-GiB bound allows (1.12 M rows at dim 768), which put every
    sub-shard search on the unfused tail (separate selection + re-score launches, strided passes over the group maxima) -- the
    16-19 % per-row penalty of the 256-query sweeps on the 5.6 M pool (12.3 ms for 1024 queries against 8 x 1.29).  The 5.6 M x 768
    M-BEIR pool on ONE GPU (8.6 GB, mbeir_retriever.py:196-206 with a single visible device) is 8 ranges of 700 000 rows -- the shards
    of the 8-GPU layout."""
    if n * dim * 2 < SUBSHARD_BYTES:
        return [(0, n)]
    max_rows = min(((SUBSHARD_BYTES - 1) // (dim * 2)) // 32 * 32, FUSED_TAIL_ROWS)
    parts = -(-n // max_rows)
    per = -(-(-(-n // parts)) // 32) * 32
    return [(lo, min(lo + per, n)) for lo in range(0, n, per)]


# (device index, stream handle) -> workspace tensor, grown on demand: a search allocates nothing in steady state.  Keyed by the
# stream the search runs on: two searches on different streams of one device get different scratch (group maxima, candidates), and a
# grown buffer replaces one that only its own stream has used -- the caching allocator re-issues the old block in stream order.
# Scratch above WS_CACHE_MAX_BYTES (a 100 000-query bulk search asks for more) is a plain per-call allocation and goes back to the
# allocator with the call; release_workspaces() drops everything (long-lived processes that search once).
_WS_CACHE = {}
WS_CACHE_MAX_BYTES = 1 << 30


def _workspace(dev, need):
    if need > WS_CACHE_MAX_BYTES:
        return torch.empty(need, device=dev, dtype=torch.uint8)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(idx).cuda_stream)
    ws = _WS_CACHE.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, device=dev, dtype=torch.uint8)
        _WS_CACHE[key] = ws
    return ws


def release_workspaces():
    """drop the cached search scratch of every device / stream"""
    _WS_CACHE.clear()


def search_shard(shard: PoolShard, queries_f16: torch.Tensor, k: int, q_inv=None, workspace=None):
    """Exact top-k of `queries` against one shard -> (scores f32 [q,k] desc, ids int64 [q,k], -1 padded).
    One C call (uniir_topk_ip: query norms, one sweep of the rows per 256 / 1024 queries, fused select + exact re-score + sort);
    a shard of >= 2 GiB goes to uniir_topk_ip_multi, which searches it as the equal sub-shards of subshard_bounds and merges on
    (score desc, id asc) -- the search of the whole shard (ids are unique).
    k > 56 (FAISS Flat accepts up to 2048; Recall@100, larger hard-negative mining depths): assembled from row slices of the
    shard, see _search_large_k."""
    from . import _lib
    queries_f16 = queries_f16.contiguous()
    nq = queries_f16.shape[0]
    dev = queries_f16.device
    if nq == 0 or shard.n == 0:          # FAISS pads missing results with -inf / -1
        return (torch.full((nq, k), float("-inf"), device=dev, dtype=torch.float32),
                torch.full((nq, k), -1, device=dev, dtype=torch.int64))
    if k > MAX_K_DIRECT:
        return _search_large_k(shard, queries_f16, k)
    out_s = torch.empty(nq, k, device=dev, dtype=torch.float32)       # every slot is written by the final sort (padding included)
    out_i = torch.empty(nq, k, device=dev, dtype=torch.int64)
    # one C call either way: uniir_topk_ip_multi cuts a shard of >= 2 GiB into the logical sub-shards of subshard_bounds itself
    # (one scan per sub-shard, then ONE tail, ONE sort and ONE merge launch for all of them) and is uniir_topk_ip below that
    multi = shard.n * shard.dim * 2 >= SUBSHARD_BYTES
    need = (_lib.load().uniir_topk_ip_multi_workspace_bytes(nq, k, shard.n, shard.dim) if multi
            else _lib.load().uniir_topk_ip_workspace_bytes_ex(nq, k, shard.n, shard.dim))
    if workspace is None or workspace.numel() < need or workspace.data_ptr() % 256:
        workspace = _workspace(dev, need)
    ops.call("uniir_topk_ip_multi" if multi else "uniir_topk_ip", shard.emb, shard.inv_norm, shard.ids, shard.n, shard.dim,
             queries_f16, nq, k, out_s, out_i, workspace, workspace.numel())
    return out_s, out_i


def _search_large_k(shard, queries_f16, k):
    """top-k for k > 56: the shard is cut into P row slices, each searched for its own top-56 (exact), and the lists are
    merged.  The merge is the exact global top-k iff no slice can hide a better row, i.e. iff every slice's WORST returned
    entry ranks after the merged k-th entry (then everything the slice did not return ranks later still).  That is checked
    on the device; when it fails (more than 56 of the top-k in one slice) P doubles.  Slices of <= 56 rows return all their
    rows, so the loop ends at the latest when every slice is that small."""
    nq, dev = queries_f16.shape[0], queries_f16.device
    kin = MAX_K_DIRECT
    parts = max(2, -(-2 * k // kin))
    while True:
        parts = min(parts, shard.n)
        per = -(-(-(-shard.n // parts)) // 16) * 16          # slices start on 16-row boundaries (aligned inverse norms)
        bounds = [(lo, min(lo + per, shard.n)) for lo in range(0, shard.n, per)]
        res = [search_shard(_shard_view(shard, lo, hi), queries_f16, kin) for lo, hi in bounds]
        scores = torch.stack([r[0] for r in res]).contiguous()          # [P, nq, kin]
        ids = torch.stack([r[1] for r in res]).contiguous()
        out_s = torch.empty(nq, k, device=dev, dtype=torch.float32)
        out_i = torch.empty(nq, k, device=dev, dtype=torch.int64)
        ops.call("uniir_topk_merge_ex", scores, ids, len(bounds), nq, kin, k, out_s, out_i)
        big = torch.tensor([hi - lo > kin for lo, hi in bounds], device=dev)
        if not bool(big.any()):
            return out_s, out_i
        worst_s, worst_i = scores[:, :, kin - 1], ids[:, :, kin - 1]                      # [P, nq]
        kth_s, kth_i = out_s[:, k - 1].unsqueeze(0), out_i[:, k - 1].unsqueeze(0)
        # a slice is safe for a query when its worst entry is strictly worse than the merged k-th (score desc, id asc), or the
        # merged list is not even full (kth id -1: every returned entry of every slice is already in it)
        after = (worst_s < kth_s) | ((worst_s == kth_s) & (worst_i > kth_i)) | (worst_i < 0)
        unsafe = big.unsqueeze(1) & ~after & (kth_i >= 0)
        if not bool(unsafe.any()):
            return out_s, out_i
        parts *= 2


def merge_shards(scores, ids):
    """scores/ids: [nshard, q, k] -> merged [q, k] by (score desc, id asc)."""
    nshard, nq, k = scores.shape
    out_s = torch.empty(nq, k, device=scores.device, dtype=torch.float32)
    out_i = torch.empty(nq, k, device=scores.device, dtype=torch.int64)
    ops.call("uniir_topk_merge", scores.contiguous(), ids.contiguous(), nshard, nq, k, out_s, out_i)
    return out_s, out_i


# ------------------------------------------------------------------------------------------------------------
# Embedder <-> retriever fusion.  The reference gathers every rank's embeddings on rank 0,
# writes .npy, re-reads them into a FAISS index file and re-uploads that file for every search
# (src/common/mbeir_embedder.py:63-116, mbeir_retriever.py:69-117,197-206).  Here each rank keeps the fp16 rows it
# encoded (its ContiguousDistributedSampler slice) in HBM as its pool shard; queries travel, the pool never does.
# ------------------------------------------------------------------------------------------------------------
def embed_resident(model, data_loader, device):
    """the embedder loop of mbeir_embedder.generate_embeds_and_ids_for_dataset_with_gather without the gather:
    -> (fp16 [n, d] on `device`, hashed ids int64 [n] on `device`) of this rank's slice"""
    chunks, ids = [], []
    with torch.no_grad():
        for batch in data_loader:
            for key, v in batch.items():
                if isinstance(v, torch.Tensor):
                    batch[key] = v.to(device, non_blocking=True)
                elif hasattr(v, "to_device"):                              # deferred device image transform
                    batch[key] = v.to_device(torch.device(device))
                elif hasattr(v, "input_ids") and hasattr(v, "items"):     # BLIP: transformers BatchEncoding
                    for kk, vv in v.items():
                        v[kk] = vv.to(device)
            emb, batch_ids = model(batch, encode_mbeir_batch=True)
            chunks.append(emb.half())          # the reference stores fp16 too (use_fp16=True)
            ids.extend(int(i) for i in batch_ids)
    emb = torch.cat(chunks, dim=0) if chunks else torch.zeros(0, 64, dtype=torch.float16, device=device)
    return emb.contiguous(), torch.tensor(ids, dtype=torch.int64, device=device)


def search_resident(shard: PoolShard, queries_f16: torch.Tensor, k: int):
    """Exact global top-k of this rank's queries over the union of all ranks' resident shards:
    all-gather the queries (small) -> every rank searches its own shard -> all-gather the per-shard top-k ->
    k-way merge on (score desc, id asc) -> each rank keeps the rows of its own queries.  Identical to the single-shard
    search over the concatenated pool (ids are unique across shards)."""
    from . import comm
    all_q, sizes = comm.all_gather_varlen(queries_f16.contiguous())
    s, i = search_shard(shard, all_q, k)
    gs, gi = comm.gather_topk(s, i)
    ms, mi = (gs[0], gi[0]) if gs.shape[0] == 1 else merge_shards(gs, gi)
    lo = sum(sizes[: comm.rank()])
    return ms[lo:lo + sizes[comm.rank()]], mi[lo:lo + sizes[comm.rank()]]
