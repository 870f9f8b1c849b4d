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


QUERY_CHUNK = 1024   # queries per sweep of the shard (uniir_topk_coarse's group-max path takes <= 1024)


def search_shard(shard: PoolShard, queries_f16: torch.Tensor, k: int, q_inv=None, workspace=None):
    """Exact top-k of `queries` against one shard -> (scores f32 [q,k] desc, ids int64 [q,k], -1 padded)."""
    from . import _lib
    queries_f16 = queries_f16.contiguous()
    nq = queries_f16.shape[0]
    dev = queries_f16.device
    if nq == 0 or shard.n == 0:          # FAISS pads missing results with -inf / -1
        return (torch.full((nq, k), float("-inf"), device=dev, dtype=torch.float32),
                torch.full((nq, k), -1, device=dev, dtype=torch.int64))
    out_s = torch.empty(nq, k, device=dev, dtype=torch.float32)       # every slot is written by the final sort (padding included)
    out_i = torch.empty(nq, k, device=dev, dtype=torch.int64)
    if nq > QUERY_CHUNK:      # many queries (the reference hands over all of them at once): sweep the shard per 1024-query
        ws = workspace        # chunk on the MFMA group-max path, reusing one workspace
        for lo in range(0, nq, QUERY_CHUNK):
            hi = min(nq, lo + QUERY_CHUNK)
            if ws is None:
                need = _lib.load().uniir_topk_workspace_bytes(QUERY_CHUNK, min(64, k + COARSE_MARGIN), shard.n)
                ws = torch.empty(need, device=dev, dtype=torch.uint8)
            s_, i_ = search_shard(shard, queries_f16[lo:hi], k, None if q_inv is None else q_inv[lo:hi], ws)
            out_s[lo:hi], out_i[lo:hi] = s_, i_
        return out_s, out_i
    if q_inv is None:
        q_inv = query_inv_norms(queries_f16)
    kc = min(64, k + COARSE_MARGIN)
    if k > 64 - COARSE_MARGIN:
        raise RuntimeError("k too large for the coarse stage (max 56)")
    need = _lib.load().uniir_topk_workspace_bytes(nq, kc, shard.n)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, device=dev, dtype=torch.uint8)
    ncand = _lib.load().uniir_topk_ncand(nq, kc)
    cand = torch.empty(nq, ncand, device=dev, dtype=torch.int32)
    cand_s = torch.empty(nq, kc, device=dev, dtype=torch.float32)
    ops.call("uniir_topk_coarse", shard.emb, shard.inv_norm, shard.n, shard.dim, queries_f16, nq, kc, cand, cand_s,
             workspace, workspace.numel())
    exact = torch.empty(nq, ncand, device=dev, dtype=torch.float32)
    ops.call("uniir_topk_rescore", shard.emb, shard.inv_norm, shard.ids, shard.n, shard.dim, queries_f16, q_inv, nq,
             cand, ncand, k, exact, out_s, out_i)
    return out_s, out_i


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
