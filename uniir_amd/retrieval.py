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
    out_s = torch.full((nq, k), float("-inf"), device=dev, dtype=torch.float32)
    out_i = torch.full((nq, k), -1, device=dev, dtype=torch.int64)
    if nq == 0 or shard.n == 0:
        return out_s, out_i
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
