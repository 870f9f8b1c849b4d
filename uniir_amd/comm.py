"""The exchange steps of the hot path over torch.distributed (backend "nccl" = RCCL over xGMI on MI355X, "gloo" in
the CPU tests).  Only collectives the path really has:
  * all_gather_rows / reduce_scatter_rows: forward and backward of clip_sf.py:102-103
    (torch.distributed.nn.all_gather of p_embeds; its autograd backward is a reduce-scatter SUM);
  * allreduce_mean_: DDP's gradient averaging (clip_scorefusion/train.py:218) on one flat buffer;
  * gather_topk: per-shard (score, id) lists to every rank for the k-way merge (FAISS shard=True semantics,
    mbeir_retriever.py:98-100).
"""
import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def all_gather_rows(p):
    """[b, E] on every rank -> [W*b, E], rank-major (== torch.cat(all_gather(p), dim=0))."""
    W = world()
    if W == 1:
        return p
    out = torch.empty(W * p.shape[0], p.shape[1], device=p.device, dtype=p.dtype)
    if p.is_cuda and dist.get_backend() == "gloo":      # tests only (two ranks on one GPU): gloo gathers on the host
        host = torch.empty(out.shape, dtype=p.dtype)
        dist.all_gather_into_tensor(host, p.detach().cpu().contiguous())
        return out.copy_(host)
    dist.all_gather_into_tensor(out, p.contiguous())
    return out


def reduce_scatter_rows(d_all, b):
    """backward of all_gather_rows: rank r receives sum over ranks of d_all[r*b:(r+1)*b]."""
    W = world()
    if W == 1:
        return d_all
    out = torch.empty(b, d_all.shape[1], device=d_all.device, dtype=d_all.dtype)
    if dist.get_backend() == "gloo":   # gloo has no reduce_scatter: all_reduce (on the host) then slice (tests only)
        tmp = d_all.detach().cpu().clone()
        dist.all_reduce(tmp, op=dist.ReduceOp.SUM)
        out.copy_(tmp[rank() * b:(rank() + 1) * b])
    else:
        dist.reduce_scatter_tensor(out, d_all.contiguous(), op=dist.ReduceOp.SUM)
    return out


def target_offset(b):
    """clip_sf.py:135-136: sim_targets = rank * bs + arange(bs)."""
    return rank() * b


def allreduce_sum_(flat):
    if world() > 1:
        if flat.is_cuda and dist.get_backend() == "gloo":   # tests only (two ranks on one GPU)
            host = flat.detach().cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            flat.copy_(host)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def gather_topk(scores, ids):
    """[q, k] per rank -> [W, q, k] on every rank."""
    W = world()
    if W == 1:
        return scores.unsqueeze(0), ids.unsqueeze(0)
    q, k = scores.shape
    dev = scores.device
    host = scores.is_cuda and dist.get_backend() == "gloo"      # tests only (ranks sharing one GPU)
    s = torch.empty(W * q, k, device="cpu" if host else dev, dtype=scores.dtype)   # concatenation form (works on gloo too)
    i = torch.empty(W * q, k, device="cpu" if host else dev, dtype=ids.dtype)
    dist.all_gather_into_tensor(s, scores.contiguous().to(s.device))
    dist.all_gather_into_tensor(i, ids.contiguous().to(i.device))
    return s.view(W, q, k).to(dev), i.view(W, q, k).to(dev)


def all_gather_varlen(rows):
    """[n_r, ...] per rank (n_r may differ) -> (concatenation over ranks in rank order, list of n_r).  Used to hand every
    rank's queries to every pool shard; two small collectives (sizes, padded rows)."""
    W = world()
    if W == 1:
        return rows, [rows.shape[0]]
    gloo = dist.get_backend() == "gloo"          # tests only (ranks sharing one GPU): stage through the host
    dev = rows.device
    sizes = torch.zeros(W, dtype=torch.int64, device="cpu" if gloo else dev)
    mine = torch.tensor([rows.shape[0]], dtype=torch.int64, device=sizes.device)
    dist.all_gather_into_tensor(sizes, mine)
    sizes = [int(x) for x in sizes.tolist()]
    cap = max(max(sizes), 1)
    pad = torch.zeros((cap,) + tuple(rows.shape[1:]), dtype=rows.dtype, device="cpu" if gloo else dev)
    pad[: rows.shape[0]] = rows.detach().to(pad.device)
    out = torch.empty((W * cap,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=pad.device)
    dist.all_gather_into_tensor(out, pad)
    out = out.view((W, cap) + tuple(rows.shape[1:]))
    return torch.cat([out[r, : sizes[r]] for r in range(W)], dim=0).to(dev), sizes


def contiguous_shard(n, W=None, r=None):
    """common/dist_utils.py:94-115 ContiguousDistributedSampler: rank r owns [r*ceil(n/W), min((r+1)*ceil(n/W), n))."""
    W = world() if W is None else W
    r = rank() if r is None else r
    per = -(-n // W) if W > 0 else n
    lo = min(r * per, n)
    return lo, min(lo + per, n)
