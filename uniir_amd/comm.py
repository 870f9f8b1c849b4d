"""The exchange steps of the hot path over torch.distributed (backend "nccl" = RCCL over xGMI on MI355X, "gloo" in
the CPU tests).  Only collectives the path really has:
  * all_gather_rows / reduce_scatter_rows: forward and backward of clip_sf.py:102-103
    (torch.distributed.nn.all_gather of p_embeds; its autograd backward is a reduce-scatter SUM);
  * allreduce_sum_ / GradReducer: DDP's gradient averaging (clip_scorefusion/train.py:218) on the flat fp32 gradient
    buffer -- GradReducer is DDP's bucketed, backward-overlapped form: ranges of the buffer are reduced on the
    collective stream as soon as backward has produced them;
  * gather_topk: per-shard (score, id) lists to every rank for the k-way merge (FAISS shard=True semantics,
    mbeir_retriever.py:98-100).
"""
import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def all_reduce_min_float(x, device=None):
    """MIN of a python float over the ranks (a rank-consistent decision from per-rank measurements)"""
    if world() == 1:
        return float(x)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t.item())


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


def broadcast_(t, src=0):
    """in-place broadcast of rank `src`'s tensor (gloo + device memory, tests only: staged through the host)"""
    if world() > 1:
        if t.is_cuda and dist.get_backend() == "gloo":
            host = t.detach().cpu()
            dist.broadcast(host, src=src)
            t.copy_(host)
        else:
            dist.broadcast(t, src=src)
    return t


def sync_replicas(module, src=0):
    """What wrapping in DistributedDataParallel does at construction (clip_scorefusion/train.py:218, uniir_blip/train.py:217:
    `DDP(model, device_ids=[gpu])` broadcasts rank 0's parameters AND buffers -- BLIP's queues, idx_queue, new_ptr_queue,
    the momentum encoders -- before the first step): every replica leaves with rank `src`'s state, whatever its own seed or
    checkpoint gave it.  One broadcast per distinct underlying storage: parameters that are views of a flat fp32 master buffer
    (CLIP towers, BLIP online / momentum stores, the CLIP_FF T5 store) travel as ONE collective over that buffer.  Afterwards
    the parameters' version counters are bumped so that the bf16 shadows are re-derived on the next forward, and BLIP's cached
    host copy of the queue pointer is dropped.  Returns the number of collectives issued."""
    if world() == 1:
        return 0
    tensors = list(module.parameters()) + list(module.buffers())
    seen, n = set(), 0
    with torch.no_grad():
        for t in tensors:
            st = t.untyped_storage()
            if st.data_ptr() == 0 or st.data_ptr() in seen:
                continue
            seen.add(st.data_ptr())
            whole = torch.empty(0, dtype=t.dtype, device=t.device).set_(st)       # the whole storage as one 1-D tensor
            broadcast_(whole, src)
            n += 1
        params = [p for p in module.parameters() if p.is_floating_point()]
        if params:
            torch._foreach_mul_(params, 1.0)          # exact no-op on the values; bumps _version -> shadows refresh lazily
    for m in module.modules():
        if hasattr(m, "_ptr_host"):
            m._ptr_host = None
    return n


def replica_checksum(module):
    """(sum, sum of squares) over every parameter and buffer in float64: equal on all ranks iff the replicas hold the same state
    (bench.py prints it per rank after the timed region; the 2-rank tests assert equality)"""
    s = torch.zeros(2, dtype=torch.float64, device=next(module.parameters()).device)
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            if t.numel():
                d = t.detach().double()
                s[0] += d.sum()
                s[1] += (d * d).sum()
    return [float(x) for x in s.cpu()]


class GradReducer:
    """DDP's bucketed gradient all-reduce overlapped with backward (torch DistributedDataParallel as wrapped at
    clip_scorefusion/train.py:218), on the flat fp32 gradient buffer of this build.

    Backward calls ready(lo, hi) when the kernels that produce flat[lo:hi] have been enqueued on the compute stream;
    adjacent ranges are coalesced and, once `bucket_bytes` have piled up, all-reduced (SUM) with async_op=True: on RCCL
    the collective runs on the process group's own stream, ordered after the compute stream's position at the call, so it
    overlaps the rest of backward.  finish() reduces whatever has not been announced (the complement of the announced
    ranges) and makes the compute stream wait for every collective.  Every element is reduced exactly once, so the result
    is bit-identical with one all-reduce of the whole buffer whenever the backend's sum is order-independent per element
    (2 ranks: a+b; the 2-rank test checks it).  The 1/world mean stays folded into the AdamW kernel."""

    def __init__(self, flat, bucket_bytes=64 << 20):
        self.flat = flat
        self.bucket_elems = max(1, bucket_bytes // flat.element_size())
        self.pending, self.launched, self.works = [], [], []
        self.announced = []           # every range of this step, sorted: a repeat is refused in ready(), not discovered at finish()
        self.extra_done = set()       # data_ptr of further flat buffers reduced through reduce_extra() this step
        self.n_collectives = 0
        self._stream = None           # the compute stream the pending ranges were announced on (device buffers only)

    def reset(self):
        """forget the state of an armed backward whose step() never came (an exception, a skipped step): called by
        NativeAdamW.arm_overlap before every armed backward.  Collectives already in flight are waited for first."""
        for w in self.works:
            w.wait()
        self.pending, self.launched, self.works, self.announced, self.n_collectives = [], [], [], [], 0
        self.extra_done = set()
        self._stream = None

    def _sync_backend(self):
        # tests only (two gloo ranks sharing one GPU): gloo cannot reduce device memory -> staged through the host
        return self.flat.is_cuda and dist.get_backend() == "gloo"

    def ready(self, lo, hi):
        """flat[lo:hi] is final for this step.  Contract: every range is announced at most ONCE per armed backward -- one tower
        call per step and tower (a second encode of the same tower inside one armed backward would announce its blocks again
        after the first, partial, sums went out) -- checked here, at the call that breaks it."""
        if hi <= lo or world() == 1:
            return
        import bisect
        i = bisect.bisect_left(self.announced, (lo, hi))
        if (i > 0 and self.announced[i - 1][1] > lo) or (i < len(self.announced) and self.announced[i][0] < hi):
            raise RuntimeError(f"GradReducer: gradient range [{lo}, {hi}) announced twice in one armed backward (a tower ran twice "
                               "in this step, or the previous armed backward was never followed by step()); use arm_overlap(False) "
                               "for such steps")
        self.announced.insert(i, (lo, hi))
        if self.flat.is_cuda:
            # a collective is ordered after the CURRENT stream's position only.  The towers' backwards may run on two streams
            # (clip_model.CLIP.side_leg): ranges announced from another stream go out first, from that stream, so that no bucket mixes
            # producers
            cur = torch.cuda.current_stream(self.flat.device)
            if self._stream is not None and self._stream != cur and self.pending:
                with torch.cuda.stream(self._stream):
                    self.flush()
            self._stream = cur
        self.pending.append((lo, hi))
        if sum(h - l for l, h in self.pending) >= self.bucket_elems:
            self.flush()

    def reduce_extra(self, tensor):
        """a further flat gradient buffer that is final now (CLIP_FF's T5 store: its backward ends before the towers' begins):
        one asynchronous all-reduce on the collective stream, overlapped with the rest of backward like the buckets"""
        if world() == 1:
            return
        if tensor.data_ptr() in self.extra_done:      # a second contribution after the sum went out would diverge silently
            raise RuntimeError("GradReducer.reduce_extra: this buffer was already reduced in the armed backward (its module ran "
                               "twice in one step); use arm_overlap(False) for such steps")
        if self._sync_backend():
            allreduce_sum_(tensor)
        else:
            self.works.append(dist.all_reduce(tensor, op=dist.ReduceOp.SUM, async_op=True))
        self.extra_done.add(tensor.data_ptr())
        self.n_collectives += 1

    @staticmethod
    def _coalesce(ranges):
        out = []
        for lo, hi in sorted(ranges):
            if out and lo <= out[-1][1]:
                if lo < out[-1][1]:
                    raise RuntimeError("GradReducer: overlapping gradient ranges announced")
                out[-1] = (out[-1][0], hi)
            else:
                out.append((lo, hi))
        return out

    def flush(self):
        for lo, hi in self._coalesce(self.pending):
            view = self.flat[lo:hi]
            if self._sync_backend():
                allreduce_sum_(view)
            else:
                self.works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True))
            self.launched.append((lo, hi))
            self.n_collectives += 1
        self.pending = []

    def finish(self):
        """reduce the not-yet-announced remainder, wait for everything, re-arm for the next step.  Returns (collectives issued,
        data_ptrs of the extra buffers already reduced)"""
        extra = set(self.extra_done)
        if world() > 1:
            self.flush()
            done = self._coalesce(self.launched)
            cur, rest = 0, []
            for lo, hi in done:
                if lo > cur:
                    rest.append((cur, lo))
                cur = hi
            if cur < self.flat.numel():
                rest.append((cur, self.flat.numel()))
            self.pending = rest
            self.flush()
            for w in self.works:
                w.wait()          # RCCL: the current stream waits for the collective (no host block); gloo: host wait
        n = self.n_collectives
        self.pending, self.launched, self.works, self.announced, self.n_collectives = [], [], [], [], 0
        self.extra_done = set()
        self._stream = None
        return n, extra


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
