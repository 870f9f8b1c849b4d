"""Two ranks on ONE MI355X (backend gloo, both processes on cuda:0): the whole distributed CLIP_SF train step on the device
path -- all-gather of p, targets offset by rank, reduce-scatter backward, flat-gradient all-reduce with the 1/W mean folded
into AdamW -- must reproduce the single-process step on the concatenated batch (DDP semantics: the mean over ranks of the
per-rank q->p cross-entropies equals the un-gathered loss over all 2b pairs).  The 8-GPU RCCL run itself belongs to the
driver; this pins the exchange logic on real device tensors."""
import os
import sys
from types import SimpleNamespace

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(gather):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "uniir_amd", "src"))
    from oracle import clip_oracle as O
    from models.uniir_clip.clip_scorefusion.clip_sf import CLIPScoreFusion
    from uniir_amd import clip_model
    from uniir_amd.trainer import NativeTrainer
    cfg = O.tiny_config()
    clip_model.CLIP_CONFIGS["tiny-dist"] = cfg
    config = SimpleNamespace(model=SimpleNamespace(gather_embeddings=gather), data_config=SimpleNamespace(in_batch_neg_num=0))
    model = CLIPScoreFusion("tiny-dist", device="cuda:0", config=config)
    model.clip_model.load_state_dict(O.init_state_dict(cfg, seed=9))
    return O, cfg, model, NativeTrainer(model, lr=1e-3, t_total=10)


def _rank_batch(O, cfg, rank, pairs):
    full = O.synthetic_batch(cfg, 2 * pairs, seed=31)          # the global batch: 2 * pairs pairs
    lo, hi = rank * 2 * pairs, (rank + 1) * 2 * pairs          # this rank's items (pairs are [query, candidate] couples)
    out = {k: (v[lo:hi].cuda() if isinstance(v, torch.Tensor) else v) for k, v in full.items()}
    out["index_mapping"] = {"query": [[2 * i] for i in range(pairs)], "pos_cand": [[2 * i + 1] for i in range(pairs)]}
    return out


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    O, cfg, model, tr = _build(gather=True)
    out = tr.train_step(_rank_batch(O, cfg, rank, 4))
    torch.cuda.synchronize()
    q.put((rank, float(out["loss"].detach()), model.clip_model.visual.proj.detach().cpu().numpy().copy(),
           float(model.clip_model.logit_scale.detach())))       # plain numpy / floats: no torch shared-memory handles
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_device_step_equals_single_process_on_the_global_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, 29541, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r: (l, w, s) for r, l, w, s in (q.get(timeout=300) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
    O, cfg, model, tr = _build(gather=False)
    full = O.synthetic_batch(cfg, 8, seed=31)
    dfull = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in full.items()}
    out = tr.train_step(dfull)
    ref_loss = float(out["loss"].detach())
    assert abs(0.5 * (res[0][0] + res[1][0]) - ref_loss) < 2e-3 * max(1.0, abs(ref_loss))
    import numpy as np
    w_ref = model.clip_model.visual.proj.detach().cpu().numpy()
    step = np.abs(w_ref - O.init_state_dict(cfg, seed=9)["visual.proj"].numpy()).max()
    assert np.array_equal(res[0][1], res[1][1])        # replicas stay bit-identical
    for r in range(2):        # ... and match the single-process update on the global batch
        assert np.abs(res[r][1] - w_ref).max() < 0.05 * step + 1e-7, (np.abs(res[r][1] - w_ref).max(), step)
        assert abs(res[r][2] - model.clip_model.logit_scale.item()) < 1e-4


def _accum_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    O, cfg, model, _ = _build(gather=True)
    from uniir_amd.trainer import NativeTrainer
    tr = NativeTrainer(model, lr=1e-3, t_total=10, accumulation_steps=2)
    armed = []
    ready_orig = tr.opt.__class__.arm_overlap

    def spy(self, last=True):
        armed.append(bool(last))
        return ready_orig(self, last)

    tr.opt.__class__.arm_overlap = spy
    try:
        for micro in range(2):                      # one accumulation window = one optimizer step
            full = O.synthetic_batch(cfg, 8, seed=41 + micro)
            lo, hi = rank * 8, (rank + 1) * 8
            b = {k: (v[lo:hi].cuda() if isinstance(v, torch.Tensor) else v) for k, v in full.items()}
            b["index_mapping"] = {"query": [[2 * i] for i in range(4)], "pos_cand": [[2 * i + 1] for i in range(4)]}
            tr.train_step(b)
    finally:
        tr.opt.__class__.arm_overlap = ready_orig
    torch.cuda.synchronize()
    q.put((rank, armed, tr.opt.opt_step, tr.opt.last_collectives, model.clip_model.visual.proj.detach().cpu().numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_accumulation_reduces_only_on_the_last_micro_batch():
    """DDP no_sync semantics of the overlapped all-reduce: with gradient_accumulation_steps = 2 (engine.py:30-46) blocks are
    handed to the reducer only during the window's last backward; the update equals the single-process one that accumulates the
    same two global micro-batches"""
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_accum_worker, args=(r, 2, 29543, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r: (a, n, c, w) for r, a, n, c, w in (q.get(timeout=300) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
    for r in range(2):
        assert res[r][0] == [False, True] and res[r][1] == 1 and res[r][2] >= 2      # armed once, one step, several collectives
    assert np.array_equal(res[0][3], res[1][3])
    O, cfg, model, _ = _build(gather=False)
    from uniir_amd.trainer import NativeTrainer
    tr = NativeTrainer(model, lr=1e-3, t_total=10, accumulation_steps=2)
    for micro in range(2):
        full = O.synthetic_batch(cfg, 8, seed=41 + micro)
        tr.train_step({k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in full.items()})
    w_ref = model.clip_model.visual.proj.detach().cpu().numpy()
    step = np.abs(w_ref - O.init_state_dict(cfg, seed=9)["visual.proj"].numpy()).max()
    assert np.abs(res[0][3] - w_ref).max() < 0.05 * step + 1e-7, (np.abs(res[0][3] - w_ref).max(), step)


def _resident_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import numpy as np
    from uniir_amd import comm, retrieval
    g = np.random.default_rng(5)
    pool = g.standard_normal((3001, 128)).astype(np.float16)
    ids = (g.permutation(3001) + 7_000_000).astype(np.int64)
    queries = g.standard_normal((37, 128)).astype(np.float16)
    lo, hi = comm.contiguous_shard(3001)                       # the sampler's slice of the pool for this rank
    shard = retrieval.PoolShard(torch.from_numpy(pool[lo:hi]).cuda(), torch.from_numpy(ids[lo:hi]))
    qlo, qhi = (0, 30) if rank == 0 else (30, 37)              # uneven query counts per rank
    s, i = retrieval.search_resident(shard, torch.from_numpy(queries[qlo:qhi]).cuda(), 10)
    torch.cuda.synchronize()
    q.put((rank, s.cpu().numpy().copy(), i.cpu().numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_resident_shards_give_the_single_pool_topk():
    """embedder <-> retriever fusion: every rank keeps its slice of the pool in HBM, queries are exchanged, the merged
    result equals the C oracle's exact top-k over the whole pool (scores bit-exact, ids identical)"""
    import numpy as np
    sys.path.insert(0, ROOT)
    from oracle import c_oracle
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_resident_worker, args=(r, 2, 29547, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r: (s, i) for r, s, i in (q.get(timeout=300) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
    g = np.random.default_rng(5)
    pool = g.standard_normal((3001, 128)).astype(np.float16)
    ids = (g.permutation(3001) + 7_000_000).astype(np.int64)
    queries = g.standard_normal((37, 128)).astype(np.float16)
    want_s, want_i = c_oracle.topk(pool, ids, queries, 10)
    got_s = np.concatenate([res[0][0], res[1][0]])
    got_i = np.concatenate([res[0][1], res[1][1]])
    assert got_s.shape == (37, 10)
    assert np.array_equal(got_i, want_i) and np.array_equal(got_s, want_s)


def _sync_device_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "uniir_amd", "src"))
    from oracle import clip_oracle as O
    from models.uniir_clip.clip_scorefusion.clip_sf import CLIPScoreFusion
    from uniir_amd import clip_model, comm
    from uniir_amd.trainer import NativeTrainer
    cfg = O.tiny_config()
    clip_model.CLIP_CONFIGS["tiny-sync"] = cfg
    config = SimpleNamespace(model=SimpleNamespace(gather_embeddings=True), data_config=SimpleNamespace(in_batch_neg_num=0))
    model = CLIPScoreFusion("tiny-sync", device="cuda:0", config=config)
    model.clip_model.load_state_dict(O.init_state_dict(cfg, seed=50 + rank))      # rank-dependent weights
    clip = model.clip_model
    clip._sync_shadow()                                                          # flat storage + bf16 shadow of the OWN weights
    stale16 = clip.w16("visual.proj").float().clone()
    n = comm.sync_replicas(model)
    tr = NativeTrainer(model, lr=1e-3, t_total=10)
    out = tr.train_step(_rank_batch(O, cfg, rank, 4))
    torch.cuda.synchronize()
    shadow_fresh = bool((clip.w16("visual.proj").float() - clip.visual.proj.detach()).abs().max() < 1e-2)
    q.put((rank, n, comm.replica_checksum(model), float((stale16 - clip.w16("visual.proj").float()).abs().max()), shadow_fresh,
           float(out["loss"].detach())))
    dist.barrier()
    dist.destroy_process_group()


def test_replica_sync_on_the_flat_device_storage_then_a_step_keeps_replicas_identical():
    """comm.sync_replicas on the device layout (one broadcast of the flat fp32 master buffer; the bf16 shadow is re-derived):
    two ranks loaded with DIFFERENT weights train one distributed step and end bit-identical"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sync_device_worker, args=(r, 2, 29547, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    (_, n0, c0, d0, f0, l0), (_, n1, c1, d1, f1, l1) = res
    assert c0 == c1, (c0, c1)                 # identical after sync + one step (checksum over every parameter and buffer)
    assert n0 == n1 and n0 <= 4               # the flat buffer travels as ONE collective (+ conv shadow-independent leftovers)
    assert d1 > 1e-3 and f0 and f1            # rank 1's shadow really changed and matches its (new) master weights
