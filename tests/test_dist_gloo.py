"""world_size-2 gloo tests (CPU) of the N>1 exchange steps in uniir_amd/comm.py, the only collectives the path has:
all-gather of p (rank-major) with its reduce-scatter backward, the target offsets, the flat-gradient mean, the
top-k gather + merge inputs and the contiguous sharding.  The math between the collectives is the oracle's (the HIP
kernels cannot run here); the expected values are the golden G2 vectors captured from the REFERENCE running under 2
gloo ranks (tests/golden/g2_infonce_w2.npz)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


class _GatherFn(torch.autograd.Function):
    """same comm calls as uniir_amd.losses.InBatchNCEFn (all_gather_rows fwd / reduce_scatter_rows bwd)"""

    @staticmethod
    def forward(ctx, p):
        from uniir_amd import comm
        ctx.b = p.shape[0]
        return comm.all_gather_rows(p)

    @staticmethod
    def backward(ctx, d_all):
        from uniir_amd import comm
        return comm.reduce_scatter_rows(d_all.contiguous(), ctx.b)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import clip_oracle as O
    from uniir_amd import comm
    d = np.load(os.path.join(G, "g2_infonce_w2.npz"))
    txt = torch.tensor(d[f"r{rank}_txt"], requires_grad=True)
    img = torch.tensor(d[f"r{rank}_img"], requires_grad=True)
    emb = img + txt
    b = emb.shape[0] // 2
    im = {"query": [[2 * i] for i in range(b)], "pos_cand": [[2 * i + 1] for i in range(b)]}
    out = O.inbatch_contrastive_loss(emb, im, torch.tensor(float(np.exp(np.log(1 / 0.07)))), gather=_GatherFn.apply,
                                     rank=comm.rank())
    out["loss"].backward()
    res = {"loss": out["loss"].item(), "acc": out["accuracy"].item(), "score": out["score"].detach().numpy(),
           "dtxt": txt.grad.numpy(), "toff": comm.target_offset(b)}
    # flat gradient mean (DDP semantics) through allreduce_sum_ + 1/world
    flat = torch.full((5,), float(rank + 1))
    comm.allreduce_sum_(flat)
    res["flat"] = (flat / comm.world()).numpy()
    # top-k gather
    s, i = comm.gather_topk(torch.full((3, 2), float(rank)), torch.full((3, 2), rank, dtype=torch.int64))
    res["gs"], res["gi"] = s.numpy(), i.numpy()
    res["shard"] = comm.contiguous_shard(9)
    # uneven query counts per rank (3 rows on rank 0, 1 row on rank 1) -> rank-ordered concatenation everywhere
    rows, sizes = comm.all_gather_varlen(torch.full((3 - 2 * rank, 4), float(10 + rank), dtype=torch.float16))
    res["var"], res["sizes"] = rows.float().numpy(), sizes
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_exchange_matches_reference_golden():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, 29533, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    d = np.load(os.path.join(G, "g2_infonce_w2.npz"))
    for r in range(2):
        assert abs(res[r]["loss"] - float(d[f"r{r}_loss"])) < 1e-5
        assert res[r]["acc"] == float(d[f"r{r}_acc"])
        assert np.abs(res[r]["score"] - d[f"r{r}_score"]).max() < 1e-4
        assert np.abs(res[r]["dtxt"] - d[f"r{r}_dtxt"]).max() < 1e-5      # includes the reduce-scatter backward
        assert res[r]["toff"] == r * 6
        assert np.allclose(res[r]["flat"], 1.5)
        assert res[r]["gs"].shape == (2, 3, 2) and res[r]["gi"][1, 0, 0] == 1
    assert res[0]["shard"] == (0, 5) and res[1]["shard"] == (5, 9)
    for r in range(2):
        assert res[r]["sizes"] == [3, 1] and res[r]["var"].shape == (4, 4)
        assert np.array_equal(res[r]["var"][:, 0], np.array([10, 10, 10, 11], dtype=np.float32))


def _blip_queue_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uniir_amd.blip_model import BLIPFeatureFusion
    med = dict(hidden_size=128, intermediate_size=256, num_attention_heads=2, num_hidden_layers=1, vocab_size=64,
               max_position_embeddings=32)
    vit = dict(img_size=32, patch_size=16, embed_dim=128, depth=1, num_heads=2)
    m = BLIPFeatureFusion(med_config=med, vit_config=vit, embed_dim=128, queue_size=8, momentum=0.9)
    for step in range(3):          # 3 x (2 ranks x 2 rows) = 12 rows through a queue of 8: wraps once
        qf = torch.full((2, 128), float(10 * step + rank))
        cf = -qf
        m._dequeue_and_enqueue(qf, cf, torch.tensor([100 * step + 2 * rank, 100 * step + 2 * rank + 1]))
    q.put((rank, {"qq": m.query_queue[0].numpy().copy(), "cq": m.cand_queue[0].numpy().copy(),
                  "idx": m.idx_queue[0].numpy().copy(), "ptr": int(m.new_ptr_queue)}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_blip_queue_update_is_rank_major_and_identical_on_every_rank():
    """blip_ff.py:294-310 under 2 ranks: all-gathered (idx, q_m, c_m) are written rank-major at the pointer"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_blip_queue_worker, args=(r, 2, 29534, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    want_q = np.array([20, 20, 21, 21, 10, 10, 11, 11], dtype=np.float32)     # step 2 wrapped over step 0
    want_idx = np.array([200, 201, 202, 203, 100, 101, 102, 103])
    for r in range(2):
        assert np.array_equal(res[r]["qq"], want_q) and np.array_equal(res[r]["cq"], -want_q)
        assert np.array_equal(res[r]["idx"], want_idx) and res[r]["ptr"] == 4


def _reducer_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uniir_amd import comm
    g = torch.Generator().manual_seed(100 + rank)
    n = 10_000
    grads = torch.randn(n, generator=g) * torch.logspace(-6, 3, n)       # wide dynamic range: any re-association would show
    flat_ref = grads.clone()
    comm.allreduce_sum_(flat_ref)                                        # the round-1 path: one all-reduce of everything
    flat = grads.clone()
    red = comm.GradReducer(flat, bucket_bytes=4 * 1500)
    # "backward": blocks announced last-to-first like _tower_bwd does (adjacent ranges, to be coalesced), a second tower,
    # one stray range; [0, 1000) and [9000, 10000) are never announced and must be picked up by finish()
    for lo in range(4000, 1000, -500):
        red.ready(lo, lo + 500)
    for lo in range(8000, 4500, -700):
        red.ready(lo, lo + 700)
    red.ready(8700, 9000)
    launched_before_finish = red.n_collectives
    ncoll, _ = red.finish()
    # a second step on the same reducer (state re-armed)
    flat2 = flat.clone()
    red2_in = flat2.clone()
    comm.allreduce_sum_(red2_in)
    red.flat = flat2
    red.ready(0, 5000)
    # a second flat buffer handed over in the middle of backward (CLIP_FF's T5 store): reduced once, reported by finish()
    extra = torch.full((77,), float(rank + 1))
    red.reduce_extra(extra)
    twice = False
    try:
        red.reduce_extra(extra)        # a second contribution after the sum went out is refused, not ignored
    except RuntimeError:
        twice = True
    _, extra_done = red.finish()
    extra_ok = twice and bool((extra == 3.0).all()) and extra.data_ptr() in extra_done
    same2 = torch.equal(flat2, red2_in)
    # a range announced twice inside one armed backward is refused AT the second announcement ...
    red.ready(100, 200)
    dup = False
    try:
        red.ready(150, 250)
    except RuntimeError:
        dup = True
    # ... and reset() (NativeAdamW.arm_overlap) drops the state of a backward whose step() never came
    red.reset()
    red.ready(100, 200)
    red.finish()
    q.put((rank, torch.equal(flat, flat_ref), same2 and extra_ok and dup, launched_before_finish, ncoll))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_overlapped_gradient_allreduce_equals_flat_allreduce_bit_for_bit():
    """DDP-style buckets (comm.GradReducer, fed per finished block by clip_model._tower_bwd) == one all-reduce of the
    flat gradient buffer, bit for bit, including the never-announced remainder and re-use across steps"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_reducer_worker, args=(r, 2, 29536, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    for rank, same, same2, early, ncoll in res:
        assert same and same2
        assert early >= 2            # buckets really left before finish() (overlap), not one blocking call at the end
        assert ncoll > early


def _sync_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import clip_oracle as O
    from uniir_amd import clip_model, comm
    cfg = O.tiny_config()
    model = clip_model.CLIP(cfg, seed=100 + rank)            # rank-dependent initialisation: the replicas start DIFFERENT
    # buffers travel too (DDP broadcasts them: BLIP's queues / pointer); and a cached host copy of the pointer is dropped
    model.register_buffer("queue", torch.randn(4, 6, generator=torch.Generator().manual_seed(rank)))
    model.register_buffer("new_ptr_queue", torch.tensor([3 * rank + 1]))
    model._ptr_host = 3 * rank + 1
    before = comm.replica_checksum(model)
    versions = sum(p._version for p in model.parameters())
    n = comm.sync_replicas(model)
    after = comm.replica_checksum(model)
    bumped = sum(p._version for p in model.parameters()) > versions       # the bf16 shadows will be re-derived
    q.put((rank, before, after, n, int(model.new_ptr_queue), model._ptr_host, bumped,
           float(model.visual.proj.detach().double().sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_replica_sync_makes_differently_seeded_ranks_identical():
    """VERDICT r2 missing #3: what DDP's constructor does (clip_scorefusion/train.py:218): rank 0's parameters and buffers reach
    every replica.  Two ranks built from different seeds -> identical state (checksum + a spot value), rank 0 unchanged."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sync_worker, args=(r, 2, 29541, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    (r0, b0, a0, n0, ptr0, host0, bump0, spot0), (r1, b1, a1, n1, ptr1, host1, bump1, spot1) = res
    assert b0 != b1                       # they really started different
    assert a0 == a1 == b0                 # ... and both ended with rank 0's state, bit for bit (float64 sums of identical tensors)
    assert spot0 == spot1 and ptr0 == ptr1 == 1 and host0 is None and host1 is None and bump0 and bump1
    assert n0 == n1 and n0 >= 3


def test_reducer_rejects_overlapping_ranges():
    from uniir_amd import comm
    import pytest
    with pytest.raises(RuntimeError):
        comm.GradReducer._coalesce([(0, 10), (5, 20)])
    assert comm.GradReducer._coalesce([(10, 20), (0, 10), (30, 40)]) == [(0, 20), (30, 40)]


def _stash_review_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import clip_oracle as O
    from uniir_amd import clip_model, comm
    model = clip_model.CLIP(O.tiny_config(), seed=5)
    model._flat = {"dev": torch.device("cpu")}                      # "a training step has run" (review_stash never reads the buffers)
    model._measured_headroom = lambda: (float(40 << 30), float(200 << 30), float(2 << 30))
    # rank 0 chose the stash for both towers; rank 1 had less free memory and chose it for NEITHER (its old early return skipped
    # the two MIN all-reduces, so rank 0's reductions paired with rank 1's next gradient bucket)
    model._stash_choice = {"text": rank == 0, "image": rank == 0}
    calls = []
    def counted_min(x):
        calls.append(x)
        return comm.all_reduce_min_float(x)
    ret = model.review_stash(counted_min)
    # the collective that follows on every rank (a gradient bucket): must still pair up and sum correctly
    g = torch.full((1000,), float(rank + 1))
    dist.all_reduce(g)
    q.put((rank, len(calls), ret, dict(model._stash_choice), bool((g == 3.0).all()), list(model.stash_log)))
    dist.barrier()
    dist.destroy_process_group()


def test_stash_review_runs_the_same_collectives_on_a_rank_that_chose_no_stash():
    """clip_model.CLIP.review_stash returned before its two MIN all-reduces on a rank whose automatic choice was
    'off' for every tower -- a per-rank condition.  Both ranks must issue both reductions, end with the stash off everywhere, and the
    next all-reduce must pair up."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_stash_review_worker, args=(r, 2, 29547, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    (_, n0, ret0, ch0, ok0, log0), (_, n1, ret1, ch1, ok1, log1) = res
    assert n0 == 2 and n1 == 2                                      # the same collective sequence on both ranks
    assert ok0 and ok1                                              # ... so the gradient bucket after it summed 1 + 2 everywhere
    assert ch0 == {"text": False, "image": False} == ch1            # one mode on every rank: rank 0 followed rank 1
    assert ret0 == float(40 << 30) and "ranks agreed on the stash: False" in log0[-1]
    assert ret1 is None and "took part" in log1[-1]
