"""time of the coarse scan alone (uniir_topk_coarse: scan + group selection) for small query counts"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uniir_amd import _lib, ops, retrieval
dev = "cuda"
n, d = 700000, 768
pool = torch.randn(n, d, device=dev).half()
shard = retrieval.PoolShard(pool, torch.arange(n, device=dev))
for nq in (16, 64):
    q = torch.randn(nq, d, device=dev).half()
    kc = 18
    need = _lib.load().uniir_topk_workspace_bytes(nq, kc, n)
    ws = torch.empty(need, device=dev, dtype=torch.uint8)
    ncand = _lib.load().uniir_topk_ncand(nq, kc)
    cand = torch.empty(nq, ncand, device=dev, dtype=torch.int32); cs = torch.empty(nq, kc, device=dev)
    f = lambda: ops.call("uniir_topk_coarse", shard.emb, shard.inv_norm, n, d, q, nq, kc, cand, cs, ws, ws.numel())
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    print(f"nq={nq}: coarse {t*1e6:.1f} us  pool {n*d*2/t/1e9:.0f} GB/s")
