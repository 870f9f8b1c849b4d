"""one shard search (700 000 x 768 fp16, k = 10) at several query counts: whole-search time through retrieval.search_shard with a
pre-allocated workspace (HIP events over 30 searches after 5 warm-ups) -> ms, pool TB/s, fraction of the 8 TB/s HBM peak, TF/s"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uniir_amd import _lib, retrieval  # noqa: E402

dev = "cuda"
n = int(os.environ.get("MB_POOL", "700000"))
torch.manual_seed(0)
pool = torch.randn(n, 768, device=dev).half()
shard = retrieval.PoolShard(pool, torch.arange(n, device=dev))
lib = _lib.load()
for nq in [int(x) for x in os.environ.get("NQS", "16,64,128,256,1024").split(",")]:
    q = torch.randn(nq, 768, device=dev).half()
    if os.environ.get("ZERO_TAIL"):          # experiment: only the first 16 queries carry data (MFMA operand power vs store count)
        q[16:] = 0
    ws = torch.empty(lib.uniir_topk_ip_workspace_bytes(nq, 10, n), device=dev, dtype=torch.uint8)
    for _ in range(5):
        retrieval.search_shard(shard, q, 10, workspace=ws)
    torch.cuda.synchronize()
    it = 30 if nq <= 256 else 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        retrieval.search_shard(shard, q, 10, workspace=ws)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / it * 1e-3
    print(f"topk nq={nq} n={n}: {t*1e3:.4f} ms  pool {n*768*2/t/1e12:.3f} TB/s = {n*768*2/t/8e12:.3f} of 8 TB/s  "
          f"{2*nq*n*768/t/1e12:.1f} TF/s", flush=True)
