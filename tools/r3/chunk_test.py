"""bulk search (1024 / 16384 queries over the 700 k shard) with 1024-query sweeps (ping-pong GEMM scan) vs 256-query sweeps
(4-wave shared-ring streaming scan): whole-search ms, TF/s"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uniir_amd import _lib, retrieval  # noqa: E402

dev = "cuda"
n = 700000
torch.manual_seed(0)
pool = torch.randn(n, 768, device=dev).half()
shard = retrieval.PoolShard(pool, torch.arange(n, device=dev))
lib = _lib.load()
for rep in range(2):
    for chunk in (0, 256):
        assert lib.uniir_topk_set_chunk(chunk) == 0
        for nq in (1024, 16384):
            q = torch.randn(nq, 768, device=dev).half()
            ws = torch.empty(lib.uniir_topk_ip_workspace_bytes(nq, 10, n), device=dev, dtype=torch.uint8)
            for _ in range(2):
                retrieval.search_shard(shard, q, 10, workspace=ws)
            torch.cuda.synchronize()
            it = 10 if nq <= 1024 else 2
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(it):
                retrieval.search_shard(shard, q, 10, workspace=ws)
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / it * 1e-3
            print(f"chunk={chunk or 1024} nq={nq}: {t*1e3:.3f} ms  {2*nq*n*768/t/1e12:.1f} TF/s", flush=True)
lib.uniir_topk_set_chunk(0)
