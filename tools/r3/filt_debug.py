"""what the filtered scan left in the workspace after one 64-query search: ticket, thresholds, list sizes (debug aid)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uniir_amd import _lib, retrieval  # noqa: E402

dev = "cuda"
n, nq = 700000, 64
torch.manual_seed(0)
pool = torch.randn(n, 768, device=dev).half()
shard = retrieval.PoolShard(pool, torch.arange(n, device=dev))
q = torch.randn(nq, 768, device=dev).half()
lib = _lib.load()
ws = torch.zeros(lib.uniir_topk_ip_workspace_bytes(nq, 10, n), device=dev, dtype=torch.uint8)
retrieval.search_shard(shard, q, 10, workspace=ws)
torch.cuda.synchronize()
ngr = (n + 15) // 16
CTRL = 16384
LIST = 64 * 1024 * (ngr // 1024 + 4) * 4
raw = ws.cpu().numpy()
ctrl = raw[:CTRL].view(np.uint32)
print("tickets", ctrl[:8].tolist())
tauk = ctrl[16:80]


def unkey(k):
    k = np.uint32(k)
    if k == 0:
        return float("-inf")
    b = (k & np.uint32(0x7fffffff)) if (k & np.uint32(0x80000000)) else ~k
    return float(np.array([b], dtype=np.uint32).view(np.float32)[0])


print("tau[:8]", [round(unkey(k), 4) for k in tauk[:8]])
off = CTRL + 2 * LIST
cnt = raw[off:off + 64 * 1024 * 4].view(np.int32).reshape(64, 1024)
print("list entries per query: mean %.1f max %d; per (query, wave) max %d; total %d of %d groups x queries" %
      (cnt.sum(1).mean(), cnt.sum(1).max(), cnt.max(), cnt.sum(), ngr * 64))
