import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uniir_amd import retrieval
dev = "cuda"
n = int(os.environ.get("ROWS", "700000"))        # ROWS=5600000: the whole M-BEIR pool as one resident shard (uniir_topk_ip_multi)
pool = torch.empty(n, 768, device=dev, dtype=torch.float16)
for lo in range(0, n, 700000):
    pool[lo:lo + 700000] = torch.randn(min(700000, n - lo), 768, device=dev).half()
shard = retrieval.PoolShard(pool, torch.arange(n, device=dev))
for nq in (int(os.environ.get("NQ", "64")),):
    q = torch.randn(nq, 768, device=dev).half()
    for _ in range(int(os.environ.get("NSEARCH", "12"))):
        retrieval.search_shard(shard, q, 10)
torch.cuda.synchronize()
