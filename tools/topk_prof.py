import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uniir_amd import retrieval
dev = "cuda"
n = 700000
pool = torch.randn(n, 768, device=dev).half()
shard = retrieval.PoolShard(pool, torch.arange(n, device=dev))
for nq in (int(os.environ.get("NQ", "64")),):
    q = torch.randn(nq, 768, device=dev).half()
    for _ in range(int(os.environ.get("NSEARCH", "12"))):
        retrieval.search_shard(shard, q, 10)
torch.cuda.synchronize()
