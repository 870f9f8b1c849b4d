import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uniir_amd import ops
from tools.microbench import timeit
dev = "cuda"
M, N = int(os.environ.get("M", 65536)), int(os.environ.get("N", 1024))
for K in [int(k) for k in os.environ.get("KS", "64,128,256,512,1024,2048,4096,8192").split(",")]:
    x = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16()
    if os.environ.get('ZERO'): x.zero_(); w.zero_()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: ops.linear_fwd(x, w, out=y), iters=20)
    tiles = (M // 256) * (N // 256)
    print(f"K={K:5d}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s  per-tile-round {t*1e6/(tiles/256):7.2f} us")
