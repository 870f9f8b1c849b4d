"""a few launches of the 257-token attention kernels at 1024 items (profiling target)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniir_amd import ops
T, H, b = 257, 16, 1024
qkv = torch.randn(b * T, 3 * H * 64, device="cuda").bfloat16()
out, lse = ops.attention_fwd(qkv, b, T, H, 0)
do = torch.randn_like(out)
dqkv = torch.empty_like(qkv)
for _ in range(int(os.environ.get("N", 5))):
    ops.attention_fwd(qkv, b, T, H, 0, out=out, lse=lse)
    ops.attention_bwd(qkv, out, do, lse, b, T, H, 0, dqkv=dqkv)
torch.cuda.synchronize()
