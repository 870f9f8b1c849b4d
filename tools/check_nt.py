"""quick NT-only correctness check of the GEMM epilogues (for tools/build_exp.sh libraries)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uniir_amd import ops
dev = "cuda"
def rel(a, b): return ((a.float() - b.float()).norm() / b.float().norm()).item()
torch.manual_seed(0)
for (M, N, K) in [(256, 256, 192), (1000, 520, 320), (4100, 1024, 1024), (66000, 1024, 256)]:
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.1).bfloat16()
    bias = torch.randn(N, device=dev)
    ref = x.float() @ w.float().t() + bias
    y = ops.linear_fwd(x, w, bias)
    g = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    f = ops.linear_fwd(x, w, bias, epilogue=ops.EPI_BIAS_ACT, C2=g)
    res = torch.randn(M, N, device=dev)
    c2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    r = ops.linear_fwd(x, w, bias, epilogue=ops.EPI_RESID_F32, resid=res, C2=c2)
    y32 = torch.empty(M, N, device=dev); ops.gemm(x, w, y32, M, N, K, K, K, N, epilogue=ops.EPI_F32)
    acc = torch.ones(M, N, device=dev); ops.gemm(x, w, acc, M, N, K, K, K, N, epilogue=ops.EPI_ATOMIC_F32, k_splits=max(1, K // 192))
    print(M, N, K, "bf16 %.1e act %.1e/%.1e resid %.1e/%.1e f32 %.1e splitk %.1e" % (
        rel(y, ref), rel(f, ref), rel(g, f.float() * torch.sigmoid(1.702 * f.float())), rel(r, ref + res), rel(c2, ref + res),
        rel(y32, ref - bias), rel(acc, ref - bias + 1)))
