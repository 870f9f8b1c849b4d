"""A few GEMM launches of the step's shapes for rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE per dispatch).
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python tools/gemm_pmc.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uniir_amd import ops

dev = "cuda"
R = int(os.environ.get("ROWS", 263168))      # 1024 items x 257 tokens
W = 1024
shapes = [("qkv", 3 * W, W), ("out", W, W), ("fc", 4 * W, W), ("proj", W, 4 * W)]
for name, N, K in shapes:
    x = torch.randn(R, K, device=dev).bfloat16()
    w = torch.randn(N, K, device=dev).bfloat16()
    y = torch.empty(R, N, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(R, N, device=dev).bfloat16()
    dx = torch.empty(R, K, device=dev, dtype=torch.bfloat16)
    dw = torch.zeros(N, K, device=dev)
    for _ in range(2):
        ops.linear_fwd(x, w, out=y)          # NT  [R,K] x [N,K]^T
        ops.linear_dgrad(dy, w, out=dx)      # NN  [R,N] x [N,K]
        ops.linear_wgrad(dy, x, dw)          # TN  [R,N]^T x [R,K]
    torch.cuda.synchronize()
    alg = {"NT": 2 * (R * K + N * K + R * N), "NN": 2 * (R * N + N * K + R * K), "TN": 2 * (R * N + R * K) + 4 * N * K}
    print(name, "algorithmic MB:", {k: round(v / 1e6, 1) for k, v in alg.items()})
