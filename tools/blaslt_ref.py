"""Calibration only (not part of the product): what the vendor library (hipBLASLt / rocBLAS behind torch.matmul) reaches
on the step's GEMM shapes with the same random bf16 data, next to libuniir_hip's ping-pong kernel -- to separate
"kernel quality" from "power-limited clock" in the roofline fraction."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uniir_amd import ops  # noqa: E402

dev = "cuda"


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3


for (M, N, K, what) in ((263168, 3072, 1024, "qkv fwd"), (263168, 1024, 1024, "out fwd"), (263168, 4096, 1024, "fc fwd"),
                        (263168, 1024, 4096, "proj fwd")):
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    t_lib = timeit(lambda: torch.matmul(x, w.t(), out=y))
    t_own = timeit(lambda: ops.linear_fwd(x, w, None, out=y))
    # wgrad form: dW[N,K] = dy^T [N,M] x [M,K]
    dy = torch.randn(M, N, device=dev).bfloat16()
    dw = torch.empty(N, K, device=dev, dtype=torch.float32)
    t_libw = timeit(lambda: torch.matmul(dy.t(), x))
    t_ownw = timeit(lambda: ops.linear_wgrad(dy, x, dw))
    print(f"{what:9s} {M}x{N}x{K}: vendor {fl / t_lib / 1e12:7.1f} TF/s  own {fl / t_own / 1e12:7.1f} TF/s | wgrad vendor "
          f"{fl / t_libw / 1e12:7.1f} (bf16 out)  own {fl / t_ownw / 1e12:7.1f} (fp32 accumulate into dW)")
