"""shader clock / board power (amdgpu hwmon, bench.BoardSampler) while ONE kernel family runs back to back for ~2 s each:
the step's GEMM shapes with random and with zero operands, LayerNorm, the 64-query scan.  Settles what clock the matrix pipe gets."""
import os
import sys
import time

import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import bench  # noqa: E402
from uniir_amd import ops  # noqa: E402

dev = "cuda"
M = 263168


def run(name, fn, flop, secs=2.0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    b = bench.BoardSampler(0)
    b.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        n += 10
    dt = time.perf_counter() - t0
    rec = b.stop()
    tf = flop * n / dt / 1e12
    print(f"{name}: {tf:.1f} TF/s  board {rec and (rec['sclk_mhz_mean'], rec['sclk_mhz_min'], rec['sclk_mhz_max'], rec['power_w_mean'])}"
          f"  -> per-clock efficiency {tf / (2500 * rec['sclk_mhz_mean'] / 2400):.3f}" if rec else f"{name}: {tf:.1f} TF/s (no hwmon)", flush=True)


for (N, K, tag) in [(3072, 1024, "qkv"), (1024, 1024, "out"), (4096, 1024, "fc"), (1024, 4096, "proj")]:
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.zeros(N, device=dev)
    run(f"gemm fwd {tag} random", lambda: ops.linear_fwd(x, w, bias), 2.0 * M * N * K)
    x.zero_()
    w.zero_()
    run(f"gemm fwd {tag} zeros ", lambda: ops.linear_fwd(x, w, bias), 2.0 * M * N * K)
