#!/usr/bin/env python
"""LayerNorm forward / backward micro-benchmark at the ViT-L/14 stream shape of the headline step (1024 items x 257
tokens x 1024) and the text shape (1024 x 77 x 768): time per launch and the algorithmic HBM rate.
    python tools/ln_bench.py [--items 1024]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uniir_amd import ops  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", type=int, default=1024)
    args = ap.parse_args()
    dev = "cuda"
    for T, W in ((257, 1024), (77, 768)):
        R = args.items * T
        x = torch.randn(R, W, device=dev)
        dy = torch.randn(R, W, device=dev).to(torch.bfloat16)
        dres = torch.randn(R, W, device=dev)
        gam, bet = torch.randn(W, device=dev), torch.randn(W, device=dev)
        dg, db, dc = torch.zeros(W, device=dev), torch.zeros(W, device=dev), torch.zeros(W, device=dev)
        dx, dxb = torch.empty(R, W, device=dev), torch.empty(R, W, device=dev, dtype=torch.bfloat16)
        y16 = torch.empty(R, W, device=dev, dtype=torch.bfloat16)
        t_f = timeit(lambda: ops.layernorm_fwd(x, gam, bet, 1e-5, out_bf16=y16, rows=R, width=W))
        t_b = timeit(lambda: ops.layernorm_bwd(x, gam, dy, dg, db, 1e-5, dres=dres, dx=dx, dx_bf16=dxb, rows=R, width=W,
                                               dx_colsum=dc))
        bf, bb = R * W * 6, R * W * 16
        print(f"T={T} W={W} rows={R}: fwd {t_f:8.1f} us {bf / t_f / 1e6:6.2f} TB/s | bwd {t_b:8.1f} us {bb / t_b / 1e6:6.2f} TB/s")
        del x, dy, dres, dx, dxb, y16
    # fused attention at the two tower shapes (algorithmic flops: 2 matmuls forward, 5 backward, 2 T^2 64 each)
    for T, heads, causal in ((257, 16, False), (77, 12, True)):
        M = args.items
        qkv = torch.randn(M * T, 3 * heads * 64, device=dev).to(torch.bfloat16)
        out, lse = ops.attention_fwd(qkv, M, T, heads, causal)
        dout = torch.randn(M * T, heads * 64, device=dev).to(torch.bfloat16)
        t_f = timeit(lambda: ops.attention_fwd(qkv, M, T, heads, causal, out=out, lse=lse))
        t_b = timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, M, T, heads, causal))
        fl = 2.0 * T * T * 64 * M * heads * (0.5 if causal else 1.0)
        print(f"attention T={T} heads={heads} causal={causal}: fwd {t_f:8.1f} us {2 * fl / t_f / 1e6:6.1f} TF/s | "
              f"bwd {t_b:8.1f} us {5 * fl / t_b / 1e6:6.1f} TF/s")


if __name__ == "__main__":
    main()
