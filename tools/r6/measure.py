#!/usr/bin/env python
"""Round-6 closing measurements, one box, printed as a table:
  (1) attention forward / backward at 1024 items x 16 heads for 256, 257, 272 and 288 tokens: what the 257 = 16 x 16 + 1 tiling
      costs against the 256-token grid (tiles are 16 query rows x 32 keys: 257 tokens run 17 x 9 of them, 256 run 16 x 8);
  (2) bulk search 100 000 (and 16 384) queries x 700 000 x 768 with 256 (streaming scan), 512 and 1024 (ping-pong GEMM-shaped scan)
      queries per sweep (uniir_topk_set_chunk; results never depend on it), and the dim-512 pool at 1024 queries.
    python tools/r6/measure.py > gpurun_out/r06_measure.txt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uniir_amd import _lib, ops, retrieval  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def attention():
    b, H = 1024, 16
    print("# attention, 1024 items x 16 heads x 64 (bf16), ms per launch; 'grid' = MFMA tiles the kernel runs per head (16 q x 32 k)")
    print("tokens  fwd_ms  bwd_ms  fwd_vs_256  bwd_vs_256  grid")
    base = None
    for T in (256, 257, 272, 288):
        qkv = torch.randn(b * T, 3 * H * 64, device=dev).bfloat16()
        out, lse = ops.attention_fwd(qkv, b, T, H, 0)
        tf = min(timeit(lambda: ops.attention_fwd(qkv, b, T, H, 0, out=out, lse=lse), iters=20) for _ in range(3))
        do = torch.randn_like(out)
        dqkv = torch.empty_like(qkv)
        tb = min(timeit(lambda: ops.attention_bwd(qkv, out, do, lse, b, T, H, 0, dqkv=dqkv), iters=20) for _ in range(3))
        base = base or (tf, tb)
        print(f"{T:6d}  {tf*1e3:6.3f}  {tb*1e3:6.3f}  {tf/base[0]:10.3f}  {tb/base[1]:10.3f}  {(T + 15) // 16} x {(T + 31) // 32}")
        del qkv, out, lse, do, dqkv


def bulk():
    lib = _lib.load()
    print("# bulk search, one 700 000-row shard, k = 10; chunk = queries per sweep (0 = the library's own choice)")
    print("dim   queries  chunk  ms        mfma_frac  hbm_frac")
    for dim, n in ((768, 700_000), (512, 700_000)):
        pool = torch.randn(n, dim, device=dev).half()
        shard = retrieval.PoolShard(pool, torch.arange(n, device=dev))
        for nq in ((16384, 100_000) if dim == 768 else (1024, 16384)):
            q = torch.randn(nq, dim, device=dev).half()
            ref = None
            for chunk in (0, 256, 512, 1024):
                assert lib.uniir_topk_set_chunk(chunk) == 0
                s, i = retrieval.search_shard(shard, q, 10)
                if ref is None:
                    ref = (s.clone(), i.clone())
                same = bool(torch.equal(s, ref[0]) and torch.equal(i, ref[1]))
                t = min(timeit(lambda: retrieval.search_shard(shard, q, 10), iters=3, warm=1) for _ in range(2))
                sweeps = -(-nq // (chunk or 256))
                print(f"{dim:4d}  {nq:7d}  {chunk:5d}  {t*1e3:8.3f}  {2.0*nq*n*dim/t/2.5e15:9.4f}  {sweeps*n*dim*2/t/8e12:8.4f}  same_result={same}")
            lib.uniir_topk_set_chunk(0)
            del q
        del pool, shard


if __name__ == "__main__":
    which = sys.argv[1:] or ["attention", "bulk"]
    if "attention" in which:
        attention()
    if "bulk" in which:
        bulk()
