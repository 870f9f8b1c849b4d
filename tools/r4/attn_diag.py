"""Round-4 diagnostic of the attention kernels (GPU box; needs the -DUNIIR_EXP_BUILD library, UNIIR_HIP_LIB=...):
knock-out timings (what each part of the kernel costs under real contention) and an s_memtime timeline of 64 mid-grid
workgroups.  Not part of the shipped path.

  UNIIR_HIP_LIB=uniir_amd/libuniir_exp_attdiag.so python tools/r4/attn_diag.py
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from uniir_amd import _lib, ops

dev = "cuda"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    lib = _lib.load()
    raw = C.CDLL(_lib.LIB_PATH)
    setf = raw.uniir_exp_attn_set
    setf.argtypes = [C.c_void_p, C.c_int]
    setf.restype = None
    T, H, b = int(os.environ.get("T", 257)), int(os.environ.get("H", 16)), int(os.environ.get("B", 1024))
    causal = int(os.environ.get("CAUSAL", 0))
    torch.manual_seed(0)
    qkv = torch.randn(b * T, 3 * H * 64, device=dev).bfloat16()
    out, lse = ops.attention_fwd(qkv, b, T, H, causal)
    do = torch.randn_like(out)
    dqkv = torch.empty_like(qkv)
    fwd = lambda: ops.attention_fwd(qkv, b, T, H, causal, out=out, lse=lse)
    bwd = lambda: ops.attention_bwd(qkv, out, do, lse, b, T, H, causal, dqkv=dqkv)
    print(f"T={T} H={H} items={b} causal={causal}")
    names = {0: "full", 1: "no phase-1/compute", 2: "no phase-2", 3: "no compute at all", 4: "no stage A", 8: "no stage B",
             12: "no staging", 16: "no stores", 15: "empty (launch + prologue)", 31: "empty, no stores",
             28: "compute only (no staging, no stores)"}
    for mode in (0, 1, 2, 3, 4, 8, 12, 16, 28, 15, 0):
        setf(None, mode)
        tf = timeit(fwd)
        tb = timeit(bwd)
        print(f"  exp={mode:2d} {names.get(mode, ''):40s} fwd {tf:.3f} ms   bwd {tb:.3f} ms")
    # timelines
    stamps = torch.zeros(64 * 8 * 8, dtype=torch.int64, device=dev)
    for (nm, fn, ns) in (("bwd", bwd, 8), ("fwd", fwd, 4)):
        setf(stamps.data_ptr(), 0)
        for _ in range(3):
            fn()
        stamps.zero_()
        fn()
        torch.cuda.synchronize()
        s = stamps.cpu().numpy().reshape(64, 8, 8).astype(np.int64)
        setf(None, 0)
        t0 = s[:, :, 0].min(axis=1, keepdims=True)           # workgroup start
        rel = s[:, :, :ns] - t0[:, :, None]
        print(f"{nm}: s_memtime ticks relative to the workgroup's first stamp; median over 64 mid-grid workgroups, per wave")
        med = np.median(rel, axis=0)
        for w in range(8):
            print(f"  wave {w}: " + " ".join(f"{int(x):8d}" for x in med[w]))
        life = (s[:, :, :ns].max(axis=(1, 2)) - s[:, :, 0].min(axis=1))
        print(f"  workgroup lifetime: median {int(np.median(life))} min {int(life.min())} max {int(life.max())} ticks")
        # how many workgroups overlap in time on average: span of the 64 workgroups vs sum of lifetimes
        span = s[:, :, :ns].max() - s[:, :, 0].min()
        print(f"  64 workgroups span {int(span)} ticks; sum of lifetimes {int(life.sum())}")
    # which clock is s_memtime? compare with a known-duration kernel
    setf(None, 0)


if __name__ == "__main__":
    main()
