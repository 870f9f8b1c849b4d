"""Per-workgroup timeline of the ping-pong GEMM (experimental build with -DPP_TS, tools/build_exp.sh): main loop vs
epilogue time per 256x256 tile, and how the tiles of one CU follow each other.
    tools/build_exp.sh ts -DPP_TS && UNIIR_HIP_LIB=$PWD/uniir_amd/libuniir_exp_ts.so python tools/gemm_ts.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uniir_amd import _lib, ops  # noqa: E402

dev = "cuda"
lib = _lib.load()
for (M, N, K, what) in ((263168, 3072, 1024, "qkv fwd"), (263168, 1024, 1024, "out fwd"), (263168, 4096, 1024, "fc fwd"),
                        (263168, 1024, 4096, "proj fwd")):
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    b = torch.randn(N, device=dev)
    for _ in range(3):
        y = ops.linear_fwd(x, w, b)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    y = ops.linear_fwd(x, w, b)
    e.record()
    torch.cuda.synchronize()
    ntile = (M // 256) * (N // 256)
    ts = np.zeros(3 * ntile, dtype=np.uint64)
    lib.uniir_debug_read_ts.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert lib.uniir_debug_read_ts(ts.ctypes.data, 3 * ntile) == 0
    ts = ts.reshape(ntile, 3).astype(np.int64)
    # s_memtime counts shader-clock cycles on gfx950 (the counts do not change with the data, the wall time does)
    main, epi = (ts[:, 1] - ts[:, 0]).astype(np.float64), (ts[:, 2] - ts[:, 1]).astype(np.float64)
    us = a.elapsed_time(e) * 1e3
    per_cu = (main + epi).sum() / 256
    ksteps = K // 64
    ideal = 256 * 256 * 64 * 2 / (2.5e15 / 256 / 2.4e9)      # cycles per k-step at the dense bf16 peak (4069 flop/clk/CU)
    print(f"{what}: kernel {us:7.1f} us, {ntile} tiles = {ntile / 256:.2f}/CU | per tile: main loop {main.mean():8.0f} cycles "
          f"({main.mean() / ksteps:6.0f}/k-step, MFMA issue utilisation {ideal * ksteps / main.mean():.3f}), epilogue "
          f"{epi.mean():6.0f} cycles ({epi.mean() / (main.mean() + epi.mean()):.3f} of the tile, "
          f"{256 * 256 * 2 / epi.mean():.1f} B/clk/CU written) | busy cycles/CU {per_cu:9.0f} -> clock {per_cu / us / 1e3:.2f} GHz")
