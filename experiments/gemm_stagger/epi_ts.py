"""Round-5 experiment: wall-clock stamps per workgroup (tools/build_exp.sh ts2 -DPP_TS=2) of the out-proj forward with the fp32
residual epilogue: how long is a tile's epilogue when 256 / 64 / 16 CUs run, and with the start stagger on.
    UNIIR_HIP_LIB=$PWD/uniir_amd/libuniir_exp_ts2.so python tools/r5/epi_ts.py"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uniir_amd import _lib, ops

dev = "cuda"
lib = _lib.load()
lib.uniir_debug_read_ts.argtypes = [ctypes.c_void_p, ctypes.c_int]


def tune(phases, ns_epi=6000):
    for k, v in ((0, phases), (1, 1450), (2, ns_epi), (3, 1)):
        assert lib.uniir_gemm_tune(k, v) == 0


def run(M, N, K, what, epi, phases=0):
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    b = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev)
    out = torch.empty(M, N, device=dev) if epi == "resid" else torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    tune(phases)
    fn = (lambda: ops.linear_fwd(x, w, b, out=out, epilogue=ops.EPI_RESID_F32, resid=res)) if epi == "resid" else (lambda: ops.linear_fwd(x, w, b, out=out))
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); e.record()
    torch.cuda.synchronize()
    ntile = ((M + 255) // 256) * (N // 256)
    ts = np.zeros(3 * ntile, dtype=np.uint64)
    assert lib.uniir_debug_read_ts(ts.ctypes.data, 3 * ntile) == 0
    ts = ts.reshape(ntile, 3).astype(np.int64)
    main, epi_t = (ts[:, 1] - ts[:, 0]) * 0.01, (ts[:, 2] - ts[:, 1]) * 0.01      # us
    first = ts[:256] if ntile >= 256 else ts
    print(f"{what:34s} phases {phases}: kernel {a.elapsed_time(e) * 1e3:7.1f} us, {ntile:5d} tiles | main loop {main.mean():6.2f} us "
          f"(p10 {np.percentile(main, 10):6.2f} p90 {np.percentile(main, 90):6.2f}) | epilogue {epi_t.mean():6.2f} us (p10 {np.percentile(epi_t, 10):6.2f} "
          f"p90 {np.percentile(epi_t, 90):6.2f}) | span of first-round starts {(first[:, 0].max() - first[:, 0].min()) * 0.01:6.2f} us", flush=True)
    tune(0)


for epi in ("resid", "plain"):
    for rows, label in ((263168, "1028 row panels (16 rounds)"), (65536, "256 row panels (4 rounds)"), (16384, "64 row panels (1 round)"),
                        (4096, "16 row panels: 64 CUs"), (1024, "4 row panels: 16 CUs")):
        run(rows, 1024, 1024, f"out fwd {epi} {label}", epi)
    for ph in (2, 3, 4):
        run(263168, 1024, 1024, f"out fwd {epi} 1028 panels", epi, ph)
