"""The ViT-L/14 linear-layer shapes with the epilogues the train step runs them with, next to their plain forms (ms, 1024 items):
the table of VERDICT r04 item 1.  Dev tool, GPU box only.   [UNIIR_HIP_LIB=variant.so] python tools/r5/epi_forms.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniir_amd import ops

dev = "cuda"


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
    return e0.elapsed_time(e1) / iters


def main():
    items = int(os.environ.get("MB_ITEMS", "1024"))
    R, W = items * 257, 1024
    torch.manual_seed(0)
    x = torch.randn(R, W, device=dev).bfloat16()
    h4 = torch.randn(R, 4 * W, device=dev).bfloat16()
    res = torch.randn(R, W, device=dev)
    res_out = torch.empty(R, W, device=dev)
    y3 = torch.empty(R, 3 * W, device=dev, dtype=torch.bfloat16)
    y4 = torch.empty(R, 4 * W, device=dev, dtype=torch.bfloat16)
    y4b = torch.empty(R, 4 * W, device=dev, dtype=torch.bfloat16)
    y1 = torch.empty(R, W, device=dev, dtype=torch.bfloat16)
    wq = torch.randn(3 * W, W, device=dev).bfloat16()
    wf = torch.randn(4 * W, W, device=dev).bfloat16()
    wo = torch.randn(W, W, device=dev).bfloat16()
    wp = torch.randn(W, 4 * W, device=dev).bfloat16()
    b3, b4, b1 = torch.randn(3 * W, device=dev), torch.randn(4 * W, device=dev), torch.randn(W, device=dev)
    dy1 = torch.randn(R, W, device=dev).bfloat16()
    cs = torch.zeros(4 * W, device=dev)
    rs = (torch.rand(R, device=dev) > 0.1).float() / 0.9
    forms = [
        ("qkv fwd [bias]", lambda: ops.linear_fwd(x, wq, b3, out=y3)),
        ("fc fwd [bias + GELU, two outputs]", lambda: ops.linear_fwd(x, wf, b4, out=y4, epilogue=ops.EPI_BIAS_ACT, C2=y4b)),
        ("fc fwd [bias + GELU, act only]", lambda: ops.linear_fwd(x, wf, b4, out=y4, epilogue=ops.EPI_ACT_ONLY)),
        ("fc fwd plain", lambda: ops.linear_fwd(x, wf, out=y4)),
        ("out fwd [bias + fp32 residual]", lambda: ops.linear_fwd(x, wo, b1, out=res_out, epilogue=ops.EPI_RESID_F32, resid=res)),
        ("out fwd [bias + fp32 residual, row_scale]", lambda: ops.linear_fwd(x, wo, b1, out=res_out, epilogue=ops.EPI_RESID_F32, resid=res, row_scale=rs)),
        ("out fwd plain", lambda: ops.linear_fwd(x, wo, out=y1)),
        ("proj fwd [bias + fp32 residual]", lambda: ops.linear_fwd(h4, wp, b1, out=res_out, epilogue=ops.EPI_RESID_F32, resid=res)),
        ("proj fwd [bias + fp32 residual, row_scale]", lambda: ops.linear_fwd(h4, wp, b1, out=res_out, epilogue=ops.EPI_RESID_F32, resid=res, row_scale=rs)),
        ("proj fwd plain", lambda: ops.linear_fwd(h4, wp, out=y1)),
        ("proj dgrad [x act'(f), column sums]", lambda: ops.linear_dgrad(dy1, wp, out=y4, aux=h4, colsum=cs)),
        ("proj dgrad [x act'(f), act(f) out, column sums]", lambda: ops.linear_dgrad(dy1, wp, out=y4, aux=h4, act_out=y4b, colsum=cs)),
        ("proj dgrad [erf-GELU': x act'(f), column sums]", lambda: ops.linear_dgrad(dy1, wp, out=y4, aux=h4, colsum=cs, act=ops.ACT_GELU_ERF)),
        ("proj dgrad plain", lambda: ops.linear_dgrad(dy1, wp, out=y4)),
        ("fc dgrad plain (K = 4096)", lambda: ops.linear_dgrad(y4, wf, out=y1)),
        ("qkv dgrad plain (K = 3072)", lambda: ops.linear_dgrad(y3, wq, out=y1)),
    ]
    only = os.environ.get("EF_ONLY")
    for name, fn in forms:
        if only and only not in name:
            continue
        t = [timeit(fn) for _ in range(2)]
        print(f"{name:52s} {t[0]:.3f}  {t[1]:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
