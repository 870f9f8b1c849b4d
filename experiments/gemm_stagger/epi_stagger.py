"""Round-5 experiment: start stagger of the 256x256 GEMM (uniir_gemm_tune) on the shapes and fused epilogues of the ViT-L/14 step.
Prints ms per form for phases in {off, 2, 3, 4} x epilogue-time guesses.  Dev tool, GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniir_amd import ops, _lib

dev = "cuda"
lib = _lib.load()


def tune(phases, ns_k=1450, ns_epi=6000, min_rounds=6):
    for k, v in ((0, phases), (1, ns_k), (2, ns_epi), (3, min_rounds)):
        assert lib.uniir_gemm_tune(k, v) == 0


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
    forms = [
        ("qkv fwd [bias]", 16, lambda: ops.linear_fwd(x, wq, b3, out=y3)),
        ("fc fwd [bias+GELU, 2 out]", 16, lambda: ops.linear_fwd(x, wf, b4, out=y4, epilogue=ops.EPI_BIAS_ACT, C2=y4b)),
        ("out fwd [bias+resid f32]", 16, lambda: ops.linear_fwd(x, wo, b1, out=res_out, epilogue=ops.EPI_RESID_F32, resid=res)),
        ("proj fwd [bias+resid f32]", 64, lambda: ops.linear_fwd(h4, wp, b1, out=res_out, epilogue=ops.EPI_RESID_F32, resid=res)),
        ("proj dgrad [dact, colsum]", 16, lambda: ops.linear_dgrad(dy1, wp, out=y4, aux=h4, colsum=cs)),
        ("fc dgrad plain (K=4096)", 64, lambda: ops.linear_dgrad(y4, wf, out=y1)),
        ("qkv dgrad plain (K=3072)", 48, lambda: ops.linear_dgrad(y3, wq, out=y1)),
        ("out fwd plain", 16, lambda: ops.linear_fwd(x, wo, out=y1)),
        ("fc fwd plain", 16, lambda: ops.linear_fwd(x, wf, out=y4)),
    ]
    configs = [("off", 0, 0)] + [(f"P{p} epi{e // 1000}us", p, e) for p in (2, 3, 4) for e in (6000, 16000)] + [("off again", 0, 0)]
    if os.environ.get("ES_QUICK"):
        configs = [("off", 0, 0), ("P3 epi6us", 3, 6000), ("off again", 0, 0)]
    only = os.environ.get("ES_ONLY")
    for name, nk, fn in forms:
        if only and only not in name:
            continue
        line = []
        for cname, p, e in configs:
            tune(p, ns_epi=e)
            t = timeit(fn)
            line.append(f"{cname}: {t:.3f}")
        print(f"{name:28s} " + " | ".join(line), flush=True)
    tune(0)


if __name__ == "__main__":
    main()
