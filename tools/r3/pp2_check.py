"""outputs of uniir_gemm on a list of forward / dgrad problems -> one file per run; run once with UNIIR_GEMM_PP2=1 and once without,
then `python tools/r3/pp2_check.py cmp a.pt b.pt`: the 256x128 two-workgroup kernel must equal the 256x256 kernel BIT FOR BIT (same
K order, same MFMA, same epilogue arithmetic)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def run(path):
    from uniir_amd import ops
    dev = "cuda"
    torch.manual_seed(1)
    out = {}
    for (M, N, K) in [(4096, 1024, 1024), (1024, 3072, 192), (1000, 520, 256), (777, 136, 320), (65792, 1024, 1024), (2048, 4096, 4096)]:
        x = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        bias = torch.randn(N, device=dev)
        tag = f"{M}x{N}x{K}"
        out[tag + ":fwd"] = ops.linear_fwd(x, w, bias).cpu()
        g = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        f = ops.linear_fwd(x, w, bias, epilogue=ops.EPI_BIAS_ACT, C2=g)
        out[tag + ":f"], out[tag + ":g"] = f.cpu(), g.cpu()
        out[tag + ":act_only"] = ops.linear_fwd(x, w, bias, epilogue=ops.EPI_ACT_ONLY).cpu()
        res = torch.randn(M, N, device=dev)
        out[tag + ":resid"] = ops.linear_fwd(x, w, bias, epilogue=ops.EPI_RESID_F32, resid=res).cpu()
        # dgrad: dx[M,K] = dy[M,N] @ w[N,K] (B operand N-contiguous), plain and with act'(aux) + act(aux) + column sums
        dy = torch.randn(M, N, device=dev).bfloat16()
        out[tag + ":dgrad"] = ops.linear_dgrad(dy, w).cpu()
        if K >= 128 and M >= 256:
            aux = torch.randn(M, K, device=dev).bfloat16()
            act_out = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
            colsum = torch.zeros(K, device=dev)
            out[tag + ":dact"] = ops.linear_dgrad(dy, w, aux=aux, act_out=act_out, colsum=colsum).cpu()
            out[tag + ":dact_g"] = act_out.cpu()
            out[tag + ":colsum"] = colsum.cpu()
    torch.save(out, path)
    print("saved", len(out), "tensors ->", path)


def cmp(a, b):
    A, B = torch.load(a), torch.load(b)
    bad = 0
    for k in A:
        same = torch.equal(A[k], B[k])
        if not same:
            d = (A[k].float() - B[k].float()).abs().max().item()
            # column sums are atomically accumulated across workgroups: order-dependent in the last bits
            ok = k.endswith(":colsum") and d <= 1e-3 * A[k].float().abs().max().item()
            print(("~ " if ok else "!! ") + k, "max abs diff", d)
            bad += 0 if ok else 1
    print("compared", len(A), "tensors:", "ALL EQUAL" if bad == 0 else f"{bad} DIFFER")
    return bad


if __name__ == "__main__":
    if sys.argv[1] == "cmp":
        sys.exit(1 if cmp(sys.argv[2], sys.argv[3]) else 0)
    run(sys.argv[1])
