"""Per-kernel timing on the GPU box (dev tool): prints TF/s and GB/s per kernel; not part of the shipped path."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uniir_amd import ops, retrieval

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
    return e0.elapsed_time(e1) / iters * 1e-3


def attn_only(items):
    torch.manual_seed(0)
    shapes = [(257, 16, 0, items), (197, 16, 0, items), (50, 12, 0, items), (77, 12, 1, items)]
    if os.environ.get("MB_TOKENS"):          # one plain self-attention shape only (PMC passes per token count, tools/attn_pmc.sh)
        shapes = [(int(os.environ["MB_TOKENS"]), 16, 0, items)]
    for (T, H, causal, b) in shapes:
        qkv = torch.randn(b * T, 3 * H * 64, device=dev).bfloat16()
        out, lse = ops.attention_fwd(qkv, b, T, H, causal)
        t = timeit(lambda: ops.attention_fwd(qkv, b, T, H, causal, out=out, lse=lse), iters=20)
        fl = 4 * b * H * T * T * 64
        do = torch.randn_like(out)
        dqkv = torch.empty_like(qkv)
        t2 = timeit(lambda: ops.attention_bwd(qkv, out, do, lse, b, T, H, causal, dqkv=dqkv), iters=20)
        print(f"attn T={T} H={H} b={b} causal={causal}: fwd {t*1e3:.3f} ms {fl/t/1e12:.1f} TF/s | bwd {t2*1e3:.3f} ms {2.5*fl/t2/1e12:.1f} TF/s "
              f"(legacy_stage={os.environ.get('UNIIR_ATTN_LEGACY_STAGE', '0')}) chk out {out.float().abs().sum().item():.1f} dqkv {dqkv.float().abs().sum().item():.1f}")
    if os.environ.get("MB_TOKENS"):
        return
    # BLIP MED cross-attention: 100 text queries x 197 image keys, 12 heads
    b, tq, tk, H = items, 100, 197, 12
    W = H * 64
    q = torch.randn(b * tq, W, device=dev).bfloat16()
    kv = torch.randn(b * tk, 2 * W, device=dev).bfloat16()
    out, lse = ops.attention_fwd_ex(q, W, kv, kv[:, W:], 2 * W, b, tq, tk, H)
    t = timeit(lambda: ops.attention_fwd_ex(q, W, kv, kv[:, W:], 2 * W, b, tq, tk, H), iters=20)
    print(f"cross attn 100x197 fwd: {t*1e3:.3f} ms {4*b*H*tq*tk*64/t/1e12:.1f} TF/s")


def main():
    items = int(os.environ.get("MB_ITEMS", "256"))
    R = items * 257
    print(f"rows={R}")
    only = os.environ.get("MB_ONLY", "")
    if only == "attn":
        return attn_only(items)
    for (N, K, name) in ([] if os.environ.get("MB_SKIP_GEMM") else [(3072, 1024, "qkv"), (1024, 1024, "out"), (4096, 1024, "fc"), (1024, 4096, "proj")]):
        x = torch.randn(R, K, device=dev).bfloat16()
        w = torch.randn(N, K, device=dev).bfloat16()
        y = torch.empty(R, N, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: ops.linear_fwd(x, w, out=y))
        print(f"gemm fwd NT {name}: {t*1e3:.3f} ms  {2*R*N*K/t/1e12:.1f} TF/s")
        dy = torch.randn(R, N, device=dev).bfloat16()
        dx = torch.empty(R, K, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: ops.linear_dgrad(dy, w, out=dx))
        print(f"gemm dgrad NN {name}: {t*1e3:.3f} ms  {2*R*N*K/t/1e12:.1f} TF/s")
        dw = torch.zeros(N, K, device=dev)
        t = timeit(lambda: ops.linear_wgrad(dy, x, dw))
        print(f"gemm wgrad TN {name}: {t*1e3:.3f} ms  {2*R*N*K/t/1e12:.1f} TF/s")
        del x, w, y, dy, dx, dw
    # the same shapes with the epilogues the train step runs them with (csrc/tower.hip): the "in-step vs isolated" comparison
    if not os.environ.get("MB_SKIP_GEMM"):
        W = 1024
        x = torch.randn(R, W, device=dev).bfloat16()
        h4 = torch.randn(R, 4 * W, device=dev).bfloat16()
        res = torch.randn(R, W, device=dev)
        res_out = torch.empty(R, W, device=dev)
        for (N, K, name, inp) in [(3 * W, W, "qkv", x), (4 * W, W, "fc", x), (W, W, "out", x), (W, 4 * W, "proj", h4)]:
            w = torch.randn(N, K, device=dev).bfloat16()
            bias = torch.randn(N, device=dev)
            if name == "qkv":
                y = torch.empty(R, N, device=dev, dtype=torch.bfloat16)
                fn = lambda: ops.linear_fwd(inp, w, bias, out=y)
                what = "bias"
            elif name == "fc":
                y, y2 = torch.empty(R, N, device=dev, dtype=torch.bfloat16), torch.empty(R, N, device=dev, dtype=torch.bfloat16)
                fn = lambda: ops.linear_fwd(inp, w, bias, out=y, epilogue=ops.EPI_BIAS_ACT, C2=y2)
                what = "bias + GELU, two outputs"
            else:
                fn = lambda: ops.linear_fwd(inp, w, bias, out=res_out, epilogue=ops.EPI_RESID_F32, resid=res)
                what = "bias + fp32 residual in / out"
            t = timeit(fn)
            print(f"gemm fwd NT {name} [{what}]: {t*1e3:.3f} ms  {2*R*N*K/t/1e12:.1f} TF/s")
            del w
        w2 = torch.randn(W, 4 * W, device=dev).bfloat16()
        dy = torch.randn(R, W, device=dev).bfloat16()
        dx, act = torch.empty(R, 4 * W, device=dev, dtype=torch.bfloat16), torch.empty(R, 4 * W, device=dev, dtype=torch.bfloat16)
        cs = torch.zeros(4 * W, device=dev)
        t = timeit(lambda: ops.linear_dgrad(dy, w2, out=dx, aux=h4, act_out=act, colsum=cs))
        print(f"gemm dgrad NN proj [x act'(f), act(f) out, column sums]: {t*1e3:.3f} ms  {2*R*W*4*W/t/1e12:.1f} TF/s")
        t = timeit(lambda: ops.linear_dgrad(dy, w2, out=dx, aux=h4, colsum=cs))
        print(f"gemm dgrad NN proj [x act'(f), column sums; no act(f) output]: {t*1e3:.3f} ms  {2*R*W*4*W/t/1e12:.1f} TF/s")
        t = timeit(lambda: ops.linear_dgrad(dy, w2, out=dx))
        print(f"gemm dgrad NN proj [plain, same operands]: {t*1e3:.3f} ms  {2*R*W*4*W/t/1e12:.1f} TF/s")
        del x, h4, res, res_out, w2, dy, dx, act
    # attention
    for (T, H, causal, b) in [(257, 16, 0, items), (77, 12, 1, items)]:
        qkv = torch.randn(b * T, 3 * H * 64, device=dev).bfloat16()
        out, lse = ops.attention_fwd(qkv, b, T, H, causal)
        t = timeit(lambda: ops.attention_fwd(qkv, b, T, H, causal, out=out, lse=lse))
        fl = 4 * b * H * T * T * 64
        print(f"attn fwd T={T}: {t*1e3:.3f} ms  {fl/t/1e12:.1f} TF/s (dense count)")
        do = torch.randn_like(out)
        dqkv = torch.empty_like(qkv)
        t = timeit(lambda: ops.attention_bwd(qkv, out, do, lse, b, T, H, causal, dqkv=dqkv))
        print(f"attn bwd T={T}: {t*1e3:.3f} ms  {2.5*fl/t/1e12:.1f} TF/s (2.5x fwd count)")
    # layernorm
    x = torch.randn(R, 1024, device=dev)
    g, bb = torch.ones(1024, device=dev), torch.zeros(1024, device=dev)
    y = torch.empty(R, 1024, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: ops.layernorm_fwd(x, g, bb, out_bf16=y))
    print(f"ln fwd: {t*1e3:.3f} ms  {R*1024*6/t/1e9:.0f} GB/s")
    dy = torch.randn(R, 1024, device=dev).bfloat16()
    dx = torch.empty(R, 1024, device=dev)
    dg, db = torch.zeros(1024, device=dev), torch.zeros(1024, device=dev)
    t = timeit(lambda: ops.layernorm_bwd(x, g, dy, dg, db, dx=dx))
    print(f"ln bwd: {t*1e3:.3f} ms  {R*1024*10/t/1e9:.0f} GB/s")
    # top-k
    n = int(os.environ.get("MB_POOL", "700000"))
    pool = torch.randn(n, 768, device=dev).half()
    ids = torch.arange(n, device=dev)
    shard = retrieval.PoolShard(pool, ids)
    for nq in (16, 64, 128, 1024):
        q = torch.randn(nq, 768, device=dev).half()
        t = timeit(lambda: retrieval.search_shard(shard, q, 10), iters=5, warm=2)
        print(f"topk nq={nq} n={n}: {t*1e3:.3f} ms  pool {n*768*2/t/1e9:.0f} GB/s  {2*nq*n*768/t/1e12:.1f} TF/s  {nq*n/t/1e6:.0f} Mscores/s")


if __name__ == "__main__":
    main()
