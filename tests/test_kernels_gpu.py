"""GPU parity tests of the individual HIP kernels (through the C ABI) against plain PyTorch fp32 references
of the same op.  bf16 kernels: tolerance stated per test (inputs are bf16-rounded before the reference runs,
so the only differences are accumulation order and the bf16 rounding of outputs)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from uniir_amd import ops
    return ops


def rel_err(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def bf(x):
    return x.to(torch.bfloat16)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 192), (200, 136, 72), (1000, 768, 1024), (520, 392, 128), (256, 256, 64), (4100, 1024, 640)])
def test_gemm_nt_bf16(M, N, K):
    ops = _ops()
    torch.manual_seed(0)
    x, w = bf(torch.randn(M, K, device=DEV)), bf(torch.randn(N, K, device=DEV))
    bias = torch.randn(N, device=DEV)
    y = ops.linear_fwd(x, w, bias)
    ref = x.float() @ w.float().t() + bias
    assert rel_err(y, ref) < 4e-3, rel_err(y, ref)
    # asymmetric check (transposition detector): first row / first col
    assert torch.allclose(y[0].float(), ref[0], rtol=2e-2, atol=2e-1)
    assert torch.allclose(y[:, 0].float(), ref[:, 0], rtol=2e-2, atol=2e-1)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (264, 320, 136), (1000, 1024, 768), (777, 192, 520), (2048, 3072, 1024)])
def test_gemm_dgrad_nn(M, N, K):
    ops = _ops()
    torch.manual_seed(1)
    dy, w = bf(torch.randn(M, N, device=DEV)), bf(torch.randn(N, K, device=DEV))
    dx = ops.linear_dgrad(dy, w)
    ref = dy.float() @ w.float()
    assert rel_err(dx, ref) < 4e-3, rel_err(dx, ref)


@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (1000, 136, 200), (5000, 768, 1024), (263, 384, 64), (4096, 392, 264), (8192, 768, 1024), (16384, 3072, 1024), (35457, 2304, 768)])
def test_gemm_wgrad_tn_splitk(M, N, K):
    ops = _ops()
    torch.manual_seed(2)
    Mp = (M + 7) // 8 * 8  # rows are the contraction dim here: no alignment requirement, but keep tail odd
    dy, x = bf(torch.randn(M, N, device=DEV)), bf(torch.randn(M, K, device=DEV))
    dw = torch.ones(N, K, device=DEV)
    ops.linear_wgrad(dy, x, dw)
    ref = dy.float().t() @ x.float() + 1.0
    assert rel_err(dw, ref) < 1e-3, rel_err(dw, ref)


@pytest.mark.parametrize("M,N,K", [(8192, 768, 1024), (16384, 3072, 1024), (20000, 520, 256), (5000, 1024, 768), (35457, 2304, 768),
                                   (263, 384, 64), (1000, 136, 200)])
def test_gemm_wgrad_with_bias_gradient(M, N, K):
    """uniir_gemm_desc.a_rowsum: the bias gradient (column sums of dy) out of the weight-gradient GEMM's own pass over dy -- inside
    the 256x256 transposed kernel (first five shapes: several K splits, an N that is not a tile multiple, row sums only from the
    first column panel; reduction lengths that are not a multiple of 64 -- 20000, 5000, and 35457 = the packed text tower's live
    rows -- run as the multiple-of-64 part on that kernel plus a tail product of the last rows, round 4) and by the separate pass
    for problems that run another kernel (last two); accumulates into dbias; bitwise repeatable"""
    ops = _ops()
    torch.manual_seed(12)
    dy, x = bf(torch.randn(M, N, device=DEV) + 0.25), bf(torch.randn(M, K, device=DEV))
    dw = torch.zeros(N, K, device=DEV)
    db = torch.full((N,), 3.0, device=DEV)
    ops.linear_wgrad(dy, x, dw, dbias=db)
    assert rel_err(dw, dy.float().t() @ x.float()) < 1e-3
    ref = dy.float().sum(0) + 3.0
    assert rel_err(db, ref) < 1e-5, rel_err(db, ref)
    dw2 = torch.zeros(N, K, device=DEV)
    ops.linear_wgrad(dy, x, dw2, dbias=torch.zeros(N, device=DEV))
    assert torch.equal(dw, dw2)


def test_gemm_row_sums_need_a_transposed_bf16_operand():
    ops = _ops()
    x, w = bf(torch.randn(256, 128, device=DEV)), bf(torch.randn(256, 128, device=DEV))
    out = torch.empty(256, 256, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):
        ops.gemm(x, w, out, 256, 256, 128, 128, 128, 256, a_rowsum=torch.zeros(256, device=DEV))        # A not transposed


def test_gemm_epilogues():
    ops = _ops()
    torch.manual_seed(3)
    M, N, K = 300, 256, 128
    x, w = bf(torch.randn(M, K, device=DEV)), bf(torch.randn(N, K, device=DEV) * 0.1)
    bias = torch.randn(N, device=DEV)
    # bias + QuickGELU: C = f, C2 = act(f)
    g = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    f = ops.linear_fwd(x, w, bias, epilogue=ops.EPI_BIAS_ACT, C2=g)
    fref = x.float() @ w.float().t() + bias
    assert rel_err(f, fref) < 4e-3
    gref = f.float() * torch.sigmoid(1.702 * f.float())
    assert rel_err(g, gref) < 4e-3
    # ACT_ONLY (forward-only passes): BIAS_ACT's second output alone, bit for bit, on the 256-tile and the general kernel,
    # QuickGELU and erf-GELU, ragged edges
    for (m2, n2, k2) in ((M, N, K), (1000, 520, 192), (100, 72, 40)):
        x2, w2_ = bf(torch.randn(m2, k2, device=DEV)), bf(torch.randn(n2, k2, device=DEV) * 0.1)
        b2 = torch.randn(n2, device=DEV)
        for act in (ops.ACT_QUICKGELU, ops.ACT_GELU_ERF):
            g2 = torch.empty(m2, n2, device=DEV, dtype=torch.bfloat16)
            ops.linear_fwd(x2, w2_, b2, epilogue=ops.EPI_BIAS_ACT, C2=g2, act=act)
            only = ops.linear_fwd(x2, w2_, b2, epilogue=ops.EPI_ACT_ONLY, act=act)
            assert only.dtype == torch.bfloat16 and torch.equal(only, g2)
    # residual fp32
    res = torch.randn(M, N, device=DEV)
    y = ops.linear_fwd(x, w, bias, epilogue=ops.EPI_RESID_F32, resid=res)
    assert y.dtype == torch.float32
    assert rel_err(y, fref + res) < 1e-5 + 4e-3
    assert (y - (fref + res)).abs().max() < 2e-3
    # dgrad * act'(f)
    dy = bf(torch.randn(M, N, device=DEV))
    w2 = bf(torch.randn(N, K, device=DEV) * 0.1)
    aux = bf(torch.randn(M, K, device=DEV))
    dx = ops.linear_dgrad(dy, w2, aux=aux)
    a = aux.float()
    s = torch.sigmoid(1.702 * a)
    dref = (dy.float() @ w2.float()) * (s * (1 + 1.702 * a * (1 - s)))
    assert rel_err(dx, dref) < 5e-3, rel_err(dx, dref)


@pytest.mark.parametrize("m,n,k", [(1024, 512, 256), (1000, 520, 192), (300, 264, 136)])
@pytest.mark.parametrize("act", [0, 1])
def test_gemm_dact_epilogue_with_and_without_the_second_output(m, n, k, act):
    """UNIIR_EPI_DACT: dx = (dy @ w) * act'(aux) [+ act(aux) -> C2] [+ column sums]: the forms the towers run (with the second output
    when act(f) is re-materialised, without it under uniir_clip_tower.stash_act).  Against fp32 torch; the two forms give bitwise the
    same dx (and the same column sums up to the order of their atomic additions); act(aux) equals the forward's BIAS_ACT second output up to one bf16 ulp on < 0.2 % of the elements; ragged
    tiles, QuickGELU and erf-GELU"""
    ops = _ops()
    torch.manual_seed(31)
    x, w = bf(torch.randn(m, k, device=DEV)), bf(torch.randn(n, k, device=DEV) * 0.2)
    b = torch.randn(n, device=DEV)
    g_fwd = torch.empty(m, n, device=DEV, dtype=torch.bfloat16)
    f = ops.linear_fwd(x, w, b, epilogue=ops.EPI_BIAS_ACT, C2=g_fwd, act=act)
    n2 = 256
    dy, w2 = bf(torch.randn(m, n2, device=DEV)), bf(torch.randn(n2, n, device=DEV) * 0.2)
    cs1, cs2 = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    g_bwd = torch.empty(m, n, device=DEV, dtype=torch.bfloat16)
    dx1 = ops.linear_dgrad(dy, w2, aux=f, act=act, act_out=g_bwd, colsum=cs1)
    dx2 = ops.linear_dgrad(dy, w2, aux=f, act=act, colsum=cs2)
    assert torch.equal(dx1, dx2) and rel_err(cs1, cs2) < 1e-6        # (the column sums meet through atomics: order varies)
    # act(aux) vs the forward's own act(f): the same function of the same bf16 values, compiled in two epilogues (contraction may
    # differ before the rounding): at most one bf16 ulp apart, on very few elements
    diff = (g_bwd.float() - g_fwd.float()).abs()
    assert float((diff > 0).float().mean()) < 2e-3 and bool((diff <= g_fwd.float().abs() * 2.0 ** -7 + 1e-30).all())
    a = f.float()
    if act == 0:
        sg = torch.sigmoid(1.702 * a)
        dref = sg * (1 + 1.702 * a * (1 - sg))
    else:
        dref = 0.5 * (1 + torch.erf(a * 0.7071067811865476)) + a * 0.3989422804014327 * torch.exp(-0.5 * a * a)
    exact = (dy.float() @ w2.float()) * dref
    assert rel_err(dx1, exact) < 4e-3 and rel_err(cs1, exact.sum(0)) < 1e-4


@pytest.mark.parametrize("rows,width", [(7, 512), (1000, 768), (513, 1024)])
def test_layernorm_fwd_bwd(rows, width):
    ops = _ops()
    torch.manual_seed(4)
    x = torch.randn(rows, width, device=DEV) * 2 + 0.5
    gamma, beta = torch.randn(width, device=DEV), torch.randn(width, device=DEV)
    y32 = torch.empty(rows, width, device=DEV)
    y16 = torch.empty(rows, width, device=DEV, dtype=torch.bfloat16)
    ops.layernorm_fwd(x, gamma, beta, 1e-5, out_bf16=y16, out_f32=y32)
    ref = torch.nn.functional.layer_norm(x, (width,), gamma, beta, 1e-5)
    assert (y32 - ref).abs().max() < 2e-5
    assert rel_err(y16, ref) < 4e-3
    dy = torch.randn(rows, width, device=DEV)
    dres = torch.randn(rows, width, device=DEV)
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr, (width,), gr, br, 1e-5).backward(dy)
    dgamma, dbeta = torch.zeros(width, device=DEV), torch.zeros(width, device=DEV)
    dxb = torch.empty(rows, width, device=DEV, dtype=torch.bfloat16)
    dx = ops.layernorm_bwd(x, gamma, dy, dgamma, dbeta, 1e-5, dres=dres, dx_bf16=dxb)
    assert (dx - (xr.grad + dres)).abs().max() < 1e-4
    assert rel_err(dgamma, gr.grad) < 1e-5
    assert rel_err(dbeta, br.grad) < 1e-5
    assert rel_err(dxb, dx) < 4e-3
    # bf16 dy path
    dgamma.zero_(); dbeta.zero_()
    dyb = bf(dy)
    xr.grad = None; gr.grad = None; br.grad = None
    torch.nn.functional.layer_norm(xr, (width,), gr, br, 1e-5).backward(dyb.float())
    dx = ops.layernorm_bwd(x, gamma, dyb, dgamma, dbeta, 1e-5)
    assert (dx - xr.grad).abs().max() < 1e-4
    assert rel_err(dgamma, gr.grad) < 1e-5


def _attn_ref(qkv, batch, seq, heads, causal):
    W = heads * 64
    q, k, v = qkv.float().view(batch, seq, 3, heads, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) * 0.125
    if causal:
        m = torch.full((seq, seq), float("-inf"), device=qkv.device).triu_(1)
        s = s + m
    p = torch.softmax(s, -1)
    o = (p @ v).permute(0, 2, 1, 3).reshape(batch * seq, W)
    lse = torch.logsumexp(s, -1)
    return o, lse


# 257 / 50 / 129 / 273: one / 18 / one / 17 live rows in the last key block; 197 (BLIP ViT, 13 tiles), 512 (the maximum), 17
@pytest.mark.parametrize("batch,seq,heads,causal", [(3, 257, 4, 0), (5, 77, 3, 1), (4, 50, 2, 0), (2, 33, 1, 1), (2, 16, 1, 0),
                                                    (2, 197, 3, 0), (2, 129, 2, 1), (2, 273, 2, 0), (1, 512, 2, 0),
                                                    (2, 17, 2, 1), (2, 145, 1, 0)])
def test_attention_fwd_bwd(batch, seq, heads, causal):
    ops = _ops()
    torch.manual_seed(5)
    W = heads * 64
    qkv = bf(torch.randn(batch * seq, 3 * W, device=DEV))
    out, lse = ops.attention_fwd(qkv, batch, seq, heads, causal)
    qr = qkv.float().requires_grad_(True)
    oref, lref = _attn_ref(qr, batch, seq, heads, causal)
    assert rel_err(out, oref) < 8e-3, rel_err(out, oref)
    assert (lse - lref).abs().max() < 2e-3
    do = bf(torch.randn(batch * seq, W, device=DEV))
    oref.backward(do.float())
    dqkv = ops.attention_bwd(qkv, out, do, lse, batch, seq, heads, causal)
    g = qr.grad.view(batch * seq, 3, W)
    d = dqkv.float().view(batch * seq, 3, W)
    for i, name in enumerate("qkv"):
        e = rel_err(d[:, i], g[:, i])
        assert e < 1.5e-2, (name, e)


@pytest.mark.parametrize("batch,seq,heads", [(37, 257, 16), (30, 197, 12), (1, 257, 1), (3, 222, 2)])
def test_attention_bwd_pair_kernel_against_the_general_kernel(batch, seq, heads):
    """the persistent pair-tile backward (csrc/attention_pair.hip: plain self-attention of 257 / 197 tokens; > 256 heads makes
    every workgroup walk over several heads) against the general backward, reached through the key-length entry point with full
    key lengths (same mathematics, general code path; 222 tokens is a shape the pair kernel does not take: both calls then run the
    same kernel).  Same products in the same order, except D = rowsum(dO * O) (summed as a tree instead of a chain) and the odd
    tile's rows (eight partial sums): differences of one bf16 ulp in < 0.1 % of the elements; and both against fp32 torch"""
    ops = _ops()
    torch.manual_seed(31 + seq)
    W = heads * 64
    qkv = bf(torch.randn(batch * seq, 3 * W, device=DEV))
    out, lse = ops.attention_fwd(qkv, batch, seq, heads, 0)
    do = bf(torch.randn(batch * seq, W, device=DEV))
    d_pair = torch.full_like(qkv, 5.0)                       # every element must be written
    ops.attention_bwd(qkv, out, do, lse, batch, seq, heads, 0, dqkv=d_pair)
    d_gen = torch.full_like(qkv, 3.0)
    klen = torch.full((batch,), seq, device=DEV, dtype=torch.int32)
    ops.attention_bwd_ex(qkv, 3 * W, qkv[:, W:], qkv[:, 2 * W:], 3 * W, out, do, lse, d_gen, 3 * W, d_gen[:, W:], d_gen[:, 2 * W:],
                         3 * W, batch, seq, seq, heads, key_len=klen)
    assert torch.isfinite(d_pair.float()).all()
    diff = (d_pair.float() - d_gen.float()).abs()
    assert float(diff.max()) <= 2.0 ** -8 * float(d_gen.float().abs().max())      # one bf16 ulp of the largest values at most ...
    assert float((diff > 0).float().mean()) < 1e-3                                  # ... in under 0.1 % of the elements
    assert rel_err(d_pair, d_gen) < 1e-4
    if batch * seq * heads <= 40000:
        qr = qkv.float().requires_grad_(True)
        oref, _ = _attn_ref(qr, batch, seq, heads, 0)
        oref.backward(do.float())
        g = qr.grad.view(batch * seq, 3, W)
        d = d_pair.float().view(batch * seq, 3, W)
        for i, name in enumerate("qkv"):
            assert rel_err(d[:, i], g[:, i]) < 1.5e-2, name


@pytest.mark.parametrize("batch,ctx,heads", [(9, 77, 12), (5, 77, 8), (3, 50, 2)])
def test_attention_on_packed_rows_equals_the_dense_causal_call(batch, ctx, heads):
    """uniir_attention_{fwd,bwd}_packed (the CLIP text tower on the rows up to each caption's EOT, csrc/tower.hip *_packed): item m
    owns rows row_off[m] .. row_off[m + 1] - 1.  Against the dense causal call on the same captions padded to ctx rows (whatever
    sits behind the EOT cannot reach a live row under the causal mask): outputs, log-sum-exps and gradients of the live rows are
    bitwise equal"""
    ops = _ops()
    g = torch.Generator().manual_seed(41 + batch)
    W = heads * 64
    lens = torch.randint(2, ctx + 1, (batch,), generator=g)
    lens[0], lens[-1] = ctx, 2                                           # a full caption and the shortest one
    off = torch.zeros(batch + 1, dtype=torch.int64)
    off[1:] = torch.cumsum(lens, 0)
    R = int(off[-1])
    row_off = off.to(torch.int32).to(DEV)
    packed = bf(torch.randn(R, 3 * W, device=DEV))
    dense = bf(torch.randn(batch * ctx, 3 * W, device=DEV))              # garbage behind the EOT
    live = torch.zeros(batch * ctx, dtype=torch.bool, device=DEV)
    for m in range(batch):
        dense[m * ctx:m * ctx + int(lens[m])] = packed[int(off[m]):int(off[m + 1])]
        live[m * ctx:m * ctx + int(lens[m])] = True
    out_d, lse_d = ops.attention_fwd(dense, batch, ctx, heads, 1)
    out_p = torch.full((R, W), 7.0, device=DEV, dtype=torch.bfloat16)
    lse_p = torch.zeros(batch, heads, ctx, device=DEV)
    ops.call("uniir_attention_fwd_packed", packed, out_p, lse_p, row_off, batch, ctx, heads, 1)
    assert torch.equal(out_p, out_d[live])
    lmask = live.view(batch, 1, ctx).expand(batch, heads, ctx)
    assert torch.equal(lse_p[lmask], lse_d[lmask])
    do_p = bf(torch.randn(R, W, device=DEV))
    do_d = torch.zeros(batch * ctx, W, device=DEV, dtype=torch.bfloat16)      # nothing behind the EOT carries a gradient
    do_d[live] = do_p
    dq_d = ops.attention_bwd(dense, out_d, do_d, lse_d, batch, ctx, heads, 1)
    dq_p = torch.full((R, 3 * W), 5.0, device=DEV, dtype=torch.bfloat16)
    ops.call("uniir_attention_bwd_packed", packed, out_p, do_p, lse_p, dq_p, row_off, batch, ctx, heads, 1)
    assert torch.equal(dq_p, dq_d[live])


@pytest.mark.parametrize("batch,tq,tk,heads,enc", [(3, 100, 197, 12, 1024), (2, 35, 257, 2, 128), (2, 197, 100, 3, 192)])
def test_cross_attention_with_key_lengths(batch, tq, tk, heads, enc):
    """BLIP MED cross-attention at its real shapes (med.py:160-232): 100 text queries x 197 image keys, separate Q and
    [K|V] tensors with their own leading dimensions, per-item key length (the padding mask as exact exclusion); fwd and
    bwd against fp32 torch"""
    ops = _ops()
    torch.manual_seed(21)
    W = heads * 64
    q = bf(torch.randn(batch * tq, W, device=DEV))
    kv = bf(torch.randn(batch * tk, 2 * W, device=DEV))
    klen = torch.tensor([tk, max(1, tk // 3), 17][:batch], device=DEV, dtype=torch.int32)
    out, lse = ops.attention_fwd_ex(q, W, kv, kv[:, W:], 2 * W, batch, tq, tk, heads, key_len=klen)
    qr = q.float().requires_grad_(True)
    kvr = kv.float().requires_grad_(True)
    qq = qr.view(batch, tq, heads, 64).transpose(1, 2)
    kk = kvr[:, :W].reshape(batch, tk, heads, 64).transpose(1, 2)
    vv = kvr[:, W:].reshape(batch, tk, heads, 64).transpose(1, 2)
    s = (qq @ kk.transpose(-1, -2)) * 0.125
    dead = torch.arange(tk, device=DEV)[None, :] >= klen[:, None].long()
    s = s.masked_fill(dead[:, None, None, :], float("-inf"))
    oref = (torch.softmax(s, -1) @ vv).transpose(1, 2).reshape(batch * tq, W)
    assert rel_err(out, oref) < 8e-3, rel_err(out, oref)
    assert (lse - torch.logsumexp(s, -1)).abs().max() < 2e-3
    do = bf(torch.randn(batch * tq, W, device=DEV))
    oref.backward(do.float())
    dq = torch.empty_like(q)
    dkv = torch.full_like(kv, 7.0)          # rows >= key_len must come back as zeros, not stay untouched
    ops.attention_bwd_ex(q, W, kv, kv[:, W:], 2 * W, out, do, lse, dq, W, dkv, dkv[:, W:], 2 * W, batch, tq, tk, heads,
                         key_len=klen)
    assert rel_err(dq, qr.grad) < 1.5e-2, rel_err(dq, qr.grad)
    assert rel_err(dkv, kvr.grad) < 1.5e-2, rel_err(dkv, kvr.grad)
    assert (dkv.float().view(batch, tk, 2 * W)[dead] == 0).all()


def test_attention_softmax_spike():
    """one key dominating at a late tile forces the running-max rescale path (online softmax)."""
    ops = _ops()
    torch.manual_seed(6)
    batch, seq, heads = 1, 257, 1
    qkv = torch.randn(batch * seq, 192, device=DEV)
    qkv[:, :64] *= 0.1
    qkv[200, 64:128] = qkv[5, :64] * 400.0  # key 200 spikes against query 5
    qkv = bf(qkv)
    out, lse = ops.attention_fwd(qkv, batch, seq, heads, 0)
    oref, lref = _attn_ref(qkv, batch, seq, heads, 0)
    assert rel_err(out, oref) < 8e-3
    assert torch.isfinite(out.float()).all()


@pytest.mark.parametrize("b,B,E,toff", [(4, 4, 8, 0), (32, 32, 512, 0), (48, 192, 768, 96), (512, 4096, 768, 1024)])
def test_infonce_fwd_bwd(b, B, E, toff):
    ops = _ops()
    torch.manual_seed(7)
    q = torch.nn.functional.normalize(torch.randn(b, E, device=DEV), dim=-1)
    ap = torch.nn.functional.normalize(torch.randn(B, E, device=DEV), dim=-1)
    scale = torch.tensor([1.0 / 0.07], device=DEV)
    score = torch.empty(b, B, device=DEV)
    stats = torch.empty(3 * b, device=DEV)
    loss, acc = torch.empty(1, device=DEV), torch.empty(1, device=DEV)
    ops.call("uniir_infonce_fwd", q, ap, scale, b, B, E, toff, score, stats, loss, acc)
    qr, pr, sr = q.clone().requires_grad_(True), ap.clone().requires_grad_(True), scale.clone().requires_grad_(True)
    sref = (qr @ pr.t()) * sr
    tgt = toff + torch.arange(b, device=DEV)
    lref = torch.nn.functional.cross_entropy(sref, tgt)
    aref = (sref.argmax(1) == tgt).float().mean()
    assert (score - sref).abs().max() < 1e-3  # north-star tolerance on fp32 logits
    assert abs(loss.item() - lref.item()) < 1e-5 * max(1.0, abs(lref.item()))
    assert acc.item() == aref.item()
    lref.backward()
    gbuf = torch.empty(b * B + b, device=DEV)
    dq, dp, ds = torch.empty(b, E, device=DEV), torch.empty(B, E, device=DEV), torch.empty(1, device=DEV)
    dl = torch.ones(1, device=DEV)
    ops.call("uniir_infonce_bwd", q, ap, scale, score, stats, dl, b, B, E, toff, gbuf, dq, dp, ds)
    assert rel_err(dq, qr.grad) < 1e-4
    assert rel_err(dp, pr.grad) < 1e-4
    assert abs(ds.item() - sr.grad.item()) < 1e-4 * max(1.0, abs(sr.grad.item()))


def test_elementwise_misc():
    ops = _ops()
    torch.manual_seed(8)
    # patchify vs unfold
    n, res, P = 3, 224, 14
    img = torch.randn(n, 3, res, res, device=DEV)
    kpad = 640
    patches = torch.empty(n * 256, kpad, device=DEV, dtype=torch.bfloat16)
    ops.call("uniir_patchify", img, patches, n, res, P, kpad)
    ref = torch.nn.functional.unfold(img, P, stride=P).transpose(1, 2).reshape(n * 256, 588)
    assert torch.equal(patches[:, :588].float(), bf(ref).float())
    assert (patches[:, 588:] == 0).all()
    # colsum
    x = bf(torch.randn(1000, 776, device=DEV))
    out = torch.ones(776, device=DEV)
    ops.call("uniir_colsum_bf16", x, 776, out, 1000, 776)
    assert rel_err(out, x.float().sum(0) + 1) < 1e-5
    # text embed + eot
    text = torch.zeros(5, 77, dtype=torch.int32, device=DEV)
    for i in range(5):
        L = 3 + 7 * i
        text[i, 0] = 49406
        text[i, 1:1 + L] = torch.randint(1, 49405, (L,), device=DEV, dtype=torch.int32)
        text[i, 1 + L] = 49407
    tok, pos = torch.randn(49408, 64, device=DEV), torch.randn(77, 64, device=DEV)
    xo = torch.empty(5, 77, 64, device=DEV)
    eot = torch.empty(5, dtype=torch.int32, device=DEV)
    ops.call("uniir_text_embed", text, tok, pos, xo, eot, 5, 77, 64, 49408)
    assert torch.equal(xo, tok[text.long()] + pos)
    assert torch.equal(eot.long(), text.argmax(-1))
    # adamw vs torch
    p = torch.randn(1003, device=DEV)
    g = torch.randn(1003, device=DEV)
    pt = p.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pt], lr=1e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.2)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    pb = torch.empty(1003, device=DEV, dtype=torch.bfloat16)
    pp = torch.zeros(1008, device=DEV)[:1003]
    pp.copy_(p)
    for step in range(1, 4):
        pt.grad = g.clone()
        opt.step()
        ops.call("uniir_adamw_step", pp, g, m, v, pb, 1003, 1e-3, 0.9, 0.98, 1e-6, 0.2, step, 1.0)
    assert (pp - pt.detach()).abs().max() < 1e-6
    assert torch.equal(pb.float(), bf(pp).float())


@pytest.mark.parametrize("a_t,b_t", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(256, 256, 192), (512, 384, 256), (777, 520, 320), (2048, 1024, 1024), (1000, 264, 4096)])
def test_gemm_pingpong_all_layouts(a_t, b_t, M, N, K):
    """256x256x64 ping-pong loop (gemm_core_pp.h): every operand layout, 3..64 K steps, ragged M / N edges; each case
    runs 3 times (race screen: the loop's LDS-DMA ordering is by counted vmcnt + barriers only) and must be bitwise
    repeatable and equal to the compiler-scheduled loop's result (same MFMA order per output element)."""
    ops = _ops()
    torch.manual_seed(7)
    Mp = (M + 7) // 8 * 8 if a_t else M
    A = bf(torch.randn((K, Mp) if a_t else (Mp, K), device=DEV))
    B = bf(torch.randn((K, N) if b_t else (N, K), device=DEV))
    Af = (A.float().t() if a_t else A.float())[:Mp]
    Bf = B.float() if b_t else B.float().t()
    ref = Af @ Bf
    outs = []
    for _ in range(3):
        C = torch.empty(Mp, N, device=DEV, dtype=torch.float32)
        ops.gemm(A, B, C, Mp, N, K, Mp if a_t else K, N if b_t else K, N, a_tmaj=bool(a_t), b_tmaj=bool(b_t),
                 epilogue=ops.EPI_F32)
        outs.append(C)
    assert rel_err(outs[0], ref) < 2e-3, rel_err(outs[0], ref)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    Cb = torch.empty(Mp, N, device=DEV, dtype=torch.bfloat16)
    ops.gemm(A, B, Cb, Mp, N, K, Mp if a_t else K, N if b_t else K, N, a_tmaj=bool(a_t), b_tmaj=bool(b_t))
    assert rel_err(Cb, ref) < 4e-3
    # split-K accumulate through the atomic epilogue and through the slab workspace
    Cacc = torch.ones(Mp, N, device=DEV, dtype=torch.float32)
    ops.gemm(A, B, Cacc, Mp, N, K, Mp if a_t else K, N if b_t else K, N, a_tmaj=bool(a_t), b_tmaj=bool(b_t),
             epilogue=ops.EPI_ATOMIC_F32, k_splits=max(1, K // 192))
    assert rel_err(Cacc, ref + 1.0) < 2e-3


@pytest.mark.parametrize("rows,width", [(9, 512), (700, 768)])
def test_rmsnorm_fwd_bwd(rows, width):
    """T5LayerNorm (no mean subtraction, no bias) through the LayerNorm kernels' rms mode"""
    ops = _ops()
    torch.manual_seed(11)
    x = torch.randn(rows, width, device=DEV) * 1.7 + 0.3
    gamma = torch.randn(width, device=DEV) * 0.2 + 1.0
    xr, gr = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True)
    ref = gr * (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-6))
    y32 = torch.empty(rows, width, device=DEV)
    y16 = torch.empty(rows, width, device=DEV, dtype=torch.bfloat16)
    ops.call("uniir_rmsnorm_fwd", x, width, gamma, y16, y32, rows, width, 1e-6)
    assert rel_err(y32, ref) < 1e-5 and rel_err(y16, ref) < 4e-3
    dy = torch.randn(rows, width, device=DEV)
    dres = torch.randn(rows, width, device=DEV)
    ref.backward(dy)
    dx, dg = torch.empty(rows, width, device=DEV), torch.zeros(width, device=DEV)
    ops.call("uniir_rmsnorm_bwd", x, width, gamma, dy, 1, dres, dx, width, None, dg, rows, width, 1e-6)
    assert rel_err(dx, xr.grad + dres) < 1e-5
    assert rel_err(dg, gr.grad) < 1e-4


@pytest.mark.parametrize("batch,seq,heads", [(3, 334, 2), (2, 50, 3), (1, 16, 1)])
def test_attention_relative_bias_fwd_bwd(batch, seq, heads):
    """T5-style attention: no 1/sqrt(d) scaling, bucketed relative position bias, gradient of the bias table"""
    import math
    ops = _ops()
    torch.manual_seed(13)
    W = heads * 64
    qkv = bf(torch.randn(batch * seq, 3 * W, device=DEV) * 0.35)
    nb = 32
    emb = torch.randn(nb, heads, device=DEV)
    pos = torch.arange(seq)
    rel = pos[None, :] - pos[:, None]                                     # key - query
    half = nb // 2
    n = rel.abs()
    large = 8 + (torch.log(n.float().clamp_min(1) / 8) / math.log(128 / 8) * (half - 8)).long()
    bucket2d = (rel > 0).long() * half + torch.where(n < 8, n, torch.min(large, torch.full_like(large, half - 1)))
    offs = torch.arange(-(seq - 1), seq)                                  # table indexed by key - query + seq - 1
    on = offs.abs()
    olarge = 8 + (torch.log(on.float().clamp_min(1) / 8) / math.log(128 / 8) * (half - 8)).long()
    table = ((offs > 0).long() * half + torch.where(on < 8, on, torch.min(olarge, torch.full_like(olarge, half - 1)))).to(torch.int32).to(DEV)
    out = torch.empty(batch * seq, W, device=DEV, dtype=torch.bfloat16)
    lse = torch.empty(batch, heads, seq, device=DEV)
    ops.call("uniir_attention_rel_fwd", qkv, out, lse, emb, table, nb, 1.0, batch, seq, heads, 0.0, 0)
    # reference
    embr = emb.clone().requires_grad_(True)
    x = qkv.float().view(batch, seq, 3, heads, 64).requires_grad_(True)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    bias = embr[bucket2d.to(DEV)].permute(2, 0, 1)
    p = torch.softmax(q @ k.transpose(-1, -2) + bias[None], dim=-1)
    ref = (p @ v).transpose(1, 2).reshape(batch * seq, W)
    assert rel_err(out, ref) < 8e-3, rel_err(out, ref)
    dout = bf(torch.randn(batch * seq, W, device=DEV))
    ref.backward(dout.float())
    dqkv = torch.empty_like(qkv)
    drel = torch.zeros(nb, heads, device=DEV)
    ops.call("uniir_attention_rel_bwd", qkv, out, dout, lse, dqkv, emb, table, nb, 1.0, drel, batch, seq, heads, 0.0, 0)
    assert rel_err(dqkv, x.grad.reshape(batch * seq, 3 * W)) < 2e-2, rel_err(dqkv, x.grad.reshape(batch * seq, 3 * W))
    assert rel_err(drel, embr.grad) < 2e-2, rel_err(drel, embr.grad)


def test_dropout_masks_and_elementwise():
    """counter-based dropout: keep rate, determinism, the same mask through the fp32 / bf16 / DropPath forms"""
    ops = _ops()
    rows, cols, p, seed = 513, 768, 0.1, 12345
    mask = torch.empty(rows * cols, device=DEV)
    ops.call("uniir_dropout_mask", mask, mask.numel(), p, seed)
    keep = (mask > 0).float().mean().item()
    assert abs(keep - 0.9) < 3e-3 and set(mask.unique().tolist()) <= {0.0, mask.max().item()}
    assert abs(mask.max().item() - 1 / 0.9) < 1e-6
    m2 = torch.empty_like(mask)
    ops.call("uniir_dropout_mask", m2, m2.numel(), p, seed)
    assert torch.equal(mask, m2)
    ops.call("uniir_dropout_mask", m2, m2.numel(), p, seed + 1)
    assert (mask != m2).float().mean().item() > 0.1
    # adjacent elements / rows are not correlated (a cheap independence check)
    mk = (mask > 0).float().view(rows, cols)
    assert abs((mk[:, 1:] * mk[:, :-1]).mean().item() - 0.81) < 5e-3 and abs((mk[1:] * mk[:-1]).mean().item() - 0.81) < 5e-3
    torch.manual_seed(3)
    x, res = torch.randn(rows, cols, device=DEV), torch.randn(rows, cols, device=DEV)
    rs = (torch.rand(rows // 19, device=DEV) > 0.3).float() / 0.7              # DropPath factors, 19 rows per item
    y32, y16 = torch.empty_like(x), torch.empty(rows, cols, device=DEV, dtype=torch.bfloat16)
    ops.call("uniir_dropout_f32", x, res, y32, y16, rows, cols, p, seed, rs, 19)
    ref = res + x * mask.view(rows, cols) * rs.repeat_interleave(19)[:, None]
    assert torch.allclose(y32, ref, atol=1e-6) and rel_err(y16, ref) < 4e-3
    xb = bf(x)
    yb = torch.empty_like(xb)
    ops.call("uniir_dropout_bf16", xb, yb, rows, cols, cols, p, seed, None, 0)
    assert rel_err(yb, xb.float() * mask.view(rows, cols)) < 4e-3
    ops.call("uniir_dropout_f32", x, None, y32, None, rows, cols, 0.0, 0, None, 0)     # p = 0: identity
    assert torch.equal(y32, x)


@pytest.mark.parametrize("rel", [False, True])
def test_attention_probability_dropout(rel):
    """P V with P * mask / keep, softmax statistics of the full P; backward with the regenerated mask (both kernels)"""
    ops = _ops()
    torch.manual_seed(17)
    batch, seq, heads, p, seed = 2, 77, 2, 0.1, 777
    W = heads * 64
    qkv = bf(torch.randn(batch * seq, 3 * W, device=DEV) * (0.35 if rel else 1.0))
    mask = torch.empty(batch * heads * seq * seq, device=DEV)
    ops.call("uniir_dropout_mask", mask, mask.numel(), p, seed)
    mask = mask.view(batch, heads, seq, seq)
    x = qkv.float().view(batch, seq, 3, heads, 64).requires_grad_(True)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    out = torch.empty(batch * seq, W, device=DEV, dtype=torch.bfloat16)
    if rel:
        emb = torch.randn(32, heads, device=DEV)
        table = torch.randint(0, 32, (2 * seq - 1,), device=DEV, dtype=torch.int32)
        lse = torch.empty(batch, heads, seq, device=DEV)
        ops.call("uniir_attention_rel_fwd", qkv, out, lse, emb, table, 32, 1.0, batch, seq, heads, p, seed)
        pos = torch.arange(seq, device=DEV)
        bias = emb[table[(pos[None, :] - pos[:, None] + seq - 1)].long()].permute(2, 0, 1)
        logits = q @ k.transpose(-1, -2) + bias[None]
    else:
        out, lse = ops.attention_fwd_ex(qkv, 3 * W, qkv[:, W:], qkv[:, 2 * W:], 3 * W, batch, seq, seq, heads, drop_p=p, drop_seed=seed)
        logits = q @ k.transpose(-1, -2) / 8.0
    ref = ((torch.softmax(logits, dim=-1) * mask) @ v).transpose(1, 2).reshape(batch * seq, W)
    assert rel_err(out, ref) < 8e-3, rel_err(out, ref)
    dout = bf(torch.randn(batch * seq, W, device=DEV))
    ref.backward(dout.float())
    dqkv = torch.empty_like(qkv)
    if rel:
        drel = torch.zeros(32, heads, device=DEV)
        ops.call("uniir_attention_rel_bwd", qkv, out, dout, lse, dqkv, emb, table, 32, 1.0, drel, batch, seq, heads, p, seed)
    else:
        ops.attention_bwd_ex(qkv, 3 * W, qkv[:, W:], qkv[:, 2 * W:], 3 * W, out, dout, lse, dqkv, 3 * W, dqkv[:, W:],
                             dqkv[:, 2 * W:], 3 * W, batch, seq, seq, heads, drop_p=p, drop_seed=seed)
    assert rel_err(dqkv, x.grad.reshape(batch * seq, 3 * W)) < 2e-2, rel_err(dqkv, x.grad.reshape(batch * seq, 3 * W))


def test_device_image_transform_is_bit_exact():
    """uniir_image_preprocess == Pillow's BICUBIC resize (golden G14, made by Pillow) + crop + (x / 255 - mean) / std of
    the C oracle, bit for bit: down- / up-scaling, one unchanged axis, identity, saturated patterns, BLIP's square resize"""
    import os
    import numpy as np
    from oracle import c_oracle
    from uniir_amd import clip_front
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "g14_image.npz"))
    mean, std = clip_front._MEAN, clip_front._STD
    imgs, wants = [], []
    for c in range(int(z["n_cases"])):
        img, res = z[f"c{c}_img"], z[f"c{c}_res"]
        n, oh, ow, top, left = (int(v) for v in z[f"c{c}_geom"])
        assert clip_front.resize_geometry(img.shape[0], img.shape[1], n) == (oh, ow, top, left)
        crop = res[top:top + n, left:left + n].astype(np.float32) / np.float32(255.0)        # from PILLOW's own output
        want = (crop.transpose(2, 0, 1) - np.float32(mean)[:, None, None]) / np.float32(std)[:, None, None]
        got = clip_front.preprocess_on_device([img], n, DEV)[0].cpu().numpy()
        assert np.array_equal(got, want.astype(np.float32)), (c, np.abs(got - want).max())
        assert np.array_equal(got, c_oracle.clip_preprocess(img, n, mean, std))
    # a batch of differently sized images in one call, and BLIP's square resize without crop against the oracle
    batch = [z["c0_img"], z["c1_img"], z["c3_img"]]
    out = clip_front.preprocess_on_device(batch, 32, DEV).cpu().numpy()
    for i, img in enumerate(batch):
        assert np.array_equal(out[i], c_oracle.clip_preprocess(img, 32, mean, std))
    sq = clip_front.preprocess_on_device([z["c4_img"]], 48, DEV, center_crop=False)[0].cpu().numpy()
    r = c_oracle.resize_bicubic(z["c4_img"], 48, 48).astype(np.float32) / np.float32(255.0)
    assert np.array_equal(sq, ((r.transpose(2, 0, 1) - np.float32(mean)[:, None, None]) / np.float32(std)[:, None, None]))


def test_gemm_large_shape_every_epilogue_on_two_row_ranges():
    """260 row panels x 4 column panels = 1040 tiles = 4 rounds of the 256 CUs + 16 tiles (the shape of the N = 1024 GEMMs of the
    headline step).  Every epilogue of the step, checked separately on the rows of the whole rounds and on the rows of the partial
    last round, incl. the per-row operands (resid, row_scale, aux).  (Round 3 ran the partial round as a second launch of the
    128-tile kernel: +0.5 % slower, experiments/gemm_pp2/remainder_launch.inc.)"""
    ops = _ops()
    torch.manual_seed(9)
    M, N, K = 260 * 256, 1024, 512
    lo = 256 * 256                                            # first remainder row
    x = bf(torch.randn(M, K, device=DEV))
    w = bf(torch.randn(N, K, device=DEV) * 0.05)
    bias = torch.randn(N, device=DEV)
    ref = x.float() @ w.float().t() + bias

    def both(got, want, tol):
        for sl in (slice(0, lo), slice(lo, M)):
            assert rel_err(got[sl], want[sl]) < tol, (sl, rel_err(got[sl], want[sl]))

    y = ops.linear_fwd(x, w, bias)                             # bf16 out
    both(y, ref, 4e-3)
    g = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    f = ops.linear_fwd(x, w, bias, epilogue=ops.EPI_BIAS_ACT, C2=g)
    both(f, ref, 4e-3)
    both(g, f.float() * torch.sigmoid(1.702 * f.float()), 4e-3)
    only = ops.linear_fwd(x, w, bias, epilogue=ops.EPI_ACT_ONLY)
    assert torch.equal(only, g)
    res = torch.randn(M, N, device=DEV)
    scale = torch.rand(M, device=DEV) + 0.5
    z = ops.linear_fwd(x, w, bias, epilogue=ops.EPI_RESID_F32, resid=res, row_scale=scale)
    both(z, ref * scale[:, None] + res, 4e-3)
    # dgrad with act'(aux), the recomputed activation and the column sums (fc bias gradient)
    dy = bf(torch.randn(M, N, device=DEV))
    w2 = bf(torch.randn(N, K, device=DEV) * 0.05)
    aux = bf(torch.randn(M, K, device=DEV))
    act_out = torch.empty(M, K, device=DEV, dtype=torch.bfloat16)
    colsum = torch.zeros(K, device=DEV)
    dx = ops.linear_dgrad(dy, w2, aux=aux, act_out=act_out, colsum=colsum)
    a = aux.float()
    sg = torch.sigmoid(1.702 * a)
    dref = (dy.float() @ w2.float()) * (sg * (1 + 1.702 * a * (1 - sg)))
    both(dx, dref, 5e-3)
    both(act_out, a * sg, 4e-3)
    assert rel_err(colsum, dref.sum(0)) < 5e-3


def test_cross_workgroup_sums_are_reproducible():
    """uniir_reduce_scratch (attached per stream by uniir_amd.ops): the reductions that many workgroups contribute to store their
    partials and add them in a fixed order -- each of them, run twice on the same inputs, returns the same bits, and still matches
    the fp32 reference: the GEMM epilogue's column sums (DACT, 40 row panels), the weight-gradient kernel's row sums (8 K splits x 3
    column panels), LayerNorm's weight / bias gradients and dx column sums (hundreds of workgroups), uniir_colsum_bf16, and the token
    embedding's scatter-add (dense and packed rows; every caption shares ids 1 and 2, the rest collide among 64 ids; one id has
    more rows than the sorted-bucket limit and all-zero gradients, like the padding id of an unpacked batch)."""
    ops = _ops()
    from uniir_amd import _lib
    ops._stream()
    assert ops._DETERMINISTIC and ops._RED_SCRATCH          # this stream has its scratch buffer
    torch.manual_seed(3)
    M, N, K = 10240, 1024, 512

    def twice(fn):
        a, b = fn(), fn()
        assert torch.equal(a, b), float((a - b).abs().max())
        return a

    # (1) DACT epilogue column sums
    dy, w, f = bf(torch.randn(M, K, device=DEV)), bf(torch.randn(K, N, device=DEV) * 0.05), bf(torch.randn(M, N, device=DEV))

    def dact():
        cs = torch.zeros(N, device=DEV)
        ops.linear_dgrad(dy, w, out=torch.empty(M, N, device=DEV, dtype=torch.bfloat16), aux=f, colsum=cs, act=ops.ACT_QUICKGELU)
        return cs
    cs = twice(dact)
    ff = f.float()
    sg = torch.sigmoid(1.702 * ff)
    ref = ((dy.float() @ w.float()) * (sg * (1 + 1.702 * ff * (1 - sg)))).sum(0)
    assert rel_err(cs, ref) < 2e-3, rel_err(cs, ref)
    # (2) weight gradient + bias gradient (row sums of dy^T)
    x = bf(torch.randn(M, K, device=DEV))
    dy2 = bf(torch.randn(M, 768, device=DEV) + 0.25)

    def wgrad():
        dw, db = torch.zeros(768, K, device=DEV), torch.zeros(768, device=DEV)
        ops.linear_wgrad(dy2, x, dw, dbias=db)
        return torch.cat([dw.flatten(), db])
    out = twice(wgrad)
    assert rel_err(out[-768:], dy2.float().sum(0)) < 1e-5
    # (3) LayerNorm backward
    rows, width = 50000, 1024
    xx, gamma = torch.randn(rows, width, device=DEV), torch.randn(width, device=DEV)
    dyy = bf(torch.randn(rows, width, device=DEV))

    def lnb():
        dg, db, dc = (torch.zeros(width, device=DEV) for _ in range(3))
        _lib.check(_lib.load().uniir_layernorm_bwd(xx.data_ptr(), width, gamma.data_ptr(), dyy.data_ptr(), 0, None,
                                                    torch.empty(rows, width, device=DEV).data_ptr(), width, None, dg.data_ptr(),
                                                    db.data_ptr(), dc.data_ptr(), rows, width, 1e-5, ops._stream()), "ln_bwd")
        return torch.cat([dg, db, dc])
    out = twice(lnb)
    assert rel_err(out[width:2 * width], dyy.float().sum(0)) < 1e-5
    # (4) column sums of a bf16 matrix
    def colsum():
        o = torch.zeros(1024, device=DEV)
        ops.call("uniir_colsum_bf16", dyy, 1024, o, rows, 1024)
        return o
    assert rel_err(twice(colsum), dyy.float().sum(0)) < 1e-5
    # (5) token-embedding gradient
    n, ctx, wdt, vocab = 300, 77, 128, 4096
    ids = torch.randint(3, 67, (n, ctx), dtype=torch.int32)
    ids[:, 0], lens = 1, torch.randint(5, ctx + 1, (n,))
    ids[torch.arange(n), lens - 1] = 2
    ids = torch.where(torch.arange(ctx).unsqueeze(0) < lens.unsqueeze(1), ids, torch.zeros_like(ids))      # id 0 behind the last token
    dx = torch.randn(n * ctx, wdt, device=DEV)
    live = (torch.arange(ctx).unsqueeze(0) < lens.unsqueeze(1)).flatten().to(DEV)
    dx_dense = dx * live.unsqueeze(1)                                                                      # zeros on the dead rows
    assert int((ids == 0).sum()) > 4096                                                                    # the unsorted bucket
    idd = ids.to(DEV)

    def dense():
        dt, dp = torch.zeros(vocab, wdt, device=DEV), torch.zeros(ctx, wdt, device=DEV)
        ops.call("uniir_text_embed_bwd", idd, dx_dense, dt, dp, n, ctx, wdt, vocab)
        return dt
    ref = torch.zeros(vocab, wdt, device=DEV).index_add_(0, idd.flatten().long(), dx_dense)
    got = twice(dense)
    assert (got - ref).abs().max() < 2e-4 * float(ref.abs().max())
    row_off = torch.zeros(n + 1, dtype=torch.int32)
    row_off[1:] = torch.cumsum(lens, 0)
    dxp = dx_dense[live].contiguous()
    ro = row_off.to(DEV)

    def packed():
        dt, dp = torch.zeros(vocab, wdt, device=DEV), torch.zeros(ctx, wdt, device=DEV)
        ops.call("uniir_text_embed_bwd_packed", idd, dxp, ro, dt, dp, n, ctx, wdt, vocab)
        return dt
    gotp = twice(packed)
    assert torch.equal(gotp[1:], got[1:])          # every id but the padding id: the same rows in the same (row) order
    assert (gotp - ref).abs().max() < 2e-4 * float(ref.abs().max())
