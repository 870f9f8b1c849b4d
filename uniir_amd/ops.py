"""Tensor-level wrappers over the C ABI (include/uniir_hip.h).  PyTorch is only the device-memory / stream
plumbing here: every function passes raw device pointers + the current HIP stream to libuniir_hip.so.
There is no torch fallback: a CPU tensor raises."""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import GemmDesc, check

EPI_BF16, EPI_BIAS_ACT, EPI_RESID_F32, EPI_DACT, EPI_F32, EPI_ATOMIC_F32, EPI_ACT_ONLY = range(7)
ACT_QUICKGELU, ACT_GELU_ERF, ACT_RELU = range(3)
DT_BF16, DT_F16 = 0, 1
GEMM_TIMING_STRIDE = 29   # bench.py: every 29th uniir_gemm call is bracketed by HIP events inside the library (uniir_gemm_timing);
# 441 GEMM launches per step and 441 % 29 = 6, so the sampled positions walk through every shape within a few steps


def gemm_timing_start(stride=GEMM_TIMING_STRIDE, stream=None):
    """stream: a torch.cuda.Stream -- sample the launches on that stream only (the other tower's stream shares the device: an event
    pair there would measure shared time, see uniir_gemm_timing_on)"""
    if stream is None:
        check(_lib.load().uniir_gemm_timing(int(stride)), "gemm_timing")
    else:
        check(_lib.load().uniir_gemm_timing_on(int(stride), C.c_void_p(stream.cuda_stream)), "gemm_timing_on")


def gemm_timing_stop(with_shared=False):
    """-> (sum of 2 M N K, seconds, launches) over the sampled GEMM launches [, samples left out because another stream's GEMMs
    shared the device with them: gemm_timing_start(stream=...), and whether the rule fell back to all samples because fewer than 8
    would have been left]; synchronise the device first"""
    f, ms, n, sh, fb = C.c_double(), C.c_double(), C.c_int32(), C.c_int32(), C.c_int32()
    check(_lib.load().uniir_gemm_timing_read_ex(C.byref(f), C.byref(ms), C.byref(n), C.byref(sh), C.byref(fb)), "gemm_timing_read")
    check(_lib.load().uniir_gemm_timing(0), "gemm_timing")
    if with_shared:
        return f.value, ms.value * 1e-3, n.value, sh.value, bool(fb.value)
    return f.value, ms.value * 1e-3, n.value


def gemm_timing_filter(windows, samples, merge_ms=-1.0):
    """the sampling rule alone (host arithmetic, no device): windows [(begin, end) ms], samples [(begin, duration) ms] ->
    (keep flags, fell back to all samples?)"""
    nw, n = len(windows), len(samples)
    w = (C.c_float * max(1, 2 * nw))(*[x for p in windows for x in p])
    s = (C.c_float * max(1, 2 * n))(*[x for p in samples for x in p])
    keep, fb = (C.c_uint8 * max(1, n))(), C.c_int32()
    kept = _lib.load().uniir_gemm_timing_filter(w, nw, s, n, float(merge_ms), keep, C.byref(fb))
    if kept < 0:
        check(kept, "gemm_timing_filter")
    return [bool(keep[i]) for i in range(n)], bool(fb.value)



# Reproducible reductions (include/uniir_hip.h uniir_reduce_scratch): every stream this module launches on gets a scratch buffer, so
# that bias / LayerNorm / embedding gradients are added in a fixed order and two runs of one step give the same bits.
# UNIIR_DETERMINISTIC=0 leaves the kernels on their fp32 atomics (A/B of the extra reduce launches).
RED_SCRATCH_BYTES = 64 << 20
_RED_SCRATCH = {}
_DETERMINISTIC = os.environ.get("UNIIR_DETERMINISTIC", "1") != "0"


def _stream():
    s = torch.cuda.current_stream()
    if _DETERMINISTIC:
        key = (s.device.index, s.cuda_stream)
        if key not in _RED_SCRATCH:
            if len(_RED_SCRATCH) >= 24:          # streams come and go (tests): forget the oldest entry -- in the library's table
                old = next(iter(_RED_SCRATCH))   # first, so that it never points at freed memory
                with torch.cuda.device(old[0]):
                    _lib.load().uniir_reduce_scratch(None, 0, C.c_void_p(old[1]))
                del _RED_SCRATCH[old]
            with torch.cuda.device(s.device):
                buf = torch.empty(RED_SCRATCH_BYTES, dtype=torch.uint8, device=s.device)
                check(_lib.load().uniir_reduce_scratch(C.c_void_p(buf.data_ptr()), RED_SCRATCH_BYTES, C.c_void_p(s.cuda_stream)),
                      "reduce_scratch")
            _RED_SCRATCH[key] = buf
    return C.c_void_p(s.cuda_stream)


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("uniir_amd ops need device tensors (no CPU fallback on the product path)")
    return C.c_void_p(t.data_ptr())


_SPLITK_WS = {}


def _splitk_workspace(device, nbytes, tag=None):
    """one reusable scratch buffer per (device, tag) for the split-K slabs (stream-ordered reuse: users that may run on
    different streams at the same time -- the two CLIP towers -- pass different tags)"""
    key = (device, tag)
    ws = _SPLITK_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 128 << 20), device=device, dtype=torch.uint8)
        _SPLITK_WS[key] = ws
    return ws


def uses_tile256(M, N, K):
    """mirror of gemm_shape() in csrc/gemm.hip: the 256x256 LDS-DMA kernel (the only one with the fused column-sum
    epilogue) runs when K % 64 == 0 and the problem is not tiny"""
    return K % 64 == 0 and M >= 256 and N >= 128


def gemm(A, B, C_out, M, N, K, lda, ldb, ldc, *, a_tmaj=False, b_tmaj=False, epilogue=EPI_BF16, bias=None,
         resid=None, aux=None, ldaux=0, C2=None, act=ACT_QUICKGELU, k_splits=1, alpha=1.0, dtype=DT_BF16,
         colsum=None, row_scale=None, a_rowsum=None):
    d = GemmDesc()
    d.A, d.B, d.C, d.C2 = A.data_ptr(), B.data_ptr(), C_out.data_ptr(), (C2.data_ptr() if C2 is not None else None)
    d.bias = bias.data_ptr() if bias is not None else None
    d.resid = resid.data_ptr() if resid is not None else None
    d.aux = aux.data_ptr() if aux is not None else None
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc, d.ldaux = lda, ldb, ldc, ldaux
    d.a_tmaj, d.b_tmaj = int(a_tmaj), int(b_tmaj)
    d.epilogue, d.act, d.dtype, d.k_splits, d.alpha = epilogue, act, dtype, k_splits, alpha
    d.colsum = colsum.data_ptr() if colsum is not None else None
    d.row_scale = row_scale.data_ptr() if row_scale is not None else None
    d.a_rowsum = a_rowsum.data_ptr() if a_rowsum is not None else None
    if k_splits > 1:
        ws = _splitk_workspace(A.device, 4 * k_splits * M * N)
        d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), ws.numel()
    check(_lib.load().uniir_gemm(C.byref(d), _stream()), "gemm")
    return C_out


def linear_fwd(x, w, bias=None, *, out=None, epilogue=EPI_BF16, resid=None, C2=None, act=ACT_QUICKGELU, row_scale=None):
    """y[M,N] = x[M,K] @ w[N,K]^T (+bias ...).  x, w bf16 contiguous.  row_scale (fp32 [M], EPI_RESID_F32): the branch output is
    multiplied by it before the residual is added (DropPath)."""
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, device=x.device,
                          dtype=torch.float32 if epilogue in (EPI_RESID_F32, EPI_F32) else torch.bfloat16)
    return gemm(x, w, out, M, N, K, K, K, N, epilogue=epilogue, bias=bias, resid=resid, C2=C2, act=act, row_scale=row_scale)


def linear_dgrad(dy, w, *, out=None, aux=None, act=ACT_QUICKGELU, act_out=None, colsum=None):
    """dx[M,K] = dy[M,N] @ w[N,K]  (optionally * act'(aux)).
    act_out (bf16 [M,K]): also receives act(aux), the recomputed activation (needs aux).
    colsum (fp32 [K]): += column sums of dx (the bias gradient of the layer that produced aux); fused into the GEMM
    epilogue when the 256-tile kernel runs, otherwise done by the colsum kernel on the bf16 result."""
    M, N = dy.shape
    K = w.shape[1]
    if out is None:
        out = torch.empty(M, K, device=dy.device, dtype=torch.bfloat16)
    fused = colsum is not None and aux is not None and uses_tile256(M, K, N)
    if act_out is not None and not (aux is not None and uses_tile256(M, K, N)):
        call("uniir_act_fwd", aux, act_out, aux.numel(), act)
        act_out_arg = None
    else:
        act_out_arg = act_out
    gemm(dy, w, out, M, K, N, N, K, K, b_tmaj=True, epilogue=EPI_DACT if aux is not None else EPI_BF16,
         aux=aux, ldaux=K, act=act, C2=act_out_arg, colsum=colsum if fused else None)
    if colsum is not None and not fused:
        call("uniir_colsum_bf16", out, K, colsum, M, K)
    return out


N_CU = 256  # MI355X compute units; the 256x256 GEMM kernel runs one workgroup per CU


def wgrad_splits(rows, tiles):
    """split-K factor for dW (256x256 output tiles): make tiles*splits land just under a multiple of the CU count
    (no nearly-empty last round) while every split keeps >= 16 K steps of 64 rows."""
    max_s = max(1, rows // (64 * 16))
    best, best_eff = 1, 0.0
    for s in range(1, min(max_s, 64) + 1):
        blocks = tiles * s
        rounds = -(-blocks // N_CU)
        eff = blocks / (rounds * N_CU)
        if blocks >= 0.9 * N_CU and eff >= 0.9:
            return s            # smallest split count that fills the chip (least slab traffic)
        if eff > best_eff + 1e-9:
            best, best_eff = s, eff
    return best


def linear_wgrad(dy, x, dw, *, dbias=None):
    """dw[N,K] (fp32) += dy[M,N]^T @ x[M,K];  dbias[N] (fp32) += column sums of dy (from the same pass over dy when the
    256x256 transposed kernel runs)."""
    M, N = dy.shape
    K = x.shape[1]
    tiles = ((N + 255) // 256) * ((K + 255) // 256)
    return gemm(dy, x, dw, N, K, M, N, K, K, a_tmaj=True, b_tmaj=True, epilogue=EPI_ATOMIC_F32,
                k_splits=wgrad_splits(M, tiles), a_rowsum=dbias)


def layernorm_fwd(x, gamma, beta, eps=1e-5, *, out_bf16=None, out_f32=None, rows=None, width=None, x_stride=None):
    width = width or x.shape[-1]
    rows = rows if rows is not None else x.numel() // width
    x_stride = x_stride or width
    if out_bf16 is None and out_f32 is None:
        out_bf16 = torch.empty(rows, width, device=x.device, dtype=torch.bfloat16)
    check(_lib.load().uniir_layernorm_fwd(_p(x), x_stride, _p(gamma), _p(beta), _p(out_bf16), _p(out_f32), rows,
                                          width, eps, _stream()), "layernorm_fwd")
    return out_bf16 if out_bf16 is not None else out_f32


def layernorm_bwd(x, gamma, dy, dgamma, dbeta, eps=1e-5, *, dres=None, dx=None, dx_bf16=None, rows=None,
                  width=None, x_stride=None, dx_stride=None, dx_colsum=None, branch_scale=None):
    """dx_colsum (fp32 [width]): += column sums of dx (bias gradient of the linear layer that produced x).
    branch_scale (fp32 [rows]): dx_bf16 and dx_colsum carry branch_scale[row] * dx (DropPath of the branch dx enters next)"""
    width = width or x.shape[-1]
    rows = rows if rows is not None else x.numel() // width
    x_stride = x_stride or width
    dx_stride = dx_stride or width
    if dx is None:
        dx = torch.empty(rows, width, device=x.device, dtype=torch.float32)
    if branch_scale is not None:
        check(_lib.load().uniir_layernorm_bwd_ex(_p(x), x_stride, _p(gamma), _p(dy), int(dy.dtype == torch.float32),
                                                 _p(dres), _p(dx), dx_stride, _p(dx_bf16), _p(dgamma), _p(dbeta),
                                                 _p(dx_colsum), _p(branch_scale), rows, width, eps, _stream()), "layernorm_bwd_ex")
        return dx
    check(_lib.load().uniir_layernorm_bwd(_p(x), x_stride, _p(gamma), _p(dy), int(dy.dtype == torch.float32),
                                          _p(dres), _p(dx), dx_stride, _p(dx_bf16), _p(dgamma), _p(dbeta),
                                          _p(dx_colsum), rows, width, eps, _stream()), "layernorm_bwd")
    return dx


def attention_fwd(qkv, batch, seq, heads, causal, *, out=None, lse=None):
    if out is None:
        out = torch.empty(batch * seq, heads * 64, device=qkv.device, dtype=torch.bfloat16)
    if lse is None:
        lse = torch.empty(batch, heads, seq, device=qkv.device, dtype=torch.float32)
    check(_lib.load().uniir_attention_fwd(_p(qkv), _p(out), _p(lse), batch, seq, heads, int(causal), _stream()),
          "attention_fwd")
    return out, lse


def attention_bwd(qkv, out, dout, lse, batch, seq, heads, causal, *, dqkv=None):
    if dqkv is None:
        dqkv = torch.empty_like(qkv)
    check(_lib.load().uniir_attention_bwd(_p(qkv), _p(out), _p(dout), _p(lse), _p(dqkv), batch, seq, heads,
                                          int(causal), _stream()), "attention_bwd")
    return dqkv


def attention_fwd_ex(q, q_ld, k, v, kv_ld, batch, tq, tk, heads, *, key_len=None, causal=False, drop_p=0.0, drop_seed=0,
                     row_off=None, kv_packed=False, rows=None):
    """separate Q / K / V views (head h at column h*64 of rows with the given leading dimension); out [batch*tq, heads*64].
    row_off (int32 [batch + 1], with rows = its total): the query rows are PACKED (item m = rows row_off[m] .. row_off[m + 1] - 1,
    uniir_attention_fwd_rows); kv_packed: K / V are rows of the same numbering (self-attention), else dense [batch][tk]"""
    nrow = batch * tq if row_off is None else int(rows)
    out = torch.empty(nrow, heads * 64, device=q.device, dtype=torch.bfloat16)
    lse = torch.empty(batch, heads, tq, device=q.device, dtype=torch.float32)
    if row_off is None:
        check(_lib.load().uniir_attention_fwd_ex(_p(q), q_ld, _p(k), _p(v), kv_ld, _p(out), heads * 64, _p(lse), _p(key_len),
                                                 batch, tq, tk, heads, int(causal), float(drop_p), int(drop_seed), _stream()),
              "attention_fwd_ex")
    else:
        if causal:
            raise ValueError("packed query rows: non-causal attention only (the causal packed form is uniir_attention_fwd_packed)")
        check(_lib.load().uniir_attention_fwd_rows(_p(q), q_ld, _p(k), _p(v), kv_ld, _p(out), heads * 64, _p(lse), _p(row_off),
                                                   int(bool(kv_packed)), _p(key_len), batch, tq, tk, heads, float(drop_p),
                                                   int(drop_seed), _stream()), "attention_fwd_rows")
    return out, lse


def attention_bwd_ex(q, q_ld, k, v, kv_ld, out, dout, lse, dq, dq_ld, dk, dv, dkv_ld, batch, tq, tk, heads, *,
                     key_len=None, causal=False, drop_p=0.0, drop_seed=0, row_off=None, kv_packed=False):
    if row_off is None:
        check(_lib.load().uniir_attention_bwd_ex(_p(q), q_ld, _p(k), _p(v), kv_ld, _p(out), _p(dout), heads * 64, _p(lse),
                                                 _p(key_len), _p(dq), dq_ld, _p(dk), _p(dv), dkv_ld, batch, tq, tk, heads,
                                                 int(causal), float(drop_p), int(drop_seed), _stream()), "attention_bwd_ex")
    else:
        check(_lib.load().uniir_attention_bwd_rows(_p(q), q_ld, _p(k), _p(v), kv_ld, _p(out), _p(dout), heads * 64, _p(lse),
                                                   _p(row_off), int(bool(kv_packed)), _p(key_len), _p(dq), dq_ld, _p(dk), _p(dv),
                                                   dkv_ld, batch, tq, tk, heads, float(drop_p), int(drop_seed), _stream()),
              "attention_bwd_rows")


class DropSeeds:
    """Seeds of the counter-based dropout masks of one forward pass (csrc/common.h drop_hash): one draw from torch's CPU
    generator per forward (torch.manual_seed pins the masks), one derived 32-bit seed per dropout site; backward reuses
    the seeds kept in the activation stash."""

    def __init__(self):
        self.base = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        self.n = 0

    def next(self):
        self.n += 1
        return (self.base + self.n * 0x9E3779B1) & 0xFFFFFFFF


def dropout_f32(x, p, seed, *, resid=None, out_f32=None, out_bf16=None, rowscale=None, rows_per_scale=0, row_map=None):
    """(resid +) x * mask [* rowscale[row // rows_per_scale]] of an fp32 [rows, cols] tensor -> fp32 and / or bf16.
    row_map (int32 [rows]): x holds PACKED rows, row r being row row_map[r] of the logical tensor the mask is defined on"""
    rows, cols = x.shape
    if row_map is not None:
        check(_lib.load().uniir_dropout_f32_rows(_p(x), _p(resid), _p(out_f32), _p(out_bf16), rows, cols, float(p), int(seed),
                                                 _p(row_map), _stream()), "dropout_f32_rows")
        return
    check(_lib.load().uniir_dropout_f32(_p(x), _p(resid), _p(out_f32), _p(out_bf16), rows, cols, float(p), int(seed),
                                        _p(rowscale), int(rows_per_scale), _stream()), "dropout_f32")


def dropout_bf16_(x, p, seed, *, rowscale=None, rows_per_scale=0, row_map=None):
    """in place x *= mask [* rowscale[row // rows_per_scale]] of a contiguous bf16 [rows, cols] gradient (row_map: see dropout_f32)"""
    rows, cols = x.shape
    if row_map is not None:
        check(_lib.load().uniir_dropout_bf16_rows(_p(x), _p(x), rows, cols, cols, float(p), int(seed), _p(row_map), _stream()),
              "dropout_bf16_rows")
        return
    check(_lib.load().uniir_dropout_bf16(_p(x), _p(x), rows, cols, cols, float(p), int(seed), _p(rowscale),
                                         int(rows_per_scale), _stream()), "dropout_bf16")


def call(name, *args):
    """Generic checked call: tensors are converted to device pointers, the stream is appended."""
    conv = [(_p(a) if isinstance(a, torch.Tensor) else a) for a in args]
    check(getattr(_lib.load(), name)(*conv, _stream()), name)
