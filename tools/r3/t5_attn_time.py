"""T5-fusion attention of CLIP_FF (334 tokens, 512 sequences, 12 heads): plain kernels vs the relative-position-bias kernels, with and
without the bias gradient -- where do the 5.4 ms per backward launch go?"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uniir_amd import ops  # noqa: E402

DEV = "cuda"
batch, seq, heads, nb = 512, int(os.environ.get("SEQ", "334")), 12, 32
W = heads * 64
torch.manual_seed(0)
qkv = (torch.randn(batch * seq, 3 * W, device=DEV) * 0.35).bfloat16()
emb = torch.randn(nb, heads, device=DEV)
offs = torch.arange(-(seq - 1), seq)
on = offs.abs()
half = nb // 2
olarge = 8 + (torch.log(on.float().clamp_min(1) / 8) / math.log(128 / 8) * (half - 8)).long()
table = ((offs > 0).long() * half + torch.where(on < 8, on, torch.min(olarge, torch.full_like(olarge, half - 1)))).to(torch.int32).to(DEV)
out = torch.empty(batch * seq, W, device=DEV, dtype=torch.bfloat16)
lse = torch.empty(batch, heads, seq, device=DEV)
dout = torch.randn(batch * seq, W, device=DEV).bfloat16()
dqkv = torch.empty_like(qkv)
drel = torch.zeros(nb, heads, device=DEV)


def t(name, fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / n:.3f} ms", flush=True)


t("plain fwd", lambda: ops.attention_fwd(qkv, batch, seq, heads, False, out=out, lse=lse))
t("plain bwd", lambda: ops.attention_bwd(qkv, out, dout, lse, batch, seq, heads, False, dqkv=dqkv))
t("rel fwd", lambda: ops.call("uniir_attention_rel_fwd", qkv, out, lse, emb, table, nb, 1.0, batch, seq, heads, 0.0, 0))
t("rel bwd + d bias", lambda: ops.call("uniir_attention_rel_bwd", qkv, out, dout, lse, dqkv, emb, table, nb, 1.0, drel, batch, seq, heads, 0.0, 0))
t("rel bwd, no d bias", lambda: ops.call("uniir_attention_rel_bwd", qkv, out, dout, lse, dqkv, emb, table, nb, 1.0, None, batch, seq, heads, 0.0, 0))
t("rel fwd, dropout 0.1", lambda: ops.call("uniir_attention_rel_fwd", qkv, out, lse, emb, table, nb, 1.0, batch, seq, heads, 0.1, 7))
t("rel bwd + d bias, dropout 0.1", lambda: ops.call("uniir_attention_rel_bwd", qkv, out, dout, lse, dqkv, emb, table, nb, 1.0, drel, batch, seq, heads, 0.1, 7))
