"""Autograd glue for the fuse / select+normalise / in-batch InfoNCE kernels (include/uniir_hip.h [FUSE], [NCE]).

Mirrors UniIR src/models/uniir_clip/clip_scorefusion/clip_sf.py:53-63 (fuse), :88-97 (select + normalise),
:99-103 (differentiable all-gather of p, backward = reduce-scatter SUM), :133-144 (scores, CE, accuracy).
All arithmetic is in libuniir_hip.so; torch.distributed (RCCL) carries the one exchange step.
"""
import torch

from . import comm, ops


class FuseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, txt_emb, img_emb, txt_mask, img_mask):
        n, d = txt_emb.shape
        emb = torch.empty(n, d, device=txt_emb.device, dtype=torch.float32)
        tm, im = txt_mask.to(torch.int64).contiguous(), img_mask.to(torch.int64).contiguous()
        ops.call("uniir_fuse_embeddings", txt_emb.contiguous(), img_emb.contiguous(), tm, im, emb, n, d)
        ctx.save_for_backward(tm, im)
        return emb

    @staticmethod
    def backward(ctx, demb):
        tm, im = ctx.saved_tensors
        n, d = demb.shape
        dt, di = torch.empty_like(demb), torch.empty_like(demb)
        ops.call("uniir_fuse_embeddings_bwd", demb.contiguous(), tm, im, dt, di, n, d)
        return dt, di, None, None


class InBatchNCEFn(torch.autograd.Function):
    """(emb [M,E] fp32, idx_q int32 [b], idx_p int32 [b], scale 0-dim) -> (loss, accuracy, score [b,B])"""

    @staticmethod
    def forward(ctx, emb, idx_q, idx_p, scale, gather):
        dev = emb.device
        b, E = idx_q.numel(), emb.shape[1]
        emb = emb.contiguous()
        q = torch.empty(b, E, device=dev)
        p = torch.empty(b, E, device=dev)
        qinv, pinv = torch.empty(b, device=dev), torch.empty(b, device=dev)
        ops.call("uniir_select_normalize", emb, idx_q, q, qinv, b, E)
        ops.call("uniir_select_normalize", emb, idx_p, p, pinv, b, E)
        world = comm.world() if gather else 1
        all_p = comm.all_gather_rows(p) if world > 1 else p   # RCCL all-gather over xGMI, rank-major like torch.cat
        B = all_p.shape[0]
        sc = scale.detach().reshape(1).float().contiguous()
        score = torch.empty(b, B, device=dev)
        stats = torch.empty(3 * b, device=dev)
        loss, acc = torch.empty(1, device=dev), torch.empty(1, device=dev)
        toff = comm.target_offset(b) if world > 1 else 0
        ops.call("uniir_infonce_fwd", q, all_p, sc, b, B, E, toff, score, stats, loss, acc)
        ctx.save_for_backward(q, p, all_p, qinv, pinv, idx_q, idx_p, sc, score, stats)
        ctx.meta = (b, B, E, toff, world, emb.shape[0])
        ctx.mark_non_differentiable(acc, score)
        return loss.reshape(()), acc.reshape(()), score

    @staticmethod
    def backward(ctx, dloss, _dacc, _dscore):
        q, p, all_p, qinv, pinv, idx_q, idx_p, sc, score, stats = ctx.saved_tensors
        b, B, E, toff, world, M = ctx.meta
        dev = q.device
        gbuf = torch.empty(b * B + b, device=dev)
        dq, dall = torch.empty(b, E, device=dev), torch.empty(B, E, device=dev)
        dscale = torch.empty(1, device=dev)
        dl = dloss.reshape(1).float().contiguous()
        ops.call("uniir_infonce_bwd", q, all_p, sc, score, stats, dl, b, B, E, toff, gbuf, dq, dall, dscale)
        dp = comm.reduce_scatter_rows(dall, b) if world > 1 else dall   # backward of the autograd all-gather
        demb = torch.zeros(M, E, device=dev)
        ops.call("uniir_select_normalize_bwd", q, qinv, dq, idx_q, demb, b, E)
        ops.call("uniir_select_normalize_bwd", p, pinv, dp, idx_p, demb, b, E)
        return demb, None, None, dscale.reshape(()), None


class HardNegNCEFn(torch.autograd.Function):
    """Hard-negative branch (clip_sf.py:105-131): (emb [M,E], idx_q [b], idx_p [b], idx_n [b*N], scale 0-dim,
    in_batch_neg_num) -> (loss, accuracy).  No collective: the reference's loss uses local embeddings only here."""

    @staticmethod
    def forward(ctx, emb, idx_q, idx_p, idx_n, scale, in_batch_neg_num):
        dev = emb.device
        b, E = idx_q.numel(), emb.shape[1]
        N = idx_n.numel() // b
        inb = min(b - 1, int(in_batch_neg_num))
        emb = emb.contiguous()

        def sel(idx, rows):
            out, inv = torch.empty(rows, E, device=dev), torch.empty(rows, device=dev)
            ops.call("uniir_select_normalize", emb, idx, out, inv, rows, E)
            return out, inv

        q, qinv = sel(idx_q, b)
        p, pinv = sel(idx_p, b)
        n, ninv = sel(idx_n, b * N)
        sc = scale.detach().reshape(1).float().contiguous()
        C = 1 + N + inb
        logits = torch.empty(b, C, device=dev)
        lse, rl, hit = torch.empty(b, device=dev), torch.empty(b, device=dev), torch.empty(b, device=dev)
        ops.call("uniir_hardneg_fwd", q, p, n, sc, b, N, inb, E, logits, lse, rl, hit)
        ctx.save_for_backward(q, p, n, qinv, pinv, ninv, idx_q, idx_p, idx_n, sc, logits, lse)
        ctx.meta = (b, N, inb, E, emb.shape[0])
        acc = hit.mean()
        ctx.mark_non_differentiable(acc)
        return rl.mean(), acc

    @staticmethod
    def backward(ctx, dloss, _dacc):
        q, p, n, qinv, pinv, ninv, idx_q, idx_p, idx_n, sc, logits, lse = ctx.saved_tensors
        b, N, inb, E, M = ctx.meta
        dev = q.device
        dq, dn = torch.empty(b, E, device=dev), torch.empty(b * N, E, device=dev)
        dp, dscale = torch.zeros(b, E, device=dev), torch.zeros(1, device=dev)
        dl = dloss.reshape(1).float().contiguous()
        ops.call("uniir_hardneg_bwd", q, p, n, sc, logits, lse, dl, b, N, inb, E, dq, dp, dn, dscale)
        demb = torch.zeros(M, E, device=dev)
        ops.call("uniir_select_normalize_bwd", q, qinv, dq, idx_q, demb, b, E)
        ops.call("uniir_select_normalize_bwd", p, pinv, dp, idx_p, demb, b, E)
        ops.call("uniir_select_normalize_bwd", n, ninv, dn, idx_n, demb, b * N, E)
        return demb, None, None, None, dscale.reshape(()), None
