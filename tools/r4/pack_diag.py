"""which of the packed-vs-dense text tower equalities hold (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from oracle import clip_oracle as O
from test_clip_model_gpu import _build

import itertools
for FULL, BWD, kw in itertools.product((True,), (False,), (dict(vision_width=128, vision_layers=1, transformer_width=128, transformer_heads=2, transformer_layers=3),
           dict(vision_width=128, vision_layers=1, transformer_width=128, transformer_heads=2, transformer_layers=1), dict())):
    cfg = O.tiny_config(**kw)
    res = {}
    for packed in (True, False):
        model, orc, _ = _build(cfg, seed=5)
        model.clip_model.pack_text = packed
        batch = O.synthetic_batch(cfg, 24, seed=33)
        txt = batch["txt_batched"]
        ctx = txt.shape[1]
        if FULL:
            txt[3] = torch.randint(1, cfg["vocab_size"] - 2, (ctx,), dtype=torch.int32, generator=torch.Generator().manual_seed(7))
            txt[3, 0], txt[3, ctx - 1] = cfg["vocab_size"] - 2, cfg["vocab_size"] - 1
        dbatch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
        tok = dbatch["txt_batched"]
        model.train(); model.clip_model._ensure_flat(); model.clip_model.zero_grad()
        e = model.clip_model.encode_text(tok).detach().clone()
        if BWD:
            out = model(dbatch); out["loss"].backward()
        with torch.no_grad():
            eng = model.clip_model.encode_text(tok).clone()
        res[packed] = (e, eng)
        with torch.no_grad():
            eo = orc.encode_text(txt)
        bad = ((e.cpu() - eo).norm(dim=1) / eo.norm(dim=1))
        print("   packed", packed, "vs oracle: worst item", int(bad.argmax()), f"{float(bad.max()):.3e}", "item 3", f"{float(bad[3]):.3e}", "median", f"{float(bad.median()):.3e}")
    (ep, engp), (ed, engd) = res[True], res[False]
    d = lambda a, b: f"equal {torch.equal(a, b)} max {float((a - b).abs().max()):.3e} rows differing {int((a != b).any(1).sum())}/{a.shape[0]}"
    print(kw, "full-context item", FULL, "backward in between", BWD, "\n  grad path packed vs dense:", d(ep, ed), "\n  nograd packed vs dense:", d(engp, engd), "\n  packed grad vs nograd:", d(ep, engp),
          "\n  dense grad vs nograd:", d(ed, engd), "\n  lens", (batch["txt_batched"].argmax(-1) + 1).tolist())
