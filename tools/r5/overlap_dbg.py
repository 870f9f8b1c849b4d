"""Which bits move between the one-stream and the two-stream tower order (and between two one-stream runs)?  tools/r5 diagnostic."""
import os, sys
from types import SimpleNamespace
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "uniir_amd", "src"))
from oracle import clip_oracle as O
from models.uniir_clip.clip_scorefusion.clip_sf import CLIPScoreFusion
from uniir_amd import clip_model
from uniir_amd.trainer import NativeTrainer

cfg = O.tiny_config(vision_width=128, vision_layers=3, transformer_width=128, transformer_heads=2, transformer_layers=3)
clip_model.CLIP_CONFIGS["tiny-test"] = cfg

def run(overlap, masks=True, steps=4):
    sd = O.init_state_dict(cfg, seed=5)
    config = SimpleNamespace(model=SimpleNamespace(gather_embeddings=False), data_config=SimpleNamespace(in_batch_neg_num=0))
    model = CLIPScoreFusion("tiny-test", device="cuda", config=config)
    model.clip_model.load_state_dict(sd, strict=True)
    clip = model.clip_model
    clip.overlap_towers = overlap
    tr = NativeTrainer(model, lr=1e-3, t_total=10)
    rec = []
    for it in range(steps + 1):
        batch = O.synthetic_batch(cfg, 24, seed=100 + it)
        db = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
        if masks:
            g = torch.Generator().manual_seed(it)
            for key in ("txt_mask_batched", "image_mask_batched"):
                m = (torch.rand(db[key].shape[0], generator=g) > 0.33).to(db[key].dtype)
                db[key] = m.view(db[key].shape).cuda()
        clip._ensure_flat()
        clip.zero_grad()
        model.train()
        out = model(db)
        out["loss"].backward()
        torch.cuda.synchronize()
        g32 = clip._flat["g32"].clone()
        out2 = tr.train_step(db)
        torch.cuda.synchronize()
        rec.append((float(out["loss"].detach()), g32, clip._flat["p32"].clone()))
    return clip, rec

def where(clip, a, b):
    fl = clip._flat
    bad = (a != b).nonzero().flatten()
    names = {}
    for n, o in fl["off"].items():
        sz = 1
        for d in fl["shapes"][n]:
            sz *= d
        c = int(((bad >= o) & (bad < o + sz)).sum())
        if c:
            names[n] = (c, float((a[o:o + sz] - b[o:o + sz]).abs().max()), float(b[o:o + sz].abs().max()))
    return names

for masks in (True, False):
    ca, A = run(False, masks)
    cb, B = run(False, masks)
    cc, Cc = run(True, masks)
    for tag, X in (("one-stream again", B), ("two-stream", Cc)):
        for it, ((l0, g0, p0), (l1, g1, p1)) in enumerate(zip(A, X)):
            dg, dp = int((g0 != g1).sum()), int((p0 != p1).sum())
            print(f"masks={masks} {tag} step {it}: loss equal {l0 == l1}  grad elements differing {dg}  weights differing {dp}")
            if dg:
                w = where(ca, g1, g0)
                print("    first differing step:", len(w), "tensors;", list(w.items())[:6])
                break
