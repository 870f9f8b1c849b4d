"""Golden vectors for the CLIP_FF fusion path, generated here by importing
  * transformers.models.t5.modeling_t5.T5Stack (the third-party stack clip_ff.py:15,80-96 instantiates) -> G12 and
  * the reference's own CLIPFeatureFusion.encode_multimodal_input / compute_inbatch_contrastive_loss
    (src/models/uniir_clip/clip_featurefusion/clip_ff.py:161-265) with stub encoders that return given token features -> G13.
Run:  python tests/golden/make_golden_clipff.py     (needs /root/reference; writes tests/golden/g12_t5stack.npz, g13_clipff.npz)
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/src"
TINY_T5 = dict(num_layers=2, num_decoder_layers=2, num_heads=2, d_model=128, d_kv=64, d_ff=256, dropout_rate=0.0, vocab_size=8)


def tiny_stack(seed):
    from transformers.models.t5 import T5Config
    from transformers.models.t5.modeling_t5 import T5Stack
    conf = T5Config()
    for k, v in TINY_T5.items():
        setattr(conf, k, v)
    torch.manual_seed(seed)
    st = T5Stack(conf)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in st.named_parameters():       # HF leaves these at default init; make every term non-trivial
            if p.ndim == 1:
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * (0.5 if "relative_attention_bias" in n else p.shape[-1] ** -0.5))
    return st


def g12():
    st = tiny_stack(121)
    st.train()
    x = torch.randn(3, 10, TINY_T5["d_model"], generator=torch.Generator().manual_seed(123), requires_grad=True)
    out = st(inputs_embeds=x, attention_mask=None, use_cache=False, return_dict=True).last_hidden_state
    pooled = out.mean(dim=1)
    w = torch.randn(pooled.shape, generator=torch.Generator().manual_seed(124))
    (pooled * w).sum().backward()
    res = {f"sd::{k}": v.detach().numpy() for k, v in st.state_dict().items() if "embed_tokens" not in k}
    res.update({f"grad::{n}": p.grad.numpy() for n, p in st.named_parameters() if p.grad is not None})
    print("params without grad:", [n for n, p in st.named_parameters() if p.grad is None])
    res.update(x=x.detach().numpy(), last_hidden_state=out.detach().numpy(), pooled=pooled.detach().numpy(), w=w.numpy(),
               dx=x.grad.numpy(), cfg=json.dumps(TINY_T5))
    np.savez_compressed(os.path.join(HERE, "g12_t5stack.npz"), **res)
    print("g12 ok", out.shape, sorted(k for k in res if k.startswith("sd::"))[:3])


def g13():
    """reference CLIPFeatureFusion methods on an object assembled without clip.load: encoders are stubs returning fixed
    token features, the fusion stack is the tiny T5Stack; pins concat order, mean pooling and the loss"""
    clip_stub = types.ModuleType("clip")
    clip_stub.load = lambda *a, **k: None
    clip_stub.tokenize = lambda *a, **k: None
    model_stub = types.ModuleType("clip.model")
    model_stub.VisionTransformer = type("VisionTransformer", (torch.nn.Module,), {})
    sys.modules["clip"], sys.modules["clip.model"] = clip_stub, model_stub
    sys.path.insert(0, REF_SRC)
    from models.uniir_clip.clip_featurefusion import clip_ff
    m = clip_ff.CLIPFeatureFusion.__new__(clip_ff.CLIPFeatureFusion)
    torch.nn.Module.__init__(m)
    m.t5_layers = tiny_stack(131)
    m.loss_function = torch.nn.CrossEntropyLoss()
    m.gather_embeddings, m.in_batch_neg_num = False, 0
    scale = torch.nn.Parameter(torch.tensor(float(np.log(1 / 0.07))))
    m.clip_model = types.SimpleNamespace(logit_scale=scale)
    b, Lt, Li, D = 4, 6, 5, TINY_T5["d_model"]
    g = torch.Generator().manual_seed(133)
    txt_feat = torch.randn(2 * b, Lt, D, generator=g, requires_grad=True)
    img_feat = torch.randn(2 * b, Li, D, generator=g, requires_grad=True)
    m.encode_text = lambda t: txt_feat
    m.encode_image = lambda i: img_feat
    batch = {"txt_batched": None, "image_batched": None, "txt_mask_batched": torch.ones(2 * b), "image_mask_batched": torch.ones(2 * b),
             "index_mapping": {"query": [[2 * i] for i in range(b)], "pos_cand": [[2 * i + 1] for i in range(b)]}}
    m.train()
    emb = m.encode_multimodal_input(None, None, batch["txt_mask_batched"], batch["image_mask_batched"])
    out = m.compute_inbatch_contrastive_loss(batch)
    out["loss"].backward()
    res = {f"sd::{k}": v.detach().numpy() for k, v in m.t5_layers.state_dict().items() if "embed_tokens" not in k}
    res.update(txt_feat=txt_feat.detach().numpy(), img_feat=img_feat.detach().numpy(), emb=emb.detach().numpy(),
               loss=out["loss"].item(), acc=out["accuracy"].item(), dtxt=txt_feat.grad.numpy(), dimg=img_feat.grad.numpy(),
               dscale=scale.grad.numpy(), cfg=json.dumps(TINY_T5),
               g_q0=m.t5_layers.block[0].layer[0].SelfAttention.q.weight.grad.numpy(),
               g_rel=m.t5_layers.block[0].layer[0].SelfAttention.relative_attention_bias.weight.grad.numpy())
    np.savez_compressed(os.path.join(HERE, "g13_clipff.npz"), **res)
    print("g13 ok", out["loss"].item(), out["accuracy"].item())


if __name__ == "__main__":
    torch.set_num_threads(8)
    for w in (sys.argv[1:] or ["g12", "g13"]):
        globals()[w]()
