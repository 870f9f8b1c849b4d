"""Pins oracle/clipff_oracle.py: G12 = transformers' T5Stack (forward + every gradient), G13 = the reference's own
CLIPFeatureFusion.encode_multimodal_input / compute_inbatch_contrastive_loss with stub encoders (concat order, mean
pooling, loss).  fp32, tolerance 2e-5 relative to the tensor's max magnitude."""
import json
import os

import numpy as np
import torch

from oracle import clip_oracle as O
from oracle import clipff_oracle as FF

G = os.path.join(os.path.dirname(__file__), "golden")


def _close(a, b, tol=2e-5):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-6)
    assert err < tol, err


def _sd(z):
    return {k[4:]: torch.from_numpy(z[k]).clone().requires_grad_(True) for k in z.files if k.startswith("sd::")}


def test_g12_t5_stack_forward_backward():
    z = np.load(os.path.join(G, "g12_t5stack.npz"))
    cfg, sd = json.loads(str(z["cfg"])), _sd(z)
    x = torch.from_numpy(z["x"]).requires_grad_(True)
    out = FF.t5_stack(sd, x, cfg)
    _close(out, z["last_hidden_state"])
    pooled = out.mean(dim=1)
    _close(pooled, z["pooled"])
    (pooled * torch.from_numpy(z["w"])).sum().backward()
    _close(x.grad, z["dx"])
    for k in z.files:
        if k.startswith("grad::"):
            _close(sd[k[6:]].grad, z[k], 5e-5)


def test_g13_reference_clipff_fusion_and_loss():
    z = np.load(os.path.join(G, "g13_clipff.npz"))
    cfg, sd = json.loads(str(z["cfg"])), _sd(z)
    txt = torch.from_numpy(z["txt_feat"]).requires_grad_(True)
    img = torch.from_numpy(z["img_feat"]).requires_grad_(True)
    emb = FF.fuse_tokens(sd, txt, img, cfg)
    _close(emb, z["emb"])
    b = emb.shape[0] // 2
    im = {"query": [[2 * i] for i in range(b)], "pos_cand": [[2 * i + 1] for i in range(b)]}
    scale = torch.tensor(float(np.log(1 / 0.07)), requires_grad=True)
    out = O.inbatch_contrastive_loss(emb, im, scale.exp())
    out["loss"].backward()
    assert abs(out["loss"].item() - float(z["loss"])) < 1e-5 and out["accuracy"].item() == float(z["acc"])
    _close(txt.grad, z["dtxt"], 5e-5)
    _close(img.grad, z["dimg"], 5e-5)
    _close(scale.grad, z["dscale"], 5e-5)
    _close(sd["block.0.layer.0.SelfAttention.q.weight"].grad, z["g_q0"], 5e-5)
    _close(sd["block.0.layer.0.SelfAttention.relative_attention_bias.weight"].grad, z["g_rel"], 5e-5)
