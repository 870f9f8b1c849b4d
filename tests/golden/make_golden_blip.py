#!/usr/bin/env python
"""Golden fixtures for the BLIP_FF rows (a17-a20), captured by IMPORTING THE REFERENCE
(/root/reference/src, read-only) in the build container with the import shims of SURVEY.md section 8c:
  * transformers.modeling_utils gets apply_chunking_to_forward / prune_linear_layer / find_pruneable_heads_and_indices
    back (moved in transformers 5) so that uniir_blip/backbone/med.py imports;
  * stub modules for timm (PatchEmbed = Conv2d(3, D, P, P, bias=True) + flatten/transpose, DropPath = identity at p 0,
    trunc_normal_), fairscale (checkpoint_wrapper = identity);
  * blip_ff.create_vit / init_tokenizer monkey-patched to tiny sizes / no download; dropout probabilities set to 0 in
    the tiny med_config so that train-mode results are deterministic.

Fixtures:  g6_med.npz (BertModel multimodal: hidden states, pooler output, grads), g7_vit.npz (VisionTransformer
tokens + grads), g8_blipff.npz (BLIPFeatureFusion.compute_contrastive_loss, two steps: loss, accuracy, queues, grads).
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/src"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)


def install_shims():
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu
    mu.apply_chunking_to_forward = pu.apply_chunking_to_forward
    mu.prune_linear_layer = pu.prune_linear_layer
    mu.find_pruneable_heads_and_indices = getattr(pu, "find_pruneable_heads_and_indices", lambda *a, **k: (set(), None))

    class PatchEmbed(torch.nn.Module):
        def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
            super().__init__()
            self.grid_size = (img_size // patch_size, img_size // patch_size)
            self.num_patches = self.grid_size[0] * self.grid_size[1]
            self.proj = torch.nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

        def forward(self, x):
            return self.proj(x).flatten(2).transpose(1, 2)

    class DropPath(torch.nn.Module):
        def __init__(self, p=0.0):
            super().__init__()
            self.p = p

        def forward(self, x):
            assert self.p == 0.0 or not self.training
            return x

    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    mod("timm")
    mod("timm.models")
    mod("timm.models.vision_transformer", _cfg=lambda **k: {}, PatchEmbed=PatchEmbed)
    mod("timm.models.registry", register_model=lambda f: f)
    mod("timm.models.layers", trunc_normal_=torch.nn.init.trunc_normal_, DropPath=DropPath)
    mod("timm.models.helpers", named_apply=lambda *a, **k: None, adapt_input_conv=lambda *a, **k: None)
    mod("timm.models.hub", download_cached_file=lambda *a, **k: None)
    mod("cv2")   # backbone/transform/randaugment.py imports it at module level (train-time augmentation, unused here)
    mod("fairscale")
    mod("fairscale.nn")
    mod("fairscale.nn.checkpoint")
    mod("fairscale.nn.checkpoint.checkpoint_activations", checkpoint_wrapper=lambda m, *a, **k: m)
    tv = mod("torchvision")
    tvt = mod("torchvision.transforms", Resize=object, Compose=object, Normalize=object, ToTensor=object,
              RandomResizedCrop=object, RandomHorizontalFlip=object, InterpolationMode=types.SimpleNamespace(BICUBIC=3))
    tv.transforms = tvt
    mod("torchvision.transforms.functional", InterpolationMode=types.SimpleNamespace(BICUBIC=3))
    mod("torchvision.datasets")
    mod("torchvision.datasets.utils", download_url=lambda *a, **k: None)


TINY_MED = dict(architectures=["BertModel"], attention_probs_dropout_prob=0.0, hidden_act="gelu", hidden_dropout_prob=0.0,
                hidden_size=128, initializer_range=0.02, intermediate_size=256, layer_norm_eps=1e-12,
                max_position_embeddings=64, model_type="bert", num_attention_heads=2, num_hidden_layers=2,
                pad_token_id=0, type_vocab_size=2, vocab_size=200, encoder_width=128, add_cross_attention=True)
TINY_VIT = dict(img_size=32, patch_size=16, embed_dim=128, depth=2, num_heads=2)


def perturb(module, seed):
    """make LayerNorm / bias parameters non-trivial so that every term of the restatement is exercised"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in module.parameters():
            if p.ndim < 2:
                p.add_(0.05 * torch.randn(p.shape, generator=g))
            else:
                p.add_(0.02 * torch.randn(p.shape, generator=g))


def med_inputs(seed, n=3, L=20, Timg=5, enc_w=128, vocab=200):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1, vocab, (n, L), generator=g)
    ids[:, 0] = 101 % vocab
    mask = torch.ones(n, L, dtype=torch.long)
    for i in range(n):
        valid = int(torch.randint(4, L + 1, (1,), generator=g))
        mask[i, valid:] = 0
        ids[i, valid:] = 0
    img = torch.randn(n, Timg, enc_w, generator=g)
    return ids, mask, img


def g6():
    from models.uniir_blip.backbone import med
    med.BertPreTrainedModel.init_weights = lambda s: s.apply(s._init_weights)
    med.BertPreTrainedModel.get_head_mask = lambda s, h, n, *a, **k: [None] * n
    cfg = med.BertConfig(**TINY_MED)
    torch.manual_seed(61)
    model = med.BertModel(config=cfg, add_pooling_layer=True)
    perturb(model, 62)
    model.train()   # dropout probabilities are 0 in the tiny config
    ids, mask, img = med_inputs(63)
    img = img.clone().requires_grad_(True)
    out = model(ids, attention_mask=mask, encoder_hidden_states=img,
                encoder_attention_mask=torch.ones(img.shape[:-1], dtype=torch.long), return_dict=True)
    w = torch.randn(out.pooler_output.shape, generator=torch.Generator().manual_seed(64))
    (out.pooler_output * w).sum().backward()
    res = {f"sd::{k}": v.detach().numpy() for k, v in model.state_dict().items() if v.dtype.is_floating_point}
    res.update(ids=ids.numpy(), mask=mask.numpy(), img=img.detach().numpy(), w=w.numpy(),
               last_hidden_state=out.last_hidden_state.detach().numpy(), pooler_output=out.pooler_output.detach().numpy(),
               dimg=img.grad.numpy(), cfg=json.dumps(TINY_MED))
    for name in ["encoder.layer.0.attention.self.query.weight", "encoder.layer.1.crossattention.self.key.weight",
                 "encoder.layer.0.output.LayerNorm.weight", "embeddings.word_embeddings.weight", "pooler.dense.bias",
                 "encoder.layer.1.intermediate.dense.weight", "embeddings.LayerNorm.bias"]:
        res[f"grad::{name}"] = dict(model.named_parameters())[name].grad.numpy()
    np.savez_compressed(os.path.join(HERE, "g6_med.npz"), **res)
    print("g6 ok", out.pooler_output.shape, float(out.pooler_output.abs().mean()))


def g7():
    from models.uniir_blip.backbone.vit import VisionTransformer
    torch.manual_seed(71)
    vit = VisionTransformer(**TINY_VIT)
    perturb(vit, 72)
    vit.train()
    x = torch.randn(3, 3, 32, 32, generator=torch.Generator().manual_seed(73))
    y = vit(x)
    w = torch.randn(y.shape, generator=torch.Generator().manual_seed(74))
    (y * w).sum().backward()
    res = {f"sd::{k}": v.detach().numpy() for k, v in vit.state_dict().items()}
    res.update(x=x.numpy(), y=y.detach().numpy(), w=w.numpy(), cfg=json.dumps(TINY_VIT))
    for name in ["blocks.0.attn.qkv.weight", "blocks.1.mlp.fc2.bias", "patch_embed.proj.weight", "pos_embed", "cls_token",
                 "norm.weight", "blocks.0.norm1.bias"]:
        res[f"grad::{name}"] = dict(vit.named_parameters())[name].grad.numpy()
    np.savez_compressed(os.path.join(HERE, "g7_vit.npz"), **res)
    print("g7 ok", y.shape)


def g8():
    import torch.distributed as dist
    from models.uniir_blip.backbone import med
    from models.uniir_blip.backbone.vit import VisionTransformer
    from models.uniir_blip.blip_featurefusion import blip_ff
    med.BertPreTrainedModel.init_weights = lambda s: s.apply(s._init_weights)
    med.BertPreTrainedModel.get_head_mask = lambda s, h, n, *a, **k: [None] * n
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29547")
        dist.init_process_group("gloo", rank=0, world_size=1)
    blip_ff.create_vit = lambda vit, image_size, *a, **k: (VisionTransformer(**TINY_VIT), TINY_VIT["embed_dim"])
    blip_ff.init_tokenizer = lambda: None
    cfg_path = os.path.join("/tmp", "tiny_med_config.json")
    json.dump(TINY_MED, open(cfg_path, "w"))
    torch.manual_seed(81)
    E, K, b = TINY_MED["hidden_size"], 16, 4
    model = blip_ff.BLIPFeatureFusion(med_config=cfg_path, image_size=32, vit="base", embed_dim=E, queue_size=K,
                                      momentum=0.9, config=types.SimpleNamespace(tokenizer_max_length=20))
    perturb(model.visual_encoder, 82)
    perturb(model.text_encoder, 83)
    model.copy_params()
    model.train()
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    M = 2 * b
    out = {}
    steps = []
    for step, alpha in enumerate([0.0, 0.4]):
        ids, mask, _ = med_inputs(90 + step, n=M, L=20)
        img = torch.randn(M, 3, 32, 32, generator=torch.Generator().manual_seed(95 + step))
        batch = {
            "txt_batched": types.SimpleNamespace(input_ids=ids, attention_mask=mask),
            "image_batched": img,
            "txt_mask_batched": torch.ones(M, dtype=torch.long), "image_mask_batched": torch.ones(M, dtype=torch.long),
            "p_did_list": torch.tensor([7, 8, 7, 9]) + 10 * step,     # a repeated positive id inside the batch
            "index_mapping": {"query": [[2 * i] for i in range(b)], "pos_cand": [[2 * i + 1] for i in range(b)]},
        }
        model.zero_grad()
        res = model(batch, alpha=alpha)
        res["loss"].backward()
        steps.append((res["loss"].item(), res["accuracy"].item()))
        out.update({f"s{step}_ids": ids.numpy(), f"s{step}_mask": mask.numpy(), f"s{step}_img": img.numpy(),
                    f"s{step}_pdid": batch["p_did_list"].numpy(), f"s{step}_alpha": alpha,
                    f"s{step}_loss": res["loss"].item(), f"s{step}_acc": res["accuracy"].item(),
                    f"s{step}_dtemp": model.temp.grad.numpy(),
                    f"s{step}_g_vit_qkv0": model.visual_encoder.blocks[0].attn.qkv.weight.grad.numpy(),
                    f"s{step}_g_txt_q0": model.text_encoder.encoder.layer[0].attention.self.query.weight.grad.numpy(),
                    f"s{step}_g_pool": model.text_encoder.pooler.dense.weight.grad.numpy(),
                    f"s{step}_query_queue": model.query_queue.numpy().copy(), f"s{step}_cand_queue": model.cand_queue.numpy().copy(),
                    f"s{step}_idx_queue": model.idx_queue.numpy().copy(), f"s{step}_ptr": model.new_ptr_queue.numpy().copy(),
                    f"s{step}_m_vit_qkv0": model.visual_encoder_m.blocks[0].attn.qkv.weight.detach().numpy().copy()})
        # a plain SGD nudge between the two steps so that the momentum update has something to average
        with torch.no_grad():
            for p in list(model.visual_encoder.parameters()) + list(model.text_encoder.parameters()):
                if p.grad is not None:
                    p.add_(-0.05 * p.grad)
    for k, v in sd0.items():
        if "encoder_m." in k:      # momentum copies equal the online weights at step 0 (copy_params)
            continue
        if v.dtype.is_floating_point or k.endswith("idx_queue") or k.endswith("new_ptr_queue"):
            out[f"sd0::{k}"] = v.numpy()
    out.update(med_cfg=json.dumps(TINY_MED), vit_cfg=json.dumps(TINY_VIT), queue_size=K, momentum=0.9)
    np.savez_compressed(os.path.join(HERE, "g8_blipff.npz"), **out)
    print("g8 ok", steps)


def g8h():
    """BLIP_FF with hard negatives (blip_ff.py:127-131,159-170,196-205,233-246): two steps, N = 1 negative per query;
    torch.manual_seed before each step pins the reference's `torch.rand(1) < 0.5` enqueue choice"""
    import torch.distributed as dist
    from models.uniir_blip.backbone import med
    from models.uniir_blip.backbone.vit import VisionTransformer
    from models.uniir_blip.blip_featurefusion import blip_ff
    med.BertPreTrainedModel.init_weights = lambda s: s.apply(s._init_weights)
    med.BertPreTrainedModel.get_head_mask = lambda s, h, n, *a, **k: [None] * n
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29548")
        dist.init_process_group("gloo", rank=0, world_size=1)
    blip_ff.create_vit = lambda vit, image_size, *a, **k: (VisionTransformer(**TINY_VIT), TINY_VIT["embed_dim"])
    blip_ff.init_tokenizer = lambda: None
    cfg_path = os.path.join("/tmp", "tiny_med_config.json")
    json.dump(TINY_MED, open(cfg_path, "w"))
    torch.manual_seed(181)
    E, K, b, N = TINY_MED["hidden_size"], 16, 4, 1
    model = blip_ff.BLIPFeatureFusion(med_config=cfg_path, image_size=32, vit="base", embed_dim=E, queue_size=K,
                                      momentum=0.9, config=types.SimpleNamespace(tokenizer_max_length=20))
    perturb(model.visual_encoder, 182)
    perturb(model.text_encoder, 183)
    model.copy_params()
    model.train()
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    M = b * (2 + N)
    out = {}
    steps = []
    for step, alpha in enumerate([0.4, 0.4]):
        ids, mask, _ = med_inputs(190 + step, n=M, L=20)
        img = torch.randn(M, 3, 32, 32, generator=torch.Generator().manual_seed(195 + step))
        batch = {
            "txt_batched": types.SimpleNamespace(input_ids=ids, attention_mask=mask),
            "image_batched": img,
            "txt_mask_batched": torch.ones(M, dtype=torch.long), "image_mask_batched": torch.ones(M, dtype=torch.long),
            "p_did_list": torch.tensor([7, 8, 7, 9]) + 10 * step,
            "nc_dids_list": (torch.tensor([[31], [32], [8], [33]]) + 10 * step),     # one negative equals another row's positive id
            "index_mapping": {"query": [[3 * i] for i in range(b)], "pos_cand": [[3 * i + 1] for i in range(b)],
                              "neg_cand_list": [[3 * i + 2] for i in range(b)]},
        }
        model.zero_grad()
        torch.manual_seed(1000 + step)
        res = model(batch, alpha=alpha)
        res["loss"].backward()
        steps.append((res["loss"].item(), res["accuracy"].item()))
        out.update({f"s{step}_ids": ids.numpy(), f"s{step}_mask": mask.numpy(), f"s{step}_img": img.numpy(),
                    f"s{step}_pdid": batch["p_did_list"].numpy(), f"s{step}_ncdid": batch["nc_dids_list"].numpy(),
                    f"s{step}_alpha": alpha, f"s{step}_loss": res["loss"].item(), f"s{step}_acc": res["accuracy"].item(),
                    f"s{step}_dtemp": model.temp.grad.numpy(),
                    f"s{step}_g_pool": model.text_encoder.pooler.dense.weight.grad.numpy(),
                    f"s{step}_query_queue": model.query_queue.numpy().copy(), f"s{step}_cand_queue": model.cand_queue.numpy().copy(),
                    f"s{step}_idx_queue": model.idx_queue.numpy().copy(), f"s{step}_ptr": model.new_ptr_queue.numpy().copy()})
        with torch.no_grad():
            for p in list(model.visual_encoder.parameters()) + list(model.text_encoder.parameters()):
                if p.grad is not None:
                    p.add_(-0.05 * p.grad)
    for k, v in sd0.items():
        if "encoder_m." in k:
            continue
        if v.dtype.is_floating_point or k.endswith("idx_queue") or k.endswith("new_ptr_queue"):
            out[f"sd0::{k}"] = v.numpy()
    out.update(med_cfg=json.dumps(TINY_MED), vit_cfg=json.dumps(TINY_VIT), queue_size=K, momentum=0.9)
    np.savez_compressed(os.path.join(HERE, "g8h_blipff_hardneg.npz"), **out)
    print("g8h ok", steps)


def g8s():
    """BLIPScoreFusion (blip_scorefusion/blip_sf.py): ViT cls -> vision_proj, BERT mode="text" cls -> text_proj, masked
    sum, then the same momentum / queue loss; two steps with a text-only and an image-only item in the batch"""
    import torch.distributed as dist
    from models.uniir_blip.backbone import med
    from models.uniir_blip.backbone.vit import VisionTransformer
    from models.uniir_blip.blip_scorefusion import blip_sf
    med.BertPreTrainedModel.init_weights = lambda s: s.apply(s._init_weights)
    med.BertPreTrainedModel.get_head_mask = lambda s, h, n, *a, **k: [None] * n
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29549")
        dist.init_process_group("gloo", rank=0, world_size=1)
    blip_sf.create_vit = lambda vit, image_size, *a, **k: (VisionTransformer(**TINY_VIT), TINY_VIT["embed_dim"])
    blip_sf.init_tokenizer = lambda: None
    cfg_path = os.path.join("/tmp", "tiny_med_config.json")
    json.dump(TINY_MED, open(cfg_path, "w"))
    torch.manual_seed(281)
    E, K, b = 64, 16, 4
    model = blip_sf.BLIPScoreFusion(med_config=cfg_path, image_size=32, vit="base", embed_dim=E, queue_size=K,
                                    momentum=0.9, config=types.SimpleNamespace(tokenizer_max_length=20))
    perturb(model.visual_encoder, 282)
    perturb(model.text_encoder, 283)
    perturb(model.vision_proj, 284)
    perturb(model.text_proj, 285)
    model.copy_params()
    model.train()
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    M = 2 * b
    out = {}
    steps = []
    for step, alpha in enumerate([0.0, 0.4]):
        ids, mask, _ = med_inputs(290 + step, n=M, L=20)
        img = torch.randn(M, 3, 32, 32, generator=torch.Generator().manual_seed(295 + step))
        tmask, imask = torch.ones(M, dtype=torch.long), torch.ones(M, dtype=torch.long)
        tmask[3] = 0      # image-only item
        imask[4] = 0      # text-only item
        batch = {
            "txt_batched": types.SimpleNamespace(input_ids=ids, attention_mask=mask),
            "image_batched": img, "txt_mask_batched": tmask, "image_mask_batched": imask,
            "p_did_list": torch.tensor([7, 8, 7, 9]) + 10 * step,
            "index_mapping": {"query": [[2 * i] for i in range(b)], "pos_cand": [[2 * i + 1] for i in range(b)]},
        }
        model.zero_grad()
        res = model(batch, alpha=alpha)
        res["loss"].backward()
        steps.append((res["loss"].item(), res["accuracy"].item()))
        out.update({f"s{step}_ids": ids.numpy(), f"s{step}_mask": mask.numpy(), f"s{step}_img": img.numpy(),
                    f"s{step}_tmask": tmask.numpy(), f"s{step}_imask": imask.numpy(),
                    f"s{step}_pdid": batch["p_did_list"].numpy(), f"s{step}_alpha": alpha,
                    f"s{step}_loss": res["loss"].item(), f"s{step}_acc": res["accuracy"].item(),
                    f"s{step}_dtemp": model.temp.grad.numpy(),
                    f"s{step}_g_vit_qkv0": model.visual_encoder.blocks[0].attn.qkv.weight.grad.numpy(),
                    f"s{step}_g_txt_q0": model.text_encoder.encoder.layer[0].attention.self.query.weight.grad.numpy(),
                    f"s{step}_g_vproj": model.vision_proj.weight.grad.numpy(), f"s{step}_g_tprojb": model.text_proj.bias.grad.numpy(),
                    f"s{step}_query_queue": model.query_queue.numpy().copy(), f"s{step}_cand_queue": model.cand_queue.numpy().copy(),
                    f"s{step}_idx_queue": model.idx_queue.numpy().copy(), f"s{step}_ptr": model.new_ptr_queue.numpy().copy(),
                    f"s{step}_m_vproj": model.vision_proj_m.weight.detach().numpy().copy()})
        with torch.no_grad():
            for p in model.parameters():
                if p.grad is not None:
                    p.add_(-0.05 * p.grad)
    assert model.text_encoder.encoder.layer[0].crossattention.self.query.weight.grad is None
    for k, v in sd0.items():
        if "_m." in k:
            continue
        if v.dtype.is_floating_point or k.endswith("idx_queue") or k.endswith("new_ptr_queue"):
            out[f"sd0::{k}"] = v.numpy()
    out.update(med_cfg=json.dumps(TINY_MED), vit_cfg=json.dumps(TINY_VIT), queue_size=K, momentum=0.9, embed_dim=E)
    np.savez_compressed(os.path.join(HERE, "g8s_blipsf.npz"), **out)
    print("g8s ok", steps)


if __name__ == "__main__":
    install_shims()
    torch.set_num_threads(8)
    for w in (sys.argv[1:] or ["g6", "g7", "g8"]):
        globals()[w]()
