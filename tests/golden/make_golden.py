#!/usr/bin/env python
"""Generates the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE (read-only, /root/reference/src)
in the build container.  The reference never travels: only the numeric inputs / outputs written here do.

Run:  python tests/golden/make_golden.py        (CPU only; needs /root/reference)

Import shims: stub modules for `clip` (so that UniIR's own CLIPScoreFusion imports and runs
with an encoder of our choosing), torchvision.transforms, typeguard, faiss, omegaconf (imports only).

Fixtures:
  g1_infonce_w1.npz   reference CLIPScoreFusion.compute_inbatch_contrastive_loss, W=1, identity encoder
  g2_infonce_w2.npz   same with gather_embeddings=True on 2 gloo ranks (per-rank loss / acc / score / d emb)
  g3_hardneg.npz      hard-negative branch (clip_sf.py:105-131)
  g4_masks.npz        encode_multimodal_input mask semantics with a tiny real encoder
  g5_hf_clip.npz      independent encoder stand-in: transformers.CLIPModel (tiny, seeded) weights + outputs
  g9_host.json        hash/unhash ids, ContiguousDistributedSampler partitions, compute_recall_at_k, format_string
  g10_train.npz       the reference's own engine.train_one_epoch (AdamW 2 groups + cosine LR) on a tiny encoder
  g11_embedder.npz    mbeir_embedder.generate_embeds_and_ids_for_dataset_with_gather (non-distributed branch)
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/src"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "common"))

from oracle import clip_oracle as O  # noqa: E402

# ------------------------------------------------------------------------------------------------- stubs
_STATE = {"model": None}


def _install_stubs():
    clip = types.ModuleType("clip")

    def load(name, device=None, jit=False, download_root=None):
        return _STATE["model"], (lambda img: img)

    def tokenize(txt, context_length=77, truncate=True):
        raise RuntimeError("not used")

    clip.load, clip.tokenize = load, tokenize
    sys.modules["clip"] = clip
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")

    class Resize:
        def __init__(self, *a, **k):
            pass

    tvt.Resize = Resize
    tv.transforms = tvt
    sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tvt
    tg = types.ModuleType("typeguard")
    tg.typechecked = lambda f: f
    sys.modules["typeguard"] = tg
    sys.modules["faiss"] = types.ModuleType("faiss")
    oc = types.ModuleType("omegaconf")
    oc.OmegaConf = type("OmegaConf", (), {})
    sys.modules["omegaconf"] = oc


class IdentityCLIP(torch.nn.Module):
    """encode_text / encode_image return their inputs: only UniIR-owned code is exercised."""

    def __init__(self, logit_scale):
        super().__init__()
        self.logit_scale = torch.nn.Parameter(torch.tensor(float(logit_scale)))

    def encode_text(self, t):
        return t

    def encode_image(self, i):
        return i


def _cfg(gather, in_batch_neg_num=0):
    return types.SimpleNamespace(model=types.SimpleNamespace(gather_embeddings=gather),
                                 data_config=types.SimpleNamespace(in_batch_neg_num=in_batch_neg_num))


def _ref_loss_run(txt, img, tmask, imask, index_mapping, logit_scale, gather, in_batch_neg_num=0):
    """runs the reference class; returns loss, acc, recorded score/targets, grads wrt txt/img/logit_scale"""
    from models.uniir_clip.clip_scorefusion.clip_sf import CLIPScoreFusion
    _STATE["model"] = IdentityCLIP(logit_scale)
    m = CLIPScoreFusion("stub", "cpu", config=_cfg(gather, in_batch_neg_num))
    rec = {}
    ce = m.loss_function

    class RecordingCE(torch.nn.Module):
        def forward(self, score, targets):
            rec["score"], rec["targets"] = score.detach().clone(), targets.clone()
            return ce(score, targets)

    m.loss_function = RecordingCE()
    txt = txt.clone().requires_grad_(True)
    img = img.clone().requires_grad_(True)
    batch = {"txt_batched": txt, "image_batched": img, "txt_mask_batched": tmask, "image_mask_batched": imask,
             "index_mapping": index_mapping}
    out = m(batch)
    out["loss"].backward()
    return dict(loss=out["loss"].detach(), acc=out["accuracy"].detach(), score=rec.get("score"),
                targets=rec.get("targets"), dtxt=txt.grad, dimg=img.grad, dscale=m.clip_model.logit_scale.grad)


def g1():
    out = {}
    for tag, (b, E) in {"a": (4, 8), "b": (32, 512)}.items():
        g = torch.Generator().manual_seed(100 + b)
        M = 2 * b
        txt, img = torch.randn(M, E, generator=g), torch.randn(M, E, generator=g)
        tmask = (torch.rand(M, generator=g) > 0.3).long()
        imask = (torch.rand(M, generator=g) > 0.3).long()
        tmask[(tmask + imask) == 0] = 1
        im = {"query": [[2 * i] for i in range(b)], "pos_cand": [[2 * i + 1] for i in range(b)]}
        r = _ref_loss_run(txt, img, tmask, imask, im, np.log(1 / 0.07), gather=False)
        for k, v in dict(txt=txt, img=img, tmask=tmask, imask=imask, **r).items():
            out[f"{tag}_{k}"] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "g1_infonce_w1.npz"), **out)
    print("g1 ok", float(out["a_loss"]), float(out["b_loss"]))


def _g2_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _install_stubs()
    b, E = 6, 16
    g = torch.Generator().manual_seed(500 + rank)
    M = 2 * b
    txt, img = torch.randn(M, E, generator=g), torch.randn(M, E, generator=g)
    tmask, imask = torch.ones(M, dtype=torch.long), torch.ones(M, dtype=torch.long)
    im = {"query": [[2 * i] for i in range(b)], "pos_cand": [[2 * i + 1] for i in range(b)]}
    r = _ref_loss_run(txt, img, tmask, imask, im, np.log(1 / 0.07), gather=True)
    q.put((rank, {k: v.numpy() for k, v in dict(txt=txt, img=img, **r).items()}))
    dist.barrier()
    dist.destroy_process_group()


def g2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    procs = [ctx.Process(target=_g2_worker, args=(r, world, 29511, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get() for _ in range(world))
    for p in procs:
        p.join()
    out = {}
    for r in range(world):
        for k, v in res[r].items():
            out[f"r{r}_{k}"] = v
    np.savez_compressed(os.path.join(HERE, "g2_infonce_w2.npz"), **out)
    print("g2 ok", float(out["r0_loss"]), float(out["r1_loss"]))


def g3():
    out = {}
    b, E, nneg = 4, 8, 2
    for tag, ibn in {"n0": 0, "n2": 2}.items():
        g = torch.Generator().manual_seed(300)
        M = b * (2 + nneg)
        txt, img = torch.randn(M, E, generator=g), torch.randn(M, E, generator=g)
        tmask, imask = torch.ones(M, dtype=torch.long), torch.ones(M, dtype=torch.long)
        im = {"query": [], "pos_cand": [], "neg_cand_list": []}
        c = 0
        for i in range(b):  # collator order: query, pos, negs (mbeir_dataset.py:483-498)
            im["query"].append([c]); c += 1
            im["pos_cand"].append([c]); c += 1
            im["neg_cand_list"].append(list(range(c, c + nneg))); c += nneg
        r = _ref_loss_run(txt, img, tmask, imask, im, np.log(1 / 0.07), gather=False, in_batch_neg_num=ibn)
        for k, v in dict(txt=txt, img=img, **{k: v for k, v in r.items() if v is not None}).items():
            out[f"{tag}_{k}"] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "g3_hardneg.npz"), **out)
    print("g3 ok", float(out["n0_loss"]), float(out["n2_loss"]))


def g4():
    from models.uniir_clip.clip_scorefusion.clip_sf import CLIPScoreFusion
    cfg = O.tiny_config()
    _STATE["model"] = O.OracleCLIP(cfg, seed=7)
    m = CLIPScoreFusion("stub", "cpu", config=_cfg(False))
    batch = O.synthetic_batch(cfg, 3, seed=21)
    batch["txt_mask_batched"] = torch.tensor([1, 1, 0, 1, 0, 1])
    batch["image_mask_batched"] = torch.tensor([1, 0, 1, 1, 1, 0])
    with torch.no_grad():
        emb = m.encode_multimodal_input(batch["txt_batched"], batch["image_batched"], batch["txt_mask_batched"],
                                        batch["image_mask_batched"])
        out = m(batch)
    np.savez_compressed(os.path.join(HERE, "g4_masks.npz"), emb=emb.numpy(), loss=out["loss"].numpy(),
                        acc=out["accuracy"].numpy(), tmask=batch["txt_mask_batched"].numpy(),
                        imask=batch["image_mask_batched"].numpy())
    print("g4 ok", float(out["loss"]))


def hf_to_openai(hf_sd, cfg):
    """key map transformers.CLIPModel -> upstream names."""
    sd = {}
    sd["visual.conv1.weight"] = hf_sd["vision_model.embeddings.patch_embedding.weight"]
    sd["visual.class_embedding"] = hf_sd["vision_model.embeddings.class_embedding"]
    sd["visual.positional_embedding"] = hf_sd["vision_model.embeddings.position_embedding.weight"]
    sd["visual.ln_pre.weight"] = hf_sd["vision_model.pre_layrnorm.weight"]
    sd["visual.ln_pre.bias"] = hf_sd["vision_model.pre_layrnorm.bias"]
    sd["visual.ln_post.weight"] = hf_sd["vision_model.post_layernorm.weight"]
    sd["visual.ln_post.bias"] = hf_sd["vision_model.post_layernorm.bias"]
    sd["visual.proj"] = hf_sd["visual_projection.weight"].t().contiguous()
    sd["token_embedding.weight"] = hf_sd["text_model.embeddings.token_embedding.weight"]
    sd["positional_embedding"] = hf_sd["text_model.embeddings.position_embedding.weight"]
    sd["ln_final.weight"] = hf_sd["text_model.final_layer_norm.weight"]
    sd["ln_final.bias"] = hf_sd["text_model.final_layer_norm.bias"]
    sd["text_projection"] = hf_sd["text_projection.weight"].t().contiguous()
    sd["logit_scale"] = hf_sd["logit_scale"]
    for hfp, op, L in (("vision_model.encoder.layers", "visual.transformer.resblocks", cfg["vision_layers"]),
                       ("text_model.encoder.layers", "transformer.resblocks", cfg["transformer_layers"])):
        for i in range(L):
            h, o = f"{hfp}.{i}", f"{op}.{i}"
            sd[f"{o}.attn.in_proj_weight"] = torch.cat([hf_sd[f"{h}.self_attn.{x}_proj.weight"] for x in "qkv"], 0)
            sd[f"{o}.attn.in_proj_bias"] = torch.cat([hf_sd[f"{h}.self_attn.{x}_proj.bias"] for x in "qkv"], 0)
            sd[f"{o}.attn.out_proj.weight"] = hf_sd[f"{h}.self_attn.out_proj.weight"]
            sd[f"{o}.attn.out_proj.bias"] = hf_sd[f"{h}.self_attn.out_proj.bias"]
            sd[f"{o}.ln_1.weight"], sd[f"{o}.ln_1.bias"] = hf_sd[f"{h}.layer_norm1.weight"], hf_sd[f"{h}.layer_norm1.bias"]
            sd[f"{o}.ln_2.weight"], sd[f"{o}.ln_2.bias"] = hf_sd[f"{h}.layer_norm2.weight"], hf_sd[f"{h}.layer_norm2.bias"]
            sd[f"{o}.mlp.c_fc.weight"], sd[f"{o}.mlp.c_fc.bias"] = hf_sd[f"{h}.mlp.fc1.weight"], hf_sd[f"{h}.mlp.fc1.bias"]
            sd[f"{o}.mlp.c_proj.weight"], sd[f"{o}.mlp.c_proj.bias"] = hf_sd[f"{h}.mlp.fc2.weight"], hf_sd[f"{h}.mlp.fc2.bias"]
    return {k: v.detach().clone().float() for k, v in sd.items()}


def g5():
    saved = {k: sys.modules.pop(k) for k in ("torchvision", "torchvision.transforms") if k in sys.modules}
    from transformers import CLIPConfig, CLIPModel  # must not see the torchvision import stub
    sys.modules.update(saved)
    cfg = O.tiny_config(vision_width=128, vision_layers=2, transformer_width=64, transformer_heads=1,
                        transformer_layers=2, embed_dim=64, image_resolution=64, vision_patch_size=16, vocab_size=512)
    hc = CLIPConfig(
        text_config=dict(vocab_size=cfg["vocab_size"], hidden_size=cfg["transformer_width"],
                         intermediate_size=4 * cfg["transformer_width"], num_hidden_layers=cfg["transformer_layers"],
                         num_attention_heads=cfg["transformer_heads"], max_position_embeddings=cfg["context_length"],
                         hidden_act="quick_gelu", layer_norm_eps=1e-5, eos_token_id=cfg["vocab_size"] - 1,
                         bos_token_id=cfg["vocab_size"] - 2, pad_token_id=0, projection_dim=cfg["embed_dim"]),
        vision_config=dict(hidden_size=cfg["vision_width"], intermediate_size=4 * cfg["vision_width"],
                           num_hidden_layers=cfg["vision_layers"], num_attention_heads=cfg["vision_width"] // 64,
                           image_size=cfg["image_resolution"], patch_size=cfg["vision_patch_size"],
                           hidden_act="quick_gelu", layer_norm_eps=1e-5, projection_dim=cfg["embed_dim"]),
        projection_dim=cfg["embed_dim"])
    torch.manual_seed(55)
    hf = CLIPModel(hc).eval().float()
    # make LayerNorm / bias parameters non-trivial so that the mapping is really tested
    with torch.no_grad():
        for n, p in hf.named_parameters():
            if p.ndim < 2:
                p.add_(0.05 * torch.randn_like(p))
    sd = hf_to_openai(hf.state_dict(), cfg)
    batch = O.synthetic_batch(cfg, 3, seed=77)
    txt, img = batch["txt_batched"].long(), batch["image_batched"]
    with torch.no_grad():
        amask = torch.ones_like(txt)
        t_hf = hf.get_text_features(input_ids=txt, attention_mask=amask)
        i_hf = hf.get_image_features(pixel_values=img)
        t_hf = getattr(t_hf, "pooler_output", t_hf)
        i_hf = getattr(i_hf, "pooler_output", i_hf)
        t_or = O.encode_text(sd, txt, cfg)
        i_or = O.encode_image(sd, img, cfg)
    print("g5 oracle-vs-HF max abs diff: text", float((t_hf - t_or).abs().max()), "image", float((i_hf - i_or).abs().max()))
    out = {f"sd::{k}": v.numpy() for k, v in sd.items()}
    out.update(txt=txt.numpy().astype(np.int32), img=img.numpy(), text_features=t_hf.numpy(), image_features=i_hf.numpy(),
               cfg=json.dumps(cfg))
    np.savez_compressed(os.path.join(HERE, "g5_hf_clip.npz"), **out)


def g9():
    from data.preprocessing.utils import hash_qid, unhash_qid, hash_did, unhash_did, format_string
    from dist_utils import ContiguousDistributedSampler
    import importlib
    sys.modules.pop("interactive_retriever", None)
    # compute_recall_at_k lives in mbeir_retriever.py whose imports need more stubs
    ir = types.ModuleType("interactive_retriever")
    ir.InteractiveRetriever = object
    sys.modules["interactive_retriever"] = ir
    mr = importlib.import_module("mbeir_retriever")
    out = {"hash": [], "sampler": [], "recall": [], "format": []}
    for qid in ["0:1", "3:77", "9:499999", "10:1"]:
        h = hash_qid(qid)
        out["hash"].append({"qid": qid, "hq": h, "uq": unhash_qid(h), "hd": hash_did(qid), "ud": unhash_did(hash_did(qid))})
    for n in [0, 1, 7, 8, 9, 17, 100]:
        for W in [1, 2, 8]:
            parts = [list(iter(ContiguousDistributedSampler(list(range(n)), num_replicas=W, rank=r))) for r in range(W)]
            out["sampler"].append({"n": n, "W": W, "parts": parts})
    cases = [(["a", "b"], ["x", "b", "y"], 1), (["a", "b"], ["x", "b", "y"], 2), ([], ["x"], 1), (["q"], ["q", "r"], 1),
             (["q"], ["r", "s", "q"], 2), (["q"], ["r", "s", "q"], 3)]
    for rel, ret, k in cases:
        out["recall"].append({"rel": rel, "ret": ret, "k": k, "v": mr.compute_recall_at_k(rel, ret, k)})
    for s in ["  hello world ", '"quoted text"', "", None, "Ends with period.", "question?", "a\r\nb"]:
        out["format"].append({"in": s, "out": format_string(s)})
    out["collator"] = _g9_collators()
    out["runfile"] = _g9_runfile()
    out["dataset"] = _g9_dataset()
    json.dump(out, open(os.path.join(HERE, "g9_host.json"), "w"), indent=1)
    print("g9 ok")


# ---- batch ABI: the reference's own collators / datasets on hand-made instances -------------------
from g9_helpers import (g9_tokenizer, g9_image, g9_materialise, g9_flatten, g9_write_tree, g9_img_fn,  # noqa: E402
                        g9_dataset_rows)


G9_CASES = {
    # train mode with hard negatives: text-only / image-only / both, empty string and None both count as "no text"
    "train_neg": {"collator": "main", "mode": "train", "batch": [
        {"query": {"txt": "q zero", "img": None}, "pos_cand": {"txt": "", "img": 0.5},
         "neg_cand_list": [{"txt": "n0", "img": None}, {"txt": None, "img": 0.25}], "p_did": 700001},
        {"query": {"txt": "what is this?", "img": 1.0}, "pos_cand": {"txt": "p one", "img": None},
         "neg_cand_list": [{"txt": "n2", "img": 2.0}, {"txt": "n3", "img": None}], "p_did": 900002},
        {"query": {"txt": None, "img": 3.0}, "pos_cand": {"txt": "p two", "img": 4.0},
         "neg_cand_list": [{"txt": "n4", "img": None}, {"txt": "n5", "img": None}], "p_did": 5}]},
    # train mode without neg_cand_list
    "train_plain": {"collator": "main", "mode": "train", "batch": [
        {"query": {"txt": "alpha", "img": None}, "pos_cand": {"txt": "beta", "img": 0.125}, "p_did": 11},
        {"query": {"txt": "", "img": 6.0}, "pos_cand": {"txt": "gamma", "img": None}, "p_did": 12}]},
    # eval mode: qid / task_id popped into lists, no pos_cand in the mapping even when the instance carries one
    "eval": {"collator": "main", "mode": "eval", "batch": [
        {"query": {"txt": "e0", "img": None}, "qid": 9000001, "task_id": 3},
        {"query": {"txt": "e1", "img": 7.0}, "pos_cand": {"txt": "ignored", "img": None}, "qid": 9000002, "task_id": 0},
        {"query": {"txt": None, "img": 8.0}, "qid": 9000003, "task_id": 8}]},
    "inference_only": {"collator": "inference", "batch": [
        {"query": {"txt": "i0", "img": None}, "qid": 1, "task_id": 2},
        {"query": {"txt": "", "img": 9.0}, "qid": 2},
        {"query": {"txt": "i2", "img": 10.0}, "task_id": 4}]},
    "cand_pool": {"collator": "pool", "batch": [
        {"txt": "c0", "img": None, "did": 123}, {"txt": "", "img": 0.75, "did": 124}, {"txt": "c2", "img": 1.5, "did": 125},
        {"txt": None, "img": 2.5, "modality": "image", "did": 126}]},
    "cand_pool_no_did": {"collator": "pool", "batch": [{"txt": "x", "img": None}, {"txt": "y", "img": 1.0}]},
}


def _g9_collators():
    import importlib
    sys.modules.pop("data.mbeir_dataset", None)
    md = importlib.import_module("data.mbeir_dataset")       # the reference's own module (typeguard stubbed to identity)
    assert md.__file__.startswith(REF), md.__file__
    res = {}
    for name, case in G9_CASES.items():
        kind = case["collator"]
        if kind == "main":
            col = md.MBEIRMainCollator(g9_tokenizer, (4, 4), mode=md.Mode.TRAIN if case["mode"] == "train" else md.Mode.EVAL)
        elif kind == "inference":
            col = md.MBEIRInferenceOnlyCollator(g9_tokenizer, (4, 4))
        else:
            col = md.MBEIRCandidatePoolCollator(g9_tokenizer, 4)
        out = col([g9_materialise(b) for b in case["batch"]])
        res[name] = {"case": case, "out": g9_flatten(out)}
    return res


def _g9_runfile():
    """the run-file line of mbeir_retriever.py:438-443: the reference's own f-string, lifted from its AST and evaluated on two
    hits (the surrounding run_retrieval needs FAISS + a whole M-BEIR tree)"""
    import ast
    from data.preprocessing.utils import unhash_qid, unhash_did, hash_qid, hash_did
    src = open(os.path.join(REF, "common", "mbeir_retriever.py")).read()
    node = next(n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") == "run_file_line")
    fmt = compile(ast.Expression(node.value), "<mbeir_retriever.run_file_line>", "eval")
    rows = []
    run_id = "mbeir_mscoco_task0_union_pool_test_k10"
    dists = np.array([[0.8731712, 0.5], [1.0, -0.0317]], dtype=np.float32)
    idx = np.array([[hash_did("9:42"), hash_did("9:7")], [hash_did("1:499999"), hash_did("0:1")]], dtype=np.int64)
    for qi, hq in enumerate([hash_qid("9:3"), hash_qid("2:100")]):
        qid = unhash_qid(hq)
        for rank, (hashed_doc_id, score) in enumerate(zip(idx[qi], dists[qi]), start=1):
            doc_id = unhash_did(hashed_doc_id)
            task_id = 3 if qi == 0 else 0
            line = eval(fmt, {}, dict(qid=qid, doc_id=doc_id, rank=rank, score=score, run_id=run_id, task_id=task_id))
            rows.append({"hq": int(hq), "hd": int(hashed_doc_id), "score_f32_bits": int(np.float32(score).view(np.int32)),
                         "rank": rank, "run_id": run_id, "task_id": task_id, "line": line})
    return rows


G9_TREE = {
    "instructions": "query_modality\tcand_modality\tdataset_name\tdataset_id\tprompt_1\tprompt_2\n"
                    "text\timage\tmscoco\t9\tFind an image for the caption.\t\n"
                    "image\ttext\tmscoco\t9\tDescribe the picture\t\n"
                    "image,text\timage\tcirr\t7\tFind a similar image, modified as told.\t\n",
    "cand_pool": [
        {"did": "9:1", "txt": None, "img_path": "img/a.png", "modality": "image", "src_content": None},
        {"did": "9:2", "txt": "  a dog on grass ", "img_path": None, "modality": "text", "src_content": None},
        {"did": "9:3", "txt": '"a cat"', "img_path": None, "modality": "text", "src_content": None},
        {"did": "7:1", "txt": None, "img_path": "img/b.png", "modality": "image", "src_content": None},
        {"did": "7:2", "txt": None, "img_path": "img/a.png", "modality": "image", "src_content": None},
    ],
    "queries": [
        {"qid": "9:1", "query_txt": "a photo of a dog", "query_img_path": None, "query_modality": "text",
         "query_src_content": None, "pos_cand_list": ["9:1"], "neg_cand_list": ["7:2"], "task_id": 0},
        {"qid": "9:2", "query_txt": None, "query_img_path": "img/a.png", "query_modality": "image",
         "query_src_content": None, "pos_cand_list": ["9:2", "9:3"], "neg_cand_list": ["9:3"], "task_id": 3},
        {"qid": "7:1", "query_txt": "make it red", "query_img_path": "img/b.png", "query_modality": "image,text",
         "query_src_content": None, "pos_cand_list": ["7:1"], "neg_cand_list": ["7:2", "7:1"], "task_id": 7},
    ],
    "images": {"img/a.png": [10, 200, 30], "img/b.png": [250, 0, 120]},
}


def _g9_dataset():
    import importlib
    import tempfile
    md = importlib.import_module("data.mbeir_dataset")
    assert md.__file__.startswith(REF), md.__file__
    with tempfile.TemporaryDirectory() as root:
        g9_write_tree(root, G9_TREE)
        return {"tree": G9_TREE, "rows": g9_dataset_rows(md, root)}


def g10():
    from models.uniir_clip import engine
    from models.uniir_clip.clip_scorefusion.clip_sf import CLIPScoreFusion
    from torch import optim
    from torch.optim.lr_scheduler import CosineAnnealingLR
    from torch.cuda.amp import GradScaler
    cfg = O.tiny_config()
    sd0 = O.init_state_dict(cfg, seed=9)
    _STATE["model"] = O.OracleCLIP(cfg, sd0)
    model = CLIPScoreFusion("stub", "cpu", config=_cfg(False))
    model.float()
    # train.py:195-199 (filter_parameters / create_optimizer), lr raised so 4 steps move the weights visibly
    exclude = lambda n, p: p.ndim < 2 or any(s in n for s in ["bn", "ln", "bias", "logit_scale"])
    gain = [p for n, p in model.named_parameters() if exclude(n, p) and p.requires_grad]
    rest = [p for n, p in model.named_parameters() if not exclude(n, p) and p.requires_grad]
    lr = 1e-3
    opt = optim.AdamW([{"params": gain, "weight_decay": 0.0}, {"params": rest, "weight_decay": 0.2}], lr=lr,
                      betas=(0.9, 0.98), eps=1.0e-6)
    accum = 2
    batches = [O.synthetic_batch(cfg, 4, seed=40 + i) for i in range(4)]
    t_total = len(batches) // accum * 3
    sched = CosineAnnealingLR(opt, T_max=t_total, eta_min=0)
    config = types.SimpleNamespace(trainer_config=types.SimpleNamespace(print_freq=100, gradient_accumulation_steps=accum))
    losses, accs, lrs = [], [], []
    orig_update = engine.utils.MetricLogger.update

    def rec_update(self, **kw):
        if "loss" in kw: losses.append(kw["loss"])
        if "lr" in kw: lrs.append(kw["lr"])
        if "inbatch_accuracy" in kw: accs.append(kw["inbatch_accuracy"])
        return orig_update(self, **kw)

    engine.utils.MetricLogger.update = rec_update
    import copy
    stats = engine.train_one_epoch(model, [copy.deepcopy(b) for b in batches], opt, 0, "cpu", sched, 0, GradScaler(), config)
    engine.utils.MetricLogger.update = orig_update
    out = {f"sd0::{k}": v.numpy() for k, v in sd0.items()}
    final = {k: getattr(model.clip_model, k.replace(".", "__")).detach().numpy() for k in sd0}
    out.update({f"sd1::{k}": v for k, v in final.items()})
    out.update(losses=np.array(losses), accs=np.array(accs), lrs=np.array(lrs), cfg=json.dumps(cfg), lr=lr, accum=accum,
               t_total=t_total, n_nodecay=len(gain), n_decay=len(rest))
    np.savez_compressed(os.path.join(HERE, "g10_train.npz"), **out)
    print("g10 ok", losses, lrs)


def g11():
    tr = types.ModuleType("tqdm"); tr.tqdm = lambda x, **k: x
    sys.modules.setdefault("tqdm", tr)
    import importlib
    for name in ("utils",):
        sys.modules.pop(name, None)
    md = types.ModuleType("data.mbeir_dataset")
    for n in ["MBEIRMainDataset", "MBEIRMainCollator", "MBEIRCandidatePoolDataset", "MBEIRCandidatePoolCollator", "Mode"]:
        setattr(md, n, object)
    sys.modules["data.mbeir_dataset"] = md
    cu = types.ModuleType("utils"); cu.build_model_from_config = None; cu.set_seed = None
    sys.modules["utils"] = cu
    emb = importlib.import_module("mbeir_embedder")
    from models.uniir_clip.clip_scorefusion.clip_sf import CLIPScoreFusion
    cfg = O.tiny_config()
    _STATE["model"] = O.OracleCLIP(cfg, seed=12)
    model = CLIPScoreFusion("stub", "cpu", config=_cfg(False)).eval()
    batches = []
    for i in range(2):
        b = O.synthetic_batch(cfg, 2 + i, seed=60 + i)
        del b["index_mapping"]
        b["did_list"] = [1000 * (i + 1) + j for j in range(b["txt_batched"].shape[0])]
        batches.append(b)
    arr, ids = emb.generate_embeds_and_ids_for_dataset_with_gather(model, batches, "cpu", use_fp16=True)
    np.savez_compressed(os.path.join(HERE, "g11_embedder.npz"), emb=arr, ids=np.array(ids, dtype=np.int64))
    print("g11 ok", arr.shape, arr.dtype)


if __name__ == "__main__":
    _install_stubs()
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g9", "g10", "g11"]
    for w in which:
        globals()[w]()
