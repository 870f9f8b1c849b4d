"""End-to-end drop-in check on a tiny synthetic M-BEIR tree (authored here, format per SURVEY.md section 5.6):
train.py main() -> checkpoint -> mbeir_embedder main() -> create_index -> run_retrieval, all through the host mirrors
under uniir_amd/src and the HIP path; retrieval output is compared with the C oracle on the saved embeddings."""
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "uniir_amd", "src")
for p in (ROOT, SRC, os.path.join(SRC, "common")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _toy_tokenize(texts, context_length=77, truncate=True):
    out = torch.zeros(len(texts), context_length, dtype=torch.int32)
    for i, t in enumerate(texts):
        ids = [510] + [1 + (sum(map(ord, w)) % 500) for w in t.split()][: context_length - 2] + [511]
        out[i, : len(ids)] = torch.tensor(ids, dtype=torch.int32)
    return out


def _make_tree(root, n_cand=24, n_query=12):
    from PIL import Image
    rng = np.random.default_rng(0)
    os.makedirs(os.path.join(root, "img"), exist_ok=True)
    for sub in ("train", "val", "cand_pool", "instructions", "qrels/val"):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    words = ["red", "blue", "dog", "cat", "tree", "car", "river", "house", "bird", "stone", "cloud", "road"]
    cands = []
    for i in range(n_cand):
        rec = {"did": f"9:{i + 1}", "modality": "image,text" if i % 3 == 0 else ("image" if i % 3 == 1 else "text"),
               "txt": None, "img_path": None}
        if i % 3 != 1:
            rec["txt"] = " ".join(rng.choice(words, 4))
        if i % 3 != 2:
            path = f"img/c{i}.png"
            Image.fromarray(rng.integers(0, 255, (40, 52, 3), dtype=np.uint8)).save(os.path.join(root, path))
            rec["img_path"] = path
        cands.append(rec)
    with open(os.path.join(root, "cand_pool", "mbeir_toy_cand_pool.jsonl"), "w") as f:
        for c in cands:
            f.write(json.dumps(c) + "\n")
    qrels = []
    for split in ("train", "val"):
        with open(os.path.join(root, split, f"mbeir_toy_{split}.jsonl"), "w") as f:
            for i in range(n_query):
                pos = cands[(2 * i) % n_cand]
                q = {"qid": f"9:{i + 1}", "query_txt": " ".join(rng.choice(words, 3)), "query_img_path": None,
                     "query_modality": "text", "pos_cand_list": [pos["did"]], "neg_cand_list": []}
                f.write(json.dumps(q) + "\n")
                if split == "val":
                    task = {"image": 0, "text": 1, "image,text": 2}[pos["modality"]]
                    qrels.append(f"{q['qid']} 0 {pos['did']} 1 {task}")
    with open(os.path.join(root, "qrels", "val", "mbeir_toy_val_qrels.txt"), "w") as f:
        f.write("\n".join(qrels) + "\n")
    with open(os.path.join(root, "instructions", "query_instructions.tsv"), "w") as f:
        f.write("query_modality\tcand_modality\tdataset_name\tdataset_id\tprompt_1\tprompt_2\n")
        for cm in ("image", "text", "image,text"):
            f.write(f"text\t{cm}\tToy\t9\tfind a matching {cm.replace(',', ' and ')}\tretrieve the item\n")


def _configs(tmp, uniir_dir):
    common = f"""
experiment: {{instruct_status: "Instruct", exp_name: "InBatch", description: "toy", path_suffix: "${{model.short_name}}/${{experiment.instruct_status}}/"}}
model:
  name: "CLIPScoreFusion"
  short_name: "CLIP_SF"
  size: "Tiny"
  clip_vision_model_name: "tiny-test"
  pretrained_clip_model_dir: "checkpoint/CLIP/"
  gather_embeddings: True
  ckpt_config: {{ckpt_dir: "checkpoint/toy/", resume_training: False, ckpt_name: "clip_sf_epoch_0.pth"}}
seed: 2023
dist_config: {{dist_url: "env://"}}
"""
    train = common + """
logger_config: {logger_out_dir: "logger/toy", logger_out_file_name: "train.log"}
data_config:
  image_size: 64, 64
  hard_neg_num: 0
  in_batch_neg_num: 0
  shuffle_cand: True
  returns: null
  enable_query_instruct: True
  query_instruct_path: instructions/query_instructions.tsv
  train_query_data_path: train/mbeir_toy_train.jsonl
  train_cand_pool_path: cand_pool/mbeir_toy_cand_pool.jsonl
  val_query_data_path: val/mbeir_toy_val.jsonl
  val_cand_pool_path: cand_pool/mbeir_toy_cand_pool.jsonl
dataloader_config: {num_workers: 0, train_batch_size: 4, valid_batch_size: 4}
trainer_config: {gradient_accumulation_steps: 1, num_train_epochs: 1, learning_rate: 1e-4, warmup_steps: 0, eval_steps: 1, print_freq: 1}
evaluator: {enable_eval: True, eval_freq: 1, print_freq: 1}
"""
    embed = common + """
embed_config:
  embed_dir_name: "embed"
  use_fp16: True
  val_datasets_config: {enable_embed: True, datasets_name: ["toy"], correspond_cand_pools_name: ["toy"]}
  cand_pools_config: {enable_embed: True, embed_union_pool: False, cand_pools_name_to_embed: ["toy"]}
dataloader_config: {num_workers: 0, batch_size: 5}
data_config:
  image_size: 64, 64
  shuffle_cand: False
  enable_query_instruct: True
  train_dir_name: "train"
  val_dir_name: "val"
  test_dir_name: "test"
  cand_pool_dir_name: "cand_pool"
  query_instruct_path: instructions/query_instructions.tsv
"""
    index = common + """
index_config:
  faiss_config: {idx_type: Flat, dim: 64, metric: METRIC_INNER_PRODUCT}
  embed_dir_name: "embed"
  index_dir_name: "index"
  cand_pools_config: {enable_idx: True, cand_pools_name_to_idx: ["toy"]}
"""
    retrieval = common + """
retrieval_config:
  embed_dir_name: "embed"
  index_dir_name: "index"
  results_dir_name: "retrieval_results"
  qrel_dir_name: "qrels"
  write_to_tsv: True
  raw_retrieval: False
  val_datasets_config:
    enable_retrieve: True
    datasets_name: ["toy"]
    correspond_cand_pools_name: ["toy"]
    correspond_qrels_name: ["toy"]
    correspond_metrics_name: ["Recall@1, Recall@5, Recall@10"]
"""
    paths = {}
    for name, txt in dict(train=train, embed=embed, index=index, retrieval=retrieval).items():
        paths[name] = os.path.join(tmp, f"{name}.yaml")
        with open(paths[name], "w") as f:
            f.write(txt)
    return paths


def test_train_embed_index_retrieve_pipeline(tmp_path):
    from oracle import c_oracle
    from oracle import clip_oracle as O
    from uniir_amd import clip_front, clip_model
    from config import OmegaConf
    clip_model.CLIP_CONFIGS["tiny-test"] = O.tiny_config()
    clip_front.tokenize = _toy_tokenize
    data_dir, uniir_dir = str(tmp_path / "mbeir"), str(tmp_path / "uniir")
    _make_tree(data_dir)
    cfgs = _configs(str(tmp_path), uniir_dir)
    # the "pretrained" weights where run_inbatch.sh's config points (pretrained_clip_model_dir): a missing file is an error
    os.makedirs(os.path.join(uniir_dir, "checkpoint/CLIP"), exist_ok=True)
    torch.save({"state_dict": O.init_state_dict(O.tiny_config(), seed=4)}, os.path.join(uniir_dir, "checkpoint/CLIP/tiny-test.pt"))

    def load(name):
        c = OmegaConf.load(cfgs[name])
        c.uniir_dir, c.mbeir_data_dir = uniir_dir, data_dir
        c.dist_config.gpu_id, c.dist_config.distributed_mode = 0, False
        return c

    from models.uniir_clip.clip_scorefusion import train as train_mod
    train_mod.main(load("train"))
    ckpt = os.path.join(uniir_dir, "checkpoint/toy/clip_sf_epoch_0.pth")
    sd = torch.load(ckpt, map_location="cpu")
    assert set(sd) >= {"model", "optimizer", "scheduler", "config", "epoch", "scaler"}
    assert "clip_model.visual.transformer.resblocks.0.attn.in_proj_weight" in sd["model"]

    import mbeir_embedder
    import mbeir_retriever
    mbeir_embedder.main(load("embed"))
    base = os.path.join(uniir_dir, "embed", "CLIP_SF/Instruct")
    cemb = np.load(os.path.join(base, "cand_pool", "mbeir_toy_cand_pool_embed.npy"))
    cids = np.load(os.path.join(base, "cand_pool", "mbeir_toy_cand_pool_ids.npy"))
    qemb = np.load(os.path.join(base, "val", "mbeir_toy_val_embed.npy"))
    assert cemb.dtype == np.float16 and cemb.shape == (24, 64) and qemb.shape == (12, 64)
    assert cids.tolist() == [9 * 10_000_000 + i + 1 for i in range(24)]
    mbeir_retriever.create_index(load("index"))
    results = mbeir_retriever.run_retrieval(load("retrieval"), None)
    assert results and all(0.0 <= r["Recall@10"] <= 1.0 for r in results)
    run = os.path.join(uniir_dir, "retrieval_results", "CLIP_SF/Instruct", "run_files", "mbeir_toy_single_pool_val_k10_run.txt")
    lines = open(run).read().strip().split("\n")
    assert len(lines) == 12 * 10 and lines[0].split()[1] == "Q0"
    # retrieved ids == the C oracle's exact top-10 on the saved embeddings
    want_s, want_i = c_oracle.topk(cemb, cids, qemb, 10)
    got = np.array([[9 * 10_000_000 + int(l.split()[2].split(":")[1]) for l in lines[q * 10:(q + 1) * 10]] for q in range(12)])
    assert np.array_equal(got, want_i)
    # ---- hard-negative mining (reference mbeir_retriever.py:606-708) on the same tree: the train queries are the val ones
    hdir = os.path.join(base, "train")
    os.makedirs(hdir, exist_ok=True)
    np.save(os.path.join(hdir, "mbeir_toy_train_embed.npy"), qemb)
    np.save(os.path.join(hdir, "mbeir_toy_train_ids.npy"), np.array([9 * 500_000 + i + 1 for i in range(12)], dtype=np.int64))
    hcfg = load("retrieval")
    hcfg.retrieval_config.train_datasets_config = OmegaConf.create(
        {"enable_retrieve": True, "datasets_name": ["toy"], "correspond_cand_pools_name": ["toy"]})
    hcfg.retrieval_config.k, hcfg.retrieval_config.num_hard_negs = 6, 8
    hcfg.retrieval_config.hard_negs_dir_name = "hard_negs"
    mbeir_retriever.run_hard_negative_mining(hcfg)
    mined = [json.loads(l) for l in open(os.path.join(data_dir, "train", "hard_negs", "mbeir_toy_hard_negs_train.jsonl"))]
    src = [json.loads(l) for l in open(os.path.join(data_dir, "train", "mbeir_toy_train.jsonl"))]
    _, top6 = c_oracle.topk(cemb, cids, qemb, 6)
    assert len(mined) == 12
    for q in range(12):
        ranked = [f"9:{int(h) % 10_000_000}" for h in top6[q]]
        hard = [d for d in ranked if d not in src[q]["pos_cand_list"]]
        want = (hard * 8)[:8] if hard else []          # fewer than num_hard_negs: repeated cyclically, then cut
        assert mined[q]["qid"] == src[q]["qid"] and mined[q]["pos_cand_list"] == src[q]["pos_cand_list"]
        assert mined[q]["neg_cand_list"] == want, (q, mined[q]["neg_cand_list"], want)


def test_device_prefetcher_same_batches_same_order():
    from uniir_amd.host_utils import DevicePrefetcher

    class Enc(dict):                      # stands in for transformers' BatchEncoding (has .input_ids and .items())
        @property
        def input_ids(self):
            return self["input_ids"]

    g = torch.Generator().manual_seed(0)
    batches = [{"image_batched": torch.randn(4, 3, 8, 8, generator=g), "txt_batched": torch.randint(0, 99, (4, 7), generator=g),
                "enc": Enc(input_ids=torch.randint(0, 9, (4, 5), generator=g), attention_mask=torch.ones(4, 5, dtype=torch.long)),
                "index_mapping": {"query": [[0]]}, "n": i} for i in range(5)]
    want = [{k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in b.items()} for b in batches]
    feed = DevicePrefetcher(batches, 0)
    assert len(feed) == 5
    seen = 0
    for i, b in enumerate(feed):
        assert b["n"] == i and b["image_batched"].is_cuda and b["enc"]["input_ids"].is_cuda
        acc = b["image_batched"].double().sum() + b["txt_batched"].sum()        # consume on the compute stream
        assert torch.equal(b["image_batched"].cpu(), want[i]["image_batched"])
        assert torch.equal(b["txt_batched"].cpu(), want[i]["txt_batched"])
        assert abs(acc.item() - (want[i]["image_batched"].double().sum() + want[i]["txt_batched"].sum()).item()) < 1e-6
        seen += 1
    assert seen == 5 and list(DevicePrefetcher([], 0)) == []


def test_deferred_device_image_transform_through_dataset_collator_prefetcher(tmp_path):
    """img_preprocess_fn = RawImageTransform: the workers only decode, the collator keeps uint8 images, the prefetcher
    transforms them on the GPU -- the batch tensor is bit-identical to the CPU (Pillow + torch) transform, image-less items
    stay black"""
    from PIL import Image
    from data.mbeir_dataset import MBEIRCandidatePoolCollator, MBEIRCandidatePoolDataset
    from uniir_amd import clip_front
    from uniir_amd.host_utils import DevicePrefetcher
    root = str(tmp_path)
    _make_tree(root, n_cand=9, n_query=3)
    pool = os.path.join("cand_pool", "mbeir_toy_cand_pool.jsonl")
    n_px = 32
    batches = {}
    for mode, fn in (("cpu", clip_front._preprocess(n_px)), ("gpu", clip_front.RawImageTransform(n_px))):
        ds = MBEIRCandidatePoolDataset(root, pool, fn, print_config=False)
        col = MBEIRCandidatePoolCollator(tokenizer=_toy_tokenize, image_size=(n_px, n_px))
        if mode == "gpu":
            col.raw_transform = fn
        batch = col([ds[i] for i in range(len(ds))])
        batches[mode] = next(iter(DevicePrefetcher([batch], 0)))
    a, b = batches["cpu"]["image_batched"], batches["gpu"]["image_batched"]
    assert a.is_cuda and b.is_cuda and a.shape == b.shape == (9, 3, n_px, n_px)
    assert torch.equal(batches["cpu"]["image_mask_batched"], batches["gpu"]["image_mask_batched"])
    assert torch.equal(a, b)
    assert (b[batches["gpu"]["image_mask_batched"] == 0] == 0).all() and batches["gpu"]["image_mask_batched"].sum().item() == 6


def test_retriever_shard_loop_with_several_shards_equals_one_shard_and_the_oracle(tmp_path, monkeypatch):
    """mbeir_retriever.search_index over an index file split into 3 row shards (reference mbeir_retriever.py:98-100, 204-206:
    the index spread over all visible GPUs) -- on this one-GPU box the three shards share the device (UNIIR_RETRIEVER_SHARDS=3),
    which still runs the per-shard searches, the stacking on the first shard's device and the k-way merge: distances and ids
    identical to the single-shard search and to the C oracle over the whole pool"""
    import numpy as np
    from common import mbeir_retriever
    from oracle import c_oracle
    rng = np.random.default_rng(5)
    n, d, nq, k = 5_003, 64, 37, 10
    emb = rng.standard_normal((n, d)).astype(np.float16)
    emb[4000] = emb[7]                                   # a tie across shards: broken by id
    ids = (rng.permutation(n).astype(np.int64) * 3 + 5)
    q = rng.standard_normal((nq, d)).astype(np.float16)
    q[3] = emb[7]
    index_path, q_path = str(tmp_path / "pool.index"), str(tmp_path / "q.npy")
    with open(index_path, "wb") as f:
        np.savez(f, emb=emb, ids=ids)
    np.save(q_path, q)
    want_s, want_i = c_oracle.topk(emb, ids, q, k)
    res = {}
    for shards in ("1", "3"):
        monkeypatch.setenv("UNIIR_RETRIEVER_SHARDS", shards)
        mbeir_retriever._SHARD_CACHE.clear()
        res[shards] = mbeir_retriever.search_index(q_path, index_path, batch_size=16, num_cand_to_retrieve=k)
        assert len(mbeir_retriever._device_shards(index_path)) == int(shards)
    mbeir_retriever._SHARD_CACHE.clear()
    for shards in ("1", "3"):
        assert np.array_equal(res[shards][1], want_i), shards
        assert np.array_equal(res[shards][0], want_s), shards
