"""CPU tests of the host-side mirrors (ids, sampler, recall, string formatting, config loader, collator, config
updater) against the golden values captured from the reference (tests/golden/g9_host.json) and hand-made cases."""
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "uniir_amd", "src")
for p in (SRC, os.path.join(SRC, "common")):
    if p not in sys.path:
        sys.path.insert(0, p)

G9 = json.load(open(os.path.join(ROOT, "tests", "golden", "g9_host.json")))


def test_id_hashing_matches_reference():
    from data.preprocessing.utils import hash_did, hash_qid, unhash_did, unhash_qid
    for c in G9["hash"]:
        assert hash_qid(c["qid"]) == c["hq"] and unhash_qid(c["hq"]) == c["uq"]
        assert hash_did(c["qid"]) == c["hd"] and unhash_did(c["hd"]) == c["ud"]


def test_contiguous_sampler_matches_reference_partitions():
    from dist_utils import ContiguousDistributedSampler
    from uniir_amd import comm
    for c in G9["sampler"]:
        n, W = c["n"], c["W"]
        for r in range(W):
            got = list(iter(ContiguousDistributedSampler(list(range(n)), num_replicas=W, rank=r)))
            assert got == c["parts"][r], (n, W, r)
            lo, hi = comm.contiguous_shard(n, W, r)
            assert list(range(lo, hi)) == c["parts"][r]


def test_recall_and_format_string_match_reference():
    from data.preprocessing.utils import format_string
    from mbeir_retriever import compute_recall_at_k
    for c in G9["recall"]:
        assert compute_recall_at_k(c["rel"], c["ret"], c["k"]) == c["v"]
    for c in G9["format"]:
        assert format_string(c["in"]) == c["out"]


def test_task_tables():
    from data.preprocessing.utils import MBEIR_TASK, get_mbeir_task_id, get_mbeir_task_name
    assert get_mbeir_task_id("image,text", "image") == 7 and get_mbeir_task_name(3) == "image -> text"
    assert sorted(MBEIR_TASK.values()) == list(range(9))


def test_config_loader_interpolation(tmp_path):
    from config import OmegaConf
    p = tmp_path / "c.yaml"
    p.write_text('experiment:\n  instruct_status: "Instruct"\n  exp_name: "InBatch"\n'
                 '  path_suffix: "${model.short_name}/${experiment.instruct_status}/${experiment.exp_name}/"\n'
                 'model:\n  short_name: "CLIP_SF"\n  size: "Large"\n  dup: ${model.size}\nseed: 2023\n'
                 'data_config:\n  image_size: 224, 224\n  enable_query_instruct: True\n')
    c = OmegaConf.load(str(p))
    assert c.experiment.path_suffix == "CLIP_SF/Instruct/InBatch/" and c.model.dup == "Large" and c.seed == 2023
    c.uniir_dir = "/x"
    c.dist_config = {"gpu_id": 0}
    assert c.dist_config.gpu_id == 0 and "uniir_dir" in OmegaConf.to_yaml(c)
    from config_updater import update_mbeir_yaml_instruct_status
    update_mbeir_yaml_instruct_status(str(p), False)
    c2 = OmegaConf.load(str(p))
    assert c2.experiment.instruct_status == "NoInstruct" and c2.data_config.enable_query_instruct is False
    assert c2.experiment.path_suffix == "CLIP_SF/NoInstruct/InBatch/"


def test_main_collator_batch_abi():
    """flat interleaved order [query, pos_cand(, negs)] per instance, index_mapping, masks, black image / empty text
    padding (mbeir_dataset.py:427-434,483-498 of the reference)."""
    from data.mbeir_dataset import MBEIRCandidatePoolCollator, MBEIRMainCollator, Mode
    tok = lambda txts: torch.tensor([[len(t)] + [0] * 76 for t in txts], dtype=torch.int32)
    img = torch.ones(3, 8, 8)
    batch = [
        {"query": {"txt": "q0", "img": None}, "pos_cand": {"txt": "", "img": img},
         "neg_cand_list": [{"txt": "n0", "img": None}, {"txt": "n1", "img": img}], "p_did": 77},
        {"query": {"txt": "q1", "img": img}, "pos_cand": {"txt": "p1", "img": None},
         "neg_cand_list": [{"txt": "n2", "img": None}, {"txt": "n3", "img": None}], "p_did": 78},
    ]
    out = MBEIRMainCollator(tok, (8, 8), mode=Mode.TRAIN)(batch)
    assert out["index_mapping"] == {"query": [[0], [4]], "pos_cand": [[1], [5]], "neg_cand_list": [[2, 3], [6, 7]]}
    assert out["txt_mask_batched"].tolist() == [1, 0, 1, 1, 1, 1, 1, 1]
    assert out["image_mask_batched"].tolist() == [0, 1, 0, 1, 1, 0, 0, 0]
    assert out["image_batched"].shape == (8, 3, 8, 8) and out["image_batched"][0].abs().sum() == 0
    assert out["txt_batched"][1, 0] == 0 and out["p_did_list"].tolist() == [77, 78]
    pool = MBEIRCandidatePoolCollator(tok, 8)([{"txt": "a", "img": None, "did": 5}, {"txt": "", "img": img, "did": 6}])
    assert pool["did_list"] == [5, 6] and pool["txt_mask_batched"].tolist() == [1, 0]


def _g9():
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import g9_helpers
    return g9_helpers


def test_collators_match_the_reference_fixture_key_for_key():
    """every collator of the mirror on the instances of g9_host.json["collator"], whose expected outputs were
    produced by the REFERENCE's MBEIRMainCollator / MBEIRInferenceOnlyCollator / MBEIRCandidatePoolCollator
    (mbeir_dataset.py:440-526,529-570,572-610; tests/golden/make_golden.py g9): index_mapping counter, masks, padding rows,
    qid / task_id / p_did / did lists, key sets and dtypes."""
    from data.mbeir_dataset import MBEIRCandidatePoolCollator, MBEIRInferenceOnlyCollator, MBEIRMainCollator, Mode
    H = _g9()
    assert set(G9["collator"]) >= {"train_neg", "train_plain", "eval", "inference_only", "cand_pool", "cand_pool_no_did"}
    for name, rec in G9["collator"].items():
        case, want = rec["case"], rec["out"]
        if case["collator"] == "main":
            col = MBEIRMainCollator(H.g9_tokenizer, (4, 4), mode=Mode.TRAIN if case["mode"] == "train" else Mode.EVAL)
        elif case["collator"] == "inference":
            col = MBEIRInferenceOnlyCollator(H.g9_tokenizer, (4, 4))
        else:
            col = MBEIRCandidatePoolCollator(H.g9_tokenizer, 4)
        got = H.g9_flatten(col([H.g9_materialise(b) for b in case["batch"]]))
        assert set(got) == set(want), (name, sorted(got), sorted(want))
        for k in want:
            assert got[k] == want[k], (name, k, got[k], want[k])


def test_run_file_line_matches_the_reference_format():
    """the run-file line (mbeir_retriever.py:438-443): the mirror's writer on the fixture's hits == the reference's own f-string
    evaluated on them (float32 scores print with numpy's repr digits)"""
    import numpy as np
    from data.preprocessing.utils import unhash_did, unhash_qid
    import mbeir_retriever as mr
    for r in G9["runfile"]:
        score = np.array([r["score_f32_bits"]], dtype=np.int32).view(np.float32)[0]
        line = mr.run_file_line(unhash_qid(r["hq"]), unhash_did(r["hd"]), r["rank"], score, r["run_id"], r["task_id"])
        assert line == r["line"], (line, r["line"])


def test_datasets_match_the_reference_fixture(tmp_path):
    """MBEIRMainDataset (train with 2 hard negatives / train without instructions / eval) and MBEIRCandidatePoolDataset
    __getitem__ on a 3-query M-BEIR tree == the reference's (mbeir_dataset.py:179-277,376-411): prompt + query text through
    format_string, first positive, wrapped hard negatives, hashed ids, task ids, image hand-over to img_preprocess_fn."""
    import data.mbeir_dataset as md
    H = _g9()
    H.g9_write_tree(str(tmp_path), G9["dataset"]["tree"])
    got = H.g9_dataset_rows(md, str(tmp_path))
    want = G9["dataset"]["rows"]
    assert set(got) == set(want)
    for tag in want:
        assert len(got[tag]) == len(want[tag]), tag
        for i, (g, w) in enumerate(zip(got[tag], want[tag])):
            assert g == w, (tag, i, g, w)


def test_wgrad_split_heuristic_fills_the_chip():
    from uniir_amd.ops import wgrad_splits
    for tiles in (12, 16, 48, 64):
        s = wgrad_splits(263168, tiles)
        assert 0.9 * 256 <= tiles * s <= 256, (tiles, s)


def test_blip_ff_state_dict_keys_match_reference_checkpoint_layout():
    """the golden G8 holds the reference module's own state-dict keys; ours must be a superset with equal shapes"""
    import json as _json
    import numpy as np
    from models.uniir_blip.blip_featurefusion.blip_ff import BLIPFeatureFusion
    z = np.load(os.path.join(ROOT, "tests", "golden", "g8_blipff.npz"))
    med_cfg, vit_cfg = _json.loads(str(z["med_cfg"])), _json.loads(str(z["vit_cfg"]))
    m = BLIPFeatureFusion(med_config=med_cfg, vit_config=vit_cfg, embed_dim=med_cfg["hidden_size"],
                          queue_size=int(z["queue_size"]), momentum=float(z["momentum"]))
    own = m.state_dict()
    ref = {k[5:]: z[k].shape for k in z.files if k.startswith("sd0::")}
    for k, shp in ref.items():
        assert k in own and tuple(own[k].shape) == tuple(shp), k
        if k.startswith(("visual_encoder.", "text_encoder.")):
            km = k.replace("_encoder.", "_encoder_m.", 1)
            assert km in own and not m.get_parameter(km).requires_grad
    assert own["idx_queue"].dtype == torch.int64 and int(own["idx_queue"][0, 0]) == -100
    # full-size geometry from the reference's relative med_config path: BERT-base + ViT-B/16 parameter count
    full = BLIPFeatureFusion(med_config="../models/uniir_blip/backbone/configs/med_config.json", vit="base", queue_size=8)
    n_online = sum(p.numel() for n, p in full.named_parameters() if p.requires_grad)
    assert n_online == 85_798_656 + 137_849_088 + 1        # ViT-B/16 @224 + MED BERT-base with cross-attention + temp
    with pytest.raises(RuntimeError):                       # no CPU product path
        full._ensure_flat()


def test_blip_front_pos_embed_and_transforms():
    from PIL import Image
    from uniir_amd import blip_front
    pe = torch.arange(1 * 5 * 4, dtype=torch.float32).view(1, 5, 4)            # 2x2 grid + cls
    out = blip_front.interpolate_pos_embed(pe, 16)
    assert out.shape == (1, 17, 4) and torch.equal(out[:, 0], pe[:, 0])
    ref = torch.nn.functional.interpolate(pe[:, 1:].reshape(1, 2, 2, 4).permute(0, 3, 1, 2), size=(4, 4), mode="bicubic",
                                          align_corners=False).permute(0, 2, 3, 1).flatten(1, 2)
    assert torch.equal(out[:, 1:], ref)
    assert blip_front.interpolate_pos_embed(pe, 4) is pe
    img = Image.new("RGB", (40, 30), (255, 0, 0))
    for train in (False, True):
        t = blip_front.get_blip_transform(32, is_train=train)(img)
        assert t.shape == (3, 32, 32) and abs(t[0].mean().item() - (1 - 0.48145466) / 0.26862954) < 1e-4


def test_clip_ff_state_dict_layout_and_t5_group():
    """checkpoint layout of the reference's CLIPFeatureFusion: clip_model.* without text_projection, t5_layers.block...;
    every T5 parameter forms the third optimizer group"""
    from models.uniir_clip.clip_featurefusion.clip_ff import CLIPFeatureFusion
    m = CLIPFeatureFusion("ViT-B/32", device="cpu")
    keys = set(m.state_dict().keys())
    assert "clip_model.text_projection" not in keys and "clip_model.visual.proj" in keys
    for k in ("t5_layers.block.0.layer.0.SelfAttention.q.weight", "t5_layers.block.0.layer.0.SelfAttention.relative_attention_bias.weight",
              "t5_layers.block.1.layer.1.DenseReluDense.wo.weight", "t5_layers.final_layer_norm.weight"):
        assert k in keys, k
    assert "t5_layers.block.1.layer.0.SelfAttention.relative_attention_bias.weight" not in keys
    assert m.state_dict()["t5_layers.block.0.layer.0.SelfAttention.q.weight"].shape == (768, 512)     # 12 heads x 64, d_model 512
    assert m.state_dict()["t5_layers.block.0.layer.1.DenseReluDense.wi.weight"].shape == (2048, 512)
    grp = m.t5_optimizer_group(lr=1e-4)
    assert len(grp.params) == 18 and sum(p.numel() for p in grp.params) == sum(v.numel() for k, v in m.state_dict().items() if k.startswith("t5_layers."))
    with pytest.raises(RuntimeError):
        grp.store()            # no CPU product path
    from uniir_amd.clipff_model import rel_bucket_table
    t = rel_bucket_table(334)
    assert t.shape == (667,) and int(t[333]) == 0 and int(t[334]) == 17 and int(t[332]) == 1 and int(t.max()) == 31 and int(t[0]) == 15


def test_hard_negative_selection_and_jsonl_helpers(tmp_path):
    """select_hard_negatives == the reference's filter / multiplier + remainder padding / truncation (mbeir_retriever.py
    :664-680); the jsonl helpers keep the reference's argument order and return shapes (preprocessing/utils.py:198-269)"""
    from mbeir_retriever import select_hard_negatives
    from data.preprocessing.utils import count_entries_in_file, load_jsonl_as_list, save_list_as_jsonl

    def restated(retrieved, pos, neg, n):
        hard = [d for d in retrieved if d not in pos and d not in neg]
        if 0 < len(hard) < n:
            hard = hard * (n // len(hard)) + hard[: n % len(hard)]
        return hard[:n]

    retrieved = [f"1:{i}" for i in (5, 9, 2, 7, 3, 8, 1)]
    for pos, neg, n in ((["1:9"], [], 3), (["1:9"], ["1:5", "1:2"], 10), (retrieved, [], 4), ([], ["1:1"], 6), ([], [], 7),
                        (["1:5", "1:9", "1:2", "1:7", "1:3", "1:8"], [], 5)):
        assert select_hard_negatives(retrieved, pos, neg, n) == restated(retrieved, pos, neg, n)
    path = str(tmp_path / "x.jsonl")
    rows = [{"qid": "1:1", "neg_cand_list": ["1:2"]}, {"qid": "1:2", "neg_cand_list": []}]
    save_list_as_jsonl(rows, path)
    assert load_jsonl_as_list(path) == rows and count_entries_in_file(path) == (2, rows)
    save_list_as_jsonl(rows[:1], path, mode="a")
    assert count_entries_in_file(path)[0] == 3


def test_clip_load_reads_torchscript_archives_and_plain_state_dicts(tmp_path):
    """the published CLIP files are TorchScript archives whose state_dict carries a few extra entries; a plain state dict
    (or a checkpoint with a "state_dict" entry) must load the same way"""
    import torch
    from oracle import clip_oracle as O
    from uniir_amd import clip_front, clip_model
    cfg = O.tiny_config()
    clip_model.CLIP_CONFIGS["tiny-load"] = cfg
    sd = O.init_state_dict(cfg, seed=4)

    class Node(torch.nn.Module):
        def forward(self):
            return 0

    root = Node()
    for name, value in list(sd.items()) + [("input_resolution", torch.tensor(cfg["image_resolution"]))]:
        node, parts = root, name.split(".")
        for part in parts[:-1]:
            if not hasattr(node, part):
                node.add_module(part, Node())
            node = getattr(node, part)
        if value.is_floating_point():
            node.register_parameter(parts[-1], torch.nn.Parameter(value.clone()))
        else:
            node.register_buffer(parts[-1], value.clone())
    torch.jit.script(root).save(str(tmp_path / "tiny-load.pt"))
    model, _ = clip_front.load("tiny-load", device=None, download_root=str(tmp_path))
    got = model.state_dict()
    assert all(torch.equal(got[k], sd[k].float()) for k in sd)
    os.remove(str(tmp_path / "tiny-load.pt"))
    torch.save({"state_dict": sd}, str(tmp_path / "tiny-load.pt"))
    model2, _ = clip_front.load("tiny-load", device=None, download_root=str(tmp_path))
    assert all(torch.equal(model2.state_dict()[k], sd[k].float()) for k in sd)


def test_checkpoint_loader_survives_pickled_config_objects(tmp_path):
    """published UniIR checkpoints carry an omegaconf DictConfig under "config"; torch >= 2.6 rejects unknown globals by
    default and omegaconf is not installed: the loader must still hand back the weights (and plain checkpoints as is)"""
    import sys
    import types
    import torch
    from uniir_amd.host_utils import load_checkpoint_file
    mod = types.ModuleType("omegaconf_like.dictconfig")
    pkg = types.ModuleType("omegaconf_like")

    class DictConfig(dict):
        pass

    DictConfig.__module__, DictConfig.__qualname__ = "omegaconf_like.dictconfig", "DictConfig"
    mod.DictConfig = DictConfig
    sys.modules["omegaconf_like"], sys.modules["omegaconf_like.dictconfig"] = pkg, mod
    try:
        ckpt = {"model": {"w": torch.arange(6.0).view(2, 3)}, "config": DictConfig(a=1), "epoch": 3}
        torch.save(ckpt, str(tmp_path / "published.pth"))
    finally:
        del sys.modules["omegaconf_like"], sys.modules["omegaconf_like.dictconfig"]
    got = load_checkpoint_file(str(tmp_path / "published.pth"))          # the class is not importable any more
    assert torch.equal(got["model"]["w"], torch.arange(6.0).view(2, 3)) and got["epoch"] == 3
    torch.save({"model": {"w": torch.ones(2)}, "epoch": 0}, str(tmp_path / "plain.pth"))
    assert torch.equal(load_checkpoint_file(str(tmp_path / "plain.pth"))["model"]["w"], torch.ones(2))


def test_checkpoint_loader_does_not_resolve_arbitrary_globals(tmp_path):
    """the fallback unpickler must not import whatever a file names (os.system ...): such globals become inert
    placeholders, the tensors still load"""
    import pickle
    import torch
    from uniir_amd.host_utils import load_checkpoint_file

    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ("touch " + str(tmp_path / "pwned"),))

    path = str(tmp_path / "evil.pth")
    torch.save({"model": {"w": torch.arange(4.0)}, "config": Evil()}, path)
    sd = load_checkpoint_file(path)
    assert torch.equal(sd["model"]["w"], torch.arange(4.0))
    assert not (tmp_path / "pwned").exists()

    # gadgets that live UNDER the allowed roots must not resolve either (an exact (module, name) list, not roots)
    class TorchGadget:
        def __reduce__(self):
            import torch.utils.collect_env as ce
            return (ce.run, ("touch " + str(tmp_path / "pwned_torch"),))

    class NumpyGadget:
        def __reduce__(self):
            from numpy.testing._private.utils import runstring
            return (runstring, ("open(%r, 'w').close()" % str(tmp_path / "pwned_numpy"), {}))

    class BuiltinGadget:
        def __reduce__(self):
            return (eval, ("__import__('os').system('touch %s')" % str(tmp_path / "pwned_eval"),))

    for i, g in enumerate((TorchGadget(), NumpyGadget(), BuiltinGadget())):
        path = str(tmp_path / f"evil{i}.pth")
        torch.save({"model": {"w": torch.arange(4.0), "h": torch.ones(2, dtype=torch.bfloat16)}, "config": g, "odd": Evil()}, path)
        sd = load_checkpoint_file(path)
        assert torch.equal(sd["model"]["w"], torch.arange(4.0)) and sd["model"]["h"].dtype == torch.bfloat16
    assert not any((tmp_path / n).exists() for n in ("pwned", "pwned_torch", "pwned_numpy", "pwned_eval"))


def test_clip_load_refuses_missing_pretrained_weights(tmp_path, monkeypatch):
    """a wrong pretrained_clip_model_dir must not silently train from random weights"""
    import pytest
    from oracle import clip_oracle as O
    from uniir_amd import clip_front, clip_model
    clip_model.CLIP_CONFIGS["tiny-missing"] = O.tiny_config()
    monkeypatch.delenv("UNIIR_ALLOW_RANDOM_INIT", raising=False)
    with pytest.raises(FileNotFoundError):
        clip_front.load("tiny-missing", device=None, download_root=str(tmp_path))
    monkeypatch.setenv("UNIIR_ALLOW_RANDOM_INIT", "1")
    with pytest.warns(UserWarning):
        clip_front.load("tiny-missing", device=None, download_root=str(tmp_path))


# the key sets of the reference's large CLIP_SF configs (train/inbatch/inbatch.yaml, eval/inbatch/{embed,index,retrieval}.yaml,
# SURVEY.md section 5.6): key NAMES are the drop-in surface; the values below are this test's own
_YAML_KEYS = {
    "inbatch": """experiment.instruct_status experiment.exp_name experiment.description experiment.path_suffix wandb_config.enabled
        wandb_config.experiment_name logger_config.logger_out_dir logger_config.logger_out_file_name data_config.image_size
        data_config.hard_neg_num data_config.in_batch_neg_num data_config.shuffle_cand data_config.returns
        data_config.enable_query_instruct data_config.query_instruct_path data_config.train_query_data_path
        data_config.train_cand_pool_path data_config.val_query_data_path data_config.val_cand_pool_path
        dataloader_config.num_workers dataloader_config.train_batch_size dataloader_config.valid_batch_size
        trainer_config.gradient_accumulation_steps trainer_config.num_train_epochs trainer_config.learning_rate
        trainer_config.warmup_steps trainer_config.eval_steps trainer_config.print_freq evaluator.enable_eval
        evaluator.eval_freq evaluator.print_freq model.name model.short_name model.size model.clip_vision_model_name
        model.pretrained_clip_model_dir model.gather_embeddings model.ckpt_config.ckpt_dir model.ckpt_config.resume_training
        model.ckpt_config.ckpt_name seed dist_config.dist_url""",
    "embed": """experiment.instruct_status experiment.exp_name experiment.description experiment.path_suffix
        embed_config.embed_dir_name embed_config.use_fp16 embed_config.train_datasets_config.enable_embed
        embed_config.train_datasets_config.datasets_name embed_config.train_datasets_config.correspond_cand_pools_name
        embed_config.val_datasets_config.enable_embed embed_config.val_datasets_config.datasets_name
        embed_config.val_datasets_config.correspond_cand_pools_name embed_config.test_datasets_config.enable_embed
        embed_config.test_datasets_config.datasets_name embed_config.test_datasets_config.correspond_cand_pools_name
        embed_config.cand_pools_config.enable_embed embed_config.cand_pools_config.embed_union_pool
        embed_config.cand_pools_config.cand_pools_name_to_embed dataloader_config.num_workers dataloader_config.batch_size
        model.name model.short_name model.size model.clip_vision_model_name model.pretrained_clip_model_dir
        model.ckpt_config.ckpt_dir model.ckpt_config.ckpt_name data_config.image_size data_config.shuffle_cand
        data_config.train_dir_name data_config.val_dir_name data_config.test_dir_name data_config.cand_pool_dir_name
        data_config.query_instruct_path dist_config.dist_url seed""",
    "index": """experiment.instruct_status experiment.exp_name experiment.description experiment.path_suffix
        index_config.faiss_config.idx_type index_config.faiss_config.dim index_config.faiss_config.metric
        index_config.embed_dir_name index_config.index_dir_name index_config.cand_pools_config.enable_idx
        index_config.cand_pools_config.cand_pools_name_to_idx model.name model.short_name model.size""",
    "retrieval": """experiment.instruct_status experiment.exp_name experiment.description experiment.path_suffix
        retrieval_config.embed_dir_name retrieval_config.index_dir_name retrieval_config.results_dir_name
        retrieval_config.qrel_dir_name retrieval_config.write_to_tsv retrieval_config.raw_retrieval
        retrieval_config.retrieve_image_text_pairs retrieval_config.query_dir_name retrieval_config.candidate_dir_name
        retrieval_config.train_datasets_config.enable_retrieve retrieval_config.train_datasets_config.datasets_name
        retrieval_config.train_datasets_config.correspond_cand_pools_name retrieval_config.val_datasets_config.enable_retrieve
        retrieval_config.val_datasets_config.datasets_name retrieval_config.val_datasets_config.correspond_cand_pools_name
        retrieval_config.val_datasets_config.correspond_qrels_name retrieval_config.test_datasets_config.enable_retrieve
        retrieval_config.test_datasets_config.datasets_name retrieval_config.test_datasets_config.correspond_cand_pools_name
        retrieval_config.test_datasets_config.correspond_qrels_name retrieval_config.test_datasets_config.correspond_metrics_name
        model.name model.short_name model.size""",
}


def _value_for(key):
    leaf = key.rsplit(".", 1)[-1]
    special = {"experiment.path_suffix": "${model.short_name}/${model.size}/${experiment.instruct_status}/${experiment.exp_name}/",
               "wandb_config.experiment_name": "${experiment.description}", "trainer_config.learning_rate": "1e-5",
               "data_config.image_size": "224, 224", "model.short_name": "CLIP_SF", "model.size": "Large",
               "experiment.instruct_status": "Instruct", "experiment.exp_name": "InBatch", "experiment.description": "toy run",
               "model.clip_vision_model_name": "ViT-L/14", "index_config.faiss_config.dim": 768, "seed": 2023,
               "dist_config.dist_url": "env://", "data_config.returns": None}
    if key in special:
        return special[key]
    if leaf.startswith("enable") or leaf in ("use_fp16", "shuffle_cand", "write_to_tsv", "raw_retrieval", "gather_embeddings",
                                            "resume_training", "embed_union_pool", "enabled", "retrieve_image_text_pairs"):
        return True
    if leaf.endswith("_name") and ("datasets" in leaf or "pools" in leaf or "qrels" in leaf or "metrics" in leaf):
        return ["visualnews_task0", "mscoco_task3"] if "metrics" not in leaf else ["Recall@1, Recall@5", "Recall@10"]
    if leaf.startswith("cand_pools_name"):
        return ["visualnews_task0", "UNION"]
    if leaf in ("num_workers", "train_batch_size", "valid_batch_size", "batch_size", "gradient_accumulation_steps", "num_train_epochs",
                "warmup_steps", "eval_steps", "print_freq", "eval_freq", "hard_neg_num", "in_batch_neg_num"):
        return 5
    return f"some/{leaf}"


def test_config_loader_takes_the_full_key_set_of_the_reference_yamls(tmp_path):
    """every key of the reference's large inbatch / embed / index / retrieval YAMLs survives common/config.py with its type:
    ${a.b} interpolation inside and across sections, scientific-notation floats, lists, nulls, booleans, nested sections,
    assignment of uniir_dir / dist_config.gpu_id the way the entry points do it, and to_yaml round trip"""
    import sys
    import yaml
    sys.path.insert(0, os.path.join(ROOT, "uniir_amd", "src", "common"))
    from config import OmegaConf
    for name, keys in _YAML_KEYS.items():
        tree = {}
        for key in keys.split():
            node = tree
            parts = key.split(".")
            for part in parts[:-1]:
                node = node.setdefault(part, {})
            node[parts[-1]] = _value_for(key)
        path = tmp_path / f"{name}.yaml"
        path.write_text(yaml.safe_dump(tree, sort_keys=False))
        cfg = OmegaConf.load(str(path))
        for key in keys.split():
            cur = cfg
            for part in key.split("."):
                cur = getattr(cur, part)                     # attribute access all the way down
        assert cfg.experiment.path_suffix == "CLIP_SF/Large/Instruct/InBatch/"
        assert cfg.model.size == "Large" and isinstance(cfg.model, dict)
        if name == "inbatch":
            assert cfg.trainer_config.learning_rate == 1e-5 and isinstance(cfg.trainer_config.learning_rate, float)
            assert cfg.wandb_config.experiment_name == "toy run" and cfg.data_config.returns is None
            assert cfg.model.gather_embeddings is True and cfg.seed == 2023
            cfg.uniir_dir, cfg.mbeir_data_dir = "/data/UniIR", "/data/M-BEIR"
            cfg.dist_config.gpu_id, cfg.dist_config.distributed_mode = 3, True
            assert cfg.dist_config.gpu_id == 3 and cfg.dist_config.dist_url == "env://"
        if name == "retrieval":
            assert cfg.retrieval_config.test_datasets_config.correspond_metrics_name == ["Recall@1, Recall@5", "Recall@10"]
        again = yaml.safe_load(OmegaConf.to_yaml(cfg))
        assert again["experiment"]["path_suffix"] == "CLIP_SF/Large/Instruct/InBatch/"


def test_logical_sub_shards_of_a_resident_pool():
    """retrieval.subshard_bounds (round 4 / 5): the row ranges a resident shard is searched in.  uniir_topk_ip addresses a shard through
    31-bit buffer offsets, so every range stays below 2 GiB, starts on a 32-row boundary (aligned inverse norms, an even number of
    whole scan groups) and -- round 5 -- holds at most the 786 432 rows the fused tail's register-resident selection takes; the
    ranges are equal (the last one shorter) and cover the shard exactly; the 5.6 M x 768 M-BEIR pool on one GPU
    (mbeir_retriever.py:196-206 with a single visible device) is 8 ranges of 700 000 rows"""
    from uniir_amd import retrieval
    assert retrieval.subshard_bounds(700_000, 768) == [(0, 700_000)]
    assert retrieval.subshard_bounds(1_398_096, 768) == [(0, 1_398_096)]          # 2^31 - 4 608 bytes
    assert len(retrieval.subshard_bounds(1_398_112, 768)) == 2
    for n, d in ((5_600_000, 768), (5_600_001, 768), (1_398_112, 768), (9_000_000, 512), (2_100_000, 1024), (40_000_000, 64)):
        b = retrieval.subshard_bounds(n, d)
        assert b[0][0] == 0 and b[-1][1] == n and all(x[1] == y[0] for x, y in zip(b, b[1:]))
        assert all(lo % 32 == 0 and 0 < (hi - lo) * d * 2 < 2 ** 31 and hi - lo <= retrieval.FUSED_TAIL_ROWS for lo, hi in b)
        assert len({hi - lo for lo, hi in b[:-1]}) <= 1 and b[-1][1] - b[-1][0] <= b[0][1] - b[0][0]
        cap = min(((2 ** 31 - 1) // (d * 2)) // 32 * 32, retrieval.FUSED_TAIL_ROWS)
        assert len(b) == -(-n // cap)                   # as few ranges as the two bounds allow
    assert retrieval.subshard_bounds(5_600_000, 768) == [(lo, lo + 700_000) for lo in range(0, 5_600_000, 700_000)]


def test_text_row_offsets_of_a_token_batch():
    """clip_model.text_row_offsets (round 4, the packed text tower): caption i owns argmax(tokens[i]) + 1 rows (upstream pools at
    the EOT = the largest id, clip_sf.py:43-44 / CLIP.encode_text); int32 prefix sums on the tokens' device + the total on the host;
    remembered ON the tensor (never by address), recomputed when the tensor is modified in place; lengths attached by a prefetcher
    (`_uniir_lens`) are used instead of a device read"""
    from uniir_amd.clip_model import text_row_offsets
    tok = torch.zeros(4, 77, dtype=torch.int32)
    for i, n in enumerate((5, 77, 1, 30)):
        tok[i, :n] = torch.arange(1, n + 1, dtype=torch.int32)      # the largest id sits at position n - 1
    off, live = text_row_offsets(tok)
    assert off.dtype == torch.int32 and off.tolist() == [0, 5, 82, 83, 113] and live == 113
    assert text_row_offsets(tok)[0] is off                            # remembered on the tensor object
    tok[2, 9] = 1000                                                  # in-place change -> version counter -> recomputed
    off2, live2 = text_row_offsets(tok)
    assert off2.tolist() == [0, 5, 82, 92, 122] and live2 == 122
    other = tok.clone()                                               # a new tensor never inherits an answer
    other[0, 40] = 2000
    assert text_row_offsets(other)[0].tolist() == [0, 41, 118, 128, 158]
    hinted = torch.zeros(3, 77, dtype=torch.int32)
    hinted._uniir_lens = torch.tensor([7, 8, 9])
    assert text_row_offsets(hinted)[0].tolist() == [0, 7, 15, 24]


def test_executed_flops_of_the_packed_step():
    """bench.executed_flop_per_pair: the FLOPs the default (packed) train step executes per pair = SURVEY 8(d)'s count with every
    caption's 77 positions replaced by its live length.  Full-length captions reproduce the 1.052 TFLOP per pair the unpacked step is
    priced with (text forward 13.30 GFLOP per item); shorter ones are priced lower, by exactly 3 x the text forward difference"""
    import bench
    from uniir_amd.clip_model import CLIP_CONFIGS
    cfg = CLIP_CONFIGS["ViT-L/14"]
    W, L, E = cfg["transformer_width"], cfg["transformer_layers"], cfg["embed_dim"]
    text_fwd = lambda n: L * (24.0 * W * W * n + 4.0 * n * n * W) + 2.0 * W * E
    assert abs(text_fwd(77) - 13.30e9) < 0.01e9
    full = torch.zeros(8, 77, dtype=torch.int32)
    full[:, 76] = 49407
    f, live, dense = bench.executed_flop_per_pair(cfg, full, bench.FLOP_PER_PAIR["ViT-L/14"])
    assert f == bench.FLOP_PER_PAIR["ViT-L/14"] and live == dense == 8 * 77
    short = torch.zeros(8, 77, dtype=torch.int32)
    short[:, 19] = 49407                                              # 20 live rows per caption
    f2, live2, _ = bench.executed_flop_per_pair(cfg, short, bench.FLOP_PER_PAIR["ViT-L/14"])
    assert live2 == 160 and abs((f - f2) - 3 * 2 * (text_fwd(77) - text_fwd(20))) < 1.0


def test_executed_flops_of_the_pooled_last_block():
    """bench.pooled_last_block_saving / executed_flop_per_pair(pool_last=True): what clip_model.pool_last_block leaves out, per item and
    forward, is the last block's 24 W^2 T + 4 T^2 W minus what still runs (K | V of every row 4 W^2 T, the pooled row's Q / out_proj /
    MLP 20 W^2, one query's attention 4 T W): 5.64 GFLOP of the ViT-L/14 image tower's 162.03 (3.5 %)"""
    import bench
    from uniir_amd.clip_model import CLIP_CONFIGS
    cfg = CLIP_CONFIGS["ViT-L/14"]
    W, T = 1024, 257
    full_block = 24.0 * W * W * T + 4.0 * T * T * W
    kept = 4.0 * W * W * T + 20.0 * W * W + 4.0 * T * W
    assert bench.pooled_last_block_saving(W, T) == full_block - kept
    assert abs(bench.pooled_last_block_saving(W, T) - 5.64e9) < 0.01e9
    assert abs(24 * full_block - 162.03e9) < 1.2e9            # 24 such blocks are the image tower (+ patch embedding and projection)
    toks = torch.zeros(8, 77, dtype=torch.int32)
    toks[:, 19] = 49407                                              # 20 live rows per caption
    f0, _, _ = bench.executed_flop_per_pair(cfg, toks, bench.FLOP_PER_PAIR["ViT-L/14"], True, False)
    f1, _, _ = bench.executed_flop_per_pair(cfg, toks, bench.FLOP_PER_PAIR["ViT-L/14"], True, True)
    Wt = cfg["transformer_width"]
    want = 3 * 2 * (bench.pooled_last_block_saving(W, T) + bench.pooled_last_block_saving(Wt, 20))
    assert abs((f0 - f1) - want) < 1.0
    # the reference shape: neither packing nor pooling -> SURVEY's count whatever the captions are
    assert bench.executed_flop_per_pair(cfg, toks, bench.FLOP_PER_PAIR["ViT-L/14"], False, False)[0] == bench.FLOP_PER_PAIR["ViT-L/14"]
    assert bench.vision_flop_per_item_fwd(cfg, "ViT-L/14", True) == bench.VISION_FLOP_PER_ITEM_FWD["ViT-L/14"] - bench.pooled_last_block_saving(W, T)


def test_text_pack_row_offsets_and_row_map():
    """blip_model.TextPack (host arithmetic; the device tensors are plain copies): row offsets = prefix sums of the valid lengths, the
    row map sends packed row r of item m, position t to the padded row m * L + t; build() declines batches with nothing to drop or
    with an empty caption (the class token must exist)"""
    from uniir_amd.blip_model import TextPack
    lens = torch.tensor([3, 1, 5, 2], dtype=torch.int32)
    p = TextPack(lens, 5, torch.device("cpu"))
    assert p.R == 11 and p.M == 4 and p.L == 5
    assert p.row_off.tolist() == [0, 3, 4, 9, 11] and p.row_off.dtype == torch.int32
    assert p.row_map.tolist() == [0, 1, 2, 5, 10, 11, 12, 13, 14, 15, 16]
    assert TextPack.build(lens, 5, torch.device("cpu")).R == 11
    assert TextPack.build(torch.tensor([5, 5]), 5, torch.device("cpu")) is None          # every caption full: nothing to pack
    assert TextPack.build(torch.tensor([2, 0, 3]), 5, torch.device("cpu")) is None       # an empty caption: padded path
    assert TextPack.build(torch.tensor([2, 6]), 5, torch.device("cpu")) is None          # longer than the context: not a prefix mask
