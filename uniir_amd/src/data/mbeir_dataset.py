"""M-BEIR datasets and collators: the producers of the batch dictionary the model consumes.
Drop-in for UniIR src/data/mbeir_dataset.py (same class names, constructor arguments and output dictionaries):
  MBEIRMainDataset :115-317, MBEIRInferenceOnlyDataset :320-352, MBEIRCandidatePoolDataset :355-411,
  MBEIRMainCollator :440-526 (flat interleaved item order + index_mapping), MBEIRInferenceOnlyCollator :529-569,
  MBEIRCandidatePoolCollator :572-610.  CPU-side Python only; nothing here is on the GPU hot path.
"""
import json
import os
import random
from enum import Enum

import torch
from torch.utils.data import Dataset

from data.preprocessing.utils import format_string, get_mbeir_task_id, hash_did, hash_qid


class Mode(Enum):
    TRAIN = "train"
    EVAL = "eval"


def _read_jsonl(root, rel):
    path = os.path.join(root, rel)
    assert os.path.exists(path), f"Data Path {path} does not exist"
    assert path.endswith(".jsonl"), f"Data Path {path} is not a jsonl file"
    with open(path, "r") as f:
        return [json.loads(line) for line in f]


class MBEIRDatasetBase(Dataset):
    def __init__(self, mbeir_data_dir, img_preprocess_fn):
        self.mbeir_data_dir = mbeir_data_dir
        self.img_preprocess_fn = img_preprocess_fn or (lambda x: x)

    def _load_query_instructions(self, instructions_path):
        path = os.path.join(self.mbeir_data_dir, instructions_path)
        assert os.path.exists(path), f"Instructions Path {path} does not exist"
        assert path.endswith(".tsv"), f"Instructions Path {path} is not a tsv file"
        table = {}
        with open(path, "r") as f:
            next(f)  # header
            for line in f:
                cols = line.strip().split("\t")
                # key = dataset_id, query_modality, cand_modality ; prompts = remaining non-empty columns
                table[f"{cols[3]}, {cols[0]}, {cols[1]}"] = [p for p in cols[4:] if p]
        self.query_instructions = table

    def _load_and_preprocess_image(self, img_path):
        if not img_path:
            return None
        from PIL import Image
        path = os.path.join(self.mbeir_data_dir, img_path)
        assert os.path.exists(path), f"Image Path {path} does not exist"
        return self.img_preprocess_fn(Image.open(path).convert("RGB"))

    def _get_random_query_prompt(self, dataset_id, query_modality, cand_modality):
        """one instruction of the (dataset, query modality, candidate modality) row of the TSV, cleaned up like every text"""
        lookup = ", ".join((str(dataset_id), str(query_modality), str(cand_modality)))
        choices = self.query_instructions.get(lookup) or []
        if not choices:
            raise AssertionError(f"no instruction row for ({lookup}) in the query-instruction TSV")
        picked = format_string(random.choice(choices))
        if not picked:
            raise AssertionError(f"instruction row ({lookup}) holds an empty prompt")
        return picked

    def _item(self, txt, img_path):
        return {"txt": txt, "img": self._load_and_preprocess_image(img_path)}


class MBEIRMainDataset(MBEIRDatasetBase):
    def __init__(self, mbeir_data_dir, query_data_path, cand_pool_path, query_instruct_path, img_preprocess_fn,
                 mode=Mode.TRAIN, enable_query_instruct=True, shuffle_cand=True, hard_neg_num=0, returns=None,
                 print_config=True):
        super().__init__(mbeir_data_dir, img_preprocess_fn)
        self.query_data = _read_jsonl(mbeir_data_dir, query_data_path)
        self.cand_pool = {}
        for entry in _read_jsonl(mbeir_data_dir, cand_pool_path):
            assert entry.get("did"), f"Cannot find did for {entry}"
            self.cand_pool[entry["did"]] = entry
        self._load_query_instructions(query_instruct_path)
        self.mode, self.shuffle_cand, self.hard_neg_num = mode, shuffle_cand, hard_neg_num
        self.enable_query_instruct = enable_query_instruct
        self.returns = {"hashed_qid": True, "task_id": False, "hashed_p_did": False, **(returns or {})}
        if print_config:
            print(f"\n---Mbeir Dataset Config---\nMode: {mode}\nQuery Data Path: {query_data_path}\n"
                  f"Candidate Pool Path: {cand_pool_path}\nEnable Query Instructions: {enable_query_instruct}\n"
                  f"Shuffle Candidates: {shuffle_cand}\nHard Negative Number: {hard_neg_num}\nReturns: {self.returns}\n"
                  f"--------------------------\n")

    def __len__(self):
        return len(self.query_data)

    def __getitem__(self, index):
        e = self.query_data[index]
        qid = e.get("qid")
        q_ds = qid.split(":")[0] if qid else None
        q_mod = e.get("query_modality")
        pos_list = e.get("pos_cand_list", [])
        assert len(pos_list) > 0, f"Cannot find positive candidates for {e}"
        if self.mode == Mode.EVAL:  # OVEN / INFOSEEK: keep the positives of the query's own dataset
            pos_list = [d for d in pos_list if d.split(":")[0] == q_ds]
        pos_did = random.choice(pos_list) if self.shuffle_cand else pos_list[0]
        pos = self.cand_pool.get(pos_did)
        assert pos, f"Cannot find positive candidate {pos_did} for {e}"
        pos_mod = pos.get("modality")
        prompt = self._get_random_query_prompt(q_ds, q_mod, pos_mod)
        raw_q = e.get("query_txt") or ""
        q_txt = format_string(f"{prompt} {raw_q}") if self.enable_query_instruct else format_string(raw_q)
        negs = []
        if self.mode == Mode.TRAIN and self.hard_neg_num > 0:
            neg_ids = e.get("neg_cand_list", [])
            assert len(neg_ids) > 0, f"Cannot find negative candidates for {e}"
            if self.shuffle_cand:
                random.shuffle(neg_ids)
            for i in range(self.hard_neg_num):  # wrap around when there are fewer negatives than requested
                neg = self.cand_pool.get(neg_ids[i % len(neg_ids)])
                negs.append((format_string(neg.get("txt") or ""), neg.get("img_path")))
        instance = {"query": self._item(q_txt, e.get("query_img_path"))}
        if self.mode == Mode.EVAL:
            if self.returns.get("hashed_qid"):
                instance["qid"] = hash_qid(qid)
            if self.returns.get("task_id"):
                instance["task_id"] = get_mbeir_task_id(q_mod, pos_mod)
        if self.mode == Mode.TRAIN:
            if self.returns.get("hashed_p_did"):
                instance["p_did"] = hash_did(pos_did)
            instance["pos_cand"] = self._item(format_string(pos.get("txt") or ""), pos.get("img_path"))
            if negs:
                instance["neg_cand_list"] = [self._item(t, p) for t, p in negs]
        return instance


class MBEIRInferenceOnlyDataset(MBEIRDatasetBase):
    def __init__(self, mbeir_data_dir, queries, query_instruct_path, img_preprocess_fn, enable_query_instruct=True,
                 returns=None, print_config=True):
        super().__init__(mbeir_data_dir, img_preprocess_fn)
        self.query_data = queries
        self._load_query_instructions(query_instruct_path)
        self.enable_query_instruct = enable_query_instruct
        self.returns = {"hashed_qid": True, "task_id": False, "hashed_p_did": False, **(returns or {})}

    def __len__(self):
        return len(self.query_data)

    def __getitem__(self, index):
        e = self.query_data[index]
        qid = e.get("qid")
        q_ds = qid.split(":")[0] if qid else None
        q_mod, c_mod = e.get("query_modality"), e.get("candidate_modality")
        raw_q = e.get("query_txt") or ""
        if self.enable_query_instruct:
            q_txt = format_string(f"{self._get_random_query_prompt(q_ds, q_mod, c_mod)} {raw_q}")
        else:
            q_txt = format_string(raw_q)
        instance = {"query": self._item(q_txt, e.get("query_img_path"))}
        if self.returns.get("hashed_qid"):
            instance["qid"] = hash_qid(qid)
        if self.returns.get("task_id"):
            instance["task_id"] = get_mbeir_task_id(q_mod, c_mod)
        return instance


class MBEIRCandidatePoolDataset(MBEIRDatasetBase):
    def __init__(self, mbeir_data_dir, cand_pool_data_path, img_preprocess_fn, returns=None, print_config=True):
        super().__init__(mbeir_data_dir, img_preprocess_fn)
        self.cand_pool = _read_jsonl(mbeir_data_dir, cand_pool_data_path)
        self.returns = {"src_content": False, "hashed_did": True, **(returns or {})}
        if print_config:
            print(f"\n---Mbeir Candidate Pool Dataset Config---\nCandidate Pool Path: {cand_pool_data_path}\n"
                  f"Returns: {self.returns}\n--------------------------\n")

    def __len__(self):
        return len(self.cand_pool)

    def __getitem__(self, index):
        e = self.cand_pool[index]
        instance = {"txt": format_string(e.get("txt") or ""), "img": self._load_and_preprocess_image(e.get("img_path")),
                    "modality": e.get("modality")}
        if self.returns.get("hashed_did"):
            instance["did"] = hash_did(e.get("did"))
        if self.returns.get("src_content"):
            instance["src_content"] = e.get("src_content")
        return instance


class MBEIRCollatorBase(object):
    def __init__(self, tokenizer, image_size):
        self.tokenizer = tokenizer
        self.H, self.W = (image_size, image_size) if isinstance(image_size, int) else image_size
        self.padded_image = torch.zeros((3, self.H, self.W))   # black image for image-less items
        self.padded_txt = ""                                    # empty string for text-less items
        self.raw_transform = None          # set to the RawImageTransform used as img_preprocess_fn to defer to the GPU

    def _pack(self, items):
        """items: list of {"txt","img"} -> the four flat tensors; missing modalities are padded and masked out"""
        txts, imgs, tmask, imask = [], [], [], []
        for it in items:
            has_txt = it["txt"] not in (None, "")
            has_img = it["img"] is not None
            txts.append(it["txt"] if has_txt else self.padded_txt)
            imgs.append(it["img"] if has_img else self.padded_image)
            tmask.append(int(has_txt))
            imask.append(int(has_img))
        raw = [im for im in imgs if hasattr(im, "data") and not isinstance(im, torch.Tensor)]
        if raw:      # deferred device transform (uniir_amd.clip_front.RawImageTransform): keep the decoded bytes
            from uniir_amd.clip_front import RawImageBatch
            images = RawImageBatch([im.data if hasattr(im, "data") and not isinstance(im, torch.Tensor) else None for im in imgs],
                                   self.raw_transform)
        else:
            images = torch.stack(imgs, dim=0)
        out = {"txt_batched": self.tokenizer(txts), "image_batched": images,
               "txt_mask_batched": torch.tensor(tmask, dtype=torch.long),
               "image_mask_batched": torch.tensor(imask, dtype=torch.long)}
        tb = out["txt_batched"]
        bs = tb["input_ids"].size(0) if hasattr(tb, "input_ids") else len(tb)
        assert bs == out["image_batched"].size(0) == out["txt_mask_batched"].size(0) == out["image_mask_batched"].size(0)
        return out


class MBEIRMainCollator(MBEIRCollatorBase):
    def __init__(self, tokenizer, image_size, mode=Mode.TRAIN):
        super().__init__(tokenizer, image_size)
        self.mode = mode

    def __call__(self, batch):
        keys = ["query"]
        if self.mode == Mode.TRAIN:
            keys.append("pos_cand")
            if "neg_cand_list" in batch[0]:
                keys.append("neg_cand_list")
        qids = [b.pop("qid") for b in batch if b.get("qid") is not None] if self.mode == Mode.EVAL else []
        tids = [b.pop("task_id") for b in batch if b.get("task_id") is not None] if self.mode == Mode.EVAL else []
        pdids = [b.pop("p_did") for b in batch if b.get("p_did") is not None] if self.mode == Mode.TRAIN else []
        index_mapping = {k: [[] for _ in batch] for k in keys}
        flat = []
        for i, inst in enumerate(batch):          # per instance: query, pos_cand, then its hard negatives
            for k in keys:
                for it in (inst[k] if k == "neg_cand_list" else [inst[k]]):
                    index_mapping[k][i].append(len(flat))
                    flat.append(it)
        out = self._pack(flat)
        out["index_mapping"] = index_mapping
        if qids:
            out["qid_list"] = qids
        if tids:
            out["task_id_list"] = tids
        if pdids:
            out["p_did_list"] = torch.tensor(pdids)
        return out


class MBEIRInferenceOnlyCollator(MBEIRCollatorBase):
    def __call__(self, batch):
        out = self._pack([b["query"] for b in batch])
        out["qid_list"] = [b["qid"] for b in batch if b.get("qid") is not None]
        out["task_id_list"] = [b["task_id"] for b in batch if b.get("task_id") is not None]
        return out


class MBEIRCandidatePoolCollator(MBEIRCollatorBase):
    def __call__(self, batch):
        out = self._pack(batch)
        dids = [b["did"] for b in batch if b.get("did") is not None]
        if dids:
            out["did_list"] = dids
        return out
