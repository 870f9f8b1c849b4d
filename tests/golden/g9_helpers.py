"""Helpers shared by tests/golden/make_golden.py (g9: runs the REFERENCE's collators / datasets) and tests/test_host.py (runs the
mirror under uniir_amd/src/data/mbeir_dataset.py on the same instances).  Pure data plumbing: no reference import here."""
import json
import os

import numpy as np
import torch


def g9_tokenizer(txts):
    """deterministic stand-in tokenizer shared by the generator and tests/test_host.py: row = [len, sum(ord) % 997, first, last]"""
    return torch.tensor([[len(t), sum(map(ord, t)) % 997, ord(t[0]) if t else 0, ord(t[-1]) if t else 0] for t in txts],
                        dtype=torch.int32)


def g9_image(v, size=4):
    """instances carry an image as one number (None = no image): the tensor every pixel of which is that number"""
    return None if v is None else torch.full((3, size, size), float(v))


def g9_materialise(spec):
    """fixture instance (JSON) -> what a dataset __getitem__ hands the collator (images as tensors)"""
    def item(it):
        return {"txt": it["txt"], "img": g9_image(it["img"])}
    out = {}
    for k, v in spec.items():
        if k in ("query", "pos_cand"):
            out[k] = item(v)
        elif k == "neg_cand_list":
            out[k] = [item(x) for x in v]
        elif k in ("txt", "img"):
            out[k] = v if k == "txt" else g9_image(v)
        else:
            out[k] = v
    return out


def g9_flatten(batch_out):
    """collator output -> JSON: tensors as lists; images as (per item) mean pixel value + shape"""
    res = {}
    for k, v in batch_out.items():
        if k == "image_batched":
            res["image_shape"] = list(v.shape)
            res["image_mean"] = [float(x) for x in v.reshape(v.shape[0], -1).mean(1)]
        elif isinstance(v, torch.Tensor):
            res[k] = v.tolist()
            res[k + "::dtype"] = str(v.dtype)
        else:
            res[k] = v
    return res


def g9_write_tree(root, tree):
    """materialise G9_TREE (a 3-query / 5-candidate M-BEIR tree with two 6x5 single-colour PNGs) under `root`"""
    from PIL import Image
    os.makedirs(os.path.join(root, "img"), exist_ok=True)
    for rel, rgb in tree["images"].items():
        Image.new("RGB", (6, 5), tuple(rgb)).save(os.path.join(root, rel))
    with open(os.path.join(root, "instructions.tsv"), "w") as f:
        f.write(tree["instructions"])
    with open(os.path.join(root, "cand_pool.jsonl"), "w") as f:
        for c in tree["cand_pool"]:
            f.write(json.dumps(c) + "\n")
    with open(os.path.join(root, "queries.jsonl"), "w") as f:
        for q in tree["queries"]:
            f.write(json.dumps(q) + "\n")


def g9_img_fn(pil):
    """img_preprocess_fn of the fixture: the image's mean colour as a [3,1,1] tensor"""
    a = np.asarray(pil, dtype=np.float32)
    return torch.from_numpy(a.reshape(-1, 3).mean(0)).view(3, 1, 1)


def g9_dataset_rows(md, root):
    """every deterministic __getitem__ of the reference's three dataset classes on the tree (shuffle_cand off; one prompt per
    key so random.choice has a single outcome; hard negatives cycle / truncate to hard_neg_num)"""
    def flat(it):
        return None if it is None else {"txt": it["txt"], "img": None if it["img"] is None else [float(x) for x in it["img"].view(-1)]}
    res = {}
    for tag, kw in {"train_hn2": dict(mode=md.Mode.TRAIN, hard_neg_num=2, enable_query_instruct=True),
                    "train_noinstruct": dict(mode=md.Mode.TRAIN, hard_neg_num=0, enable_query_instruct=False),
                    "eval": dict(mode=md.Mode.EVAL, hard_neg_num=0, enable_query_instruct=True)}.items():
        ds = md.MBEIRMainDataset(root, "queries.jsonl", "cand_pool.jsonl", "instructions.tsv", g9_img_fn, shuffle_cand=False,
                                 returns={"hashed_qid": True, "task_id": True, "hashed_p_did": True}, print_config=False, **kw)
        rows = []
        for i in range(len(ds)):
            inst = ds[i]
            row = {}
            for k, v in inst.items():
                if k in ("query", "pos_cand"):
                    row[k] = flat(v)
                elif k == "neg_cand_list":
                    row[k] = [flat(x) for x in v]
                else:
                    row[k] = v
            rows.append(row)
        res[tag] = rows
    pool = md.MBEIRCandidatePoolDataset(root, "cand_pool.jsonl", g9_img_fn, returns={"hashed_did": True, "src_content": False},
                                        print_config=False)
    res["pool"] = [{k: (flat({"txt": None, "img": v})["img"] if k == "img" else v) for k, v in pool[i].items()} for i in range(len(pool))]
    return res


