"""Model factory and file helpers (drop-in for UniIR src/common/utils.py: load_qrel :16-31, load_runfile :35-65,
build_model_from_config :64-153, set_seed).  CLIPScoreFusion, CLIPFeatureFusion and BLIPFeatureFusion are built on the MI355X path."""
import os
import random

import numpy as np
import torch

from uniir_amd.host_utils import load_checkpoint_file


def set_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def load_qrel(filename):
    """lines "qid Q0 did rel task_id" -> ({qid: [relevant dids]}, {qid: task_id}) for rel > 0"""
    qrel, qid_to_taskid = {}, {}
    with open(filename, "r") as f:
        for line in f:
            qid, _, did, rel, task_id = line.strip().split()
            if int(rel) > 0:
                qrel.setdefault(qid, []).append(did)
                qid_to_taskid.setdefault(qid, task_id)
    print(f"Loaded {len(qrel)} queries from {filename}")
    if qrel:
        print(f"Average number of relevant documents per query: {sum(len(v) for v in qrel.values()) / len(qrel):.2f}")
    return qrel, qid_to_taskid


def load_runfile(filename, load_task_id=False):
    runs = {}
    with open(filename, "r") as f:
        for line in f:
            parts = line.strip().split()
            entry = {"did": parts[2], "rank": int(parts[3]), "score": float(parts[4])}
            if load_task_id:
                entry["task_id"] = parts[6]
            runs.setdefault(parts[0], []).append(entry)
    print(f"Loaded results for {len(runs)} queries from {filename}")
    return runs


def build_model_from_config(config):
    name = config.model.name
    if name in ("BLIPFeatureFusion", "BLIPScoreFusion"):
        if name == "BLIPFeatureFusion":
            from models.uniir_blip.blip_featurefusion.blip_ff import BLIPFeatureFusion
        else:
            from models.uniir_blip.blip_scorefusion.blip_sf import BLIPScoreFusion as BLIPFeatureFusion
        mc = config.model
        model = BLIPFeatureFusion(med_config=os.path.join("../models/uniir_blip", "backbone/configs/med_config.json"),
                                  image_size=mc.image_size, vit=mc.vit, vit_grad_ckpt=mc.vit_grad_ckpt,
                                  vit_ckpt_layer=mc.vit_ckpt_layer, embed_dim=mc.embed_dim, queue_size=mc.queue_size,
                                  config=mc)
        ckpt = mc.ckpt_config
        path = os.path.join(config.uniir_dir, ckpt.ckpt_dir, ckpt.ckpt_name)
        assert os.path.exists(path), f"Checkpoint file {path} does not exist."
        print(f"loading {name} checkpoint from {path}")
        model.load_state_dict(load_checkpoint_file(path)["model"])
        return model
    if name not in ("CLIPScoreFusion", "CLIPFeatureFusion"):
        raise NotImplementedError(f"Model {name} is not implemented.")
    if name == "CLIPFeatureFusion":
        from models.uniir_clip.clip_featurefusion.clip_ff import CLIPFeatureFusion as CLIPScoreFusion
    else:
        from models.uniir_clip.clip_scorefusion.clip_sf import CLIPScoreFusion
    mc = config.model
    download_root = os.path.join(config.uniir_dir, mc.pretrained_clip_model_dir)
    model = CLIPScoreFusion(model_name=mc.clip_vision_model_name, download_root=download_root)
    model.float()
    ckpt = mc.ckpt_config
    path = os.path.join(config.uniir_dir, ckpt.ckpt_dir, ckpt.ckpt_name)
    assert os.path.exists(path), f"Checkpoint file {path} does not exist."
    print(f"loading {name} checkpoint from {path}")
    model.load_state_dict(load_checkpoint_file(path)["model"])
    return model
