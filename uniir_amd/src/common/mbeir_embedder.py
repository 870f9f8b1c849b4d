"""Multi-GPU embedding extraction for M-BEIR (drop-in for UniIR src/common/mbeir_embedder.py).

Mirrors generate_embeds_and_ids_for_dataset_with_gather (:33-120: per-batch forward, .half() per batch, concat, gather
to rank 0), generate_embeds_for_config (:195-461: splits, file names, union pool) and main (:464-495).  Differences,
result-preserving:
  * ragged / empty rank shards are legal (the reference sizes every receive buffer like rank 0's tensor, :70,99, and
    crashes on an empty shard, :60): sizes are exchanged first and the gather is padded to the largest shard;
  * the towers run through libuniir_hip.so in bf16 (the reference: fp16 autocast), stored as fp16 exactly like :56.
"""
import os as _os
import sys as _sys

_SRC = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))          # <repo>/uniir_amd/src
for _p in (_os.path.dirname(_os.path.dirname(_SRC)), _SRC, _os.path.join(_SRC, "common")):
    if _p not in _sys.path:
        _sys.path.insert(0, _p)

import argparse
import gc
import os

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import DataLoader

import dist_utils
from config import OmegaConf
from data.mbeir_dataset import (MBEIRCandidatePoolCollator, MBEIRCandidatePoolDataset, MBEIRMainCollator,
                                MBEIRMainDataset, Mode)
from dist_utils import ContiguousDistributedSampler
from utils import build_model_from_config, set_seed


def _clip_towers(model):
    """the uniir_amd CLIP module behind a CLIP_SF / CLIP_FF model (DDP-wrapped or not), or None (BLIP)"""
    inner = getattr(model, "module", model)
    clip = getattr(inner, "clip_model", None)
    return clip if hasattr(clip, "precision") else None


@torch.no_grad()
def generate_embeds_and_ids_for_dataset_with_gather(model, data_loader, device, use_fp16=True):
    # the reference runs the towers under torch.cuda.amp.autocast(enabled=use_fp16) (:52-56) -- fp16 matmuls: the CLIP towers here take
    # their fp16 forward for the extraction (clip_model.precision = "fp16") and go back to what they were set to afterwards
    clip = _clip_towers(model)
    restore = None
    if clip is not None and use_fp16 and clip.precision == "bf16":
        restore, clip.precision = clip.precision, "fp16"
    try:
        return _generate(model, data_loader, device)
    finally:
        if restore is not None:
            clip.precision = restore


def _generate(model, data_loader, device):
    chunks, id_list = [], []
    for batch in data_loader:
        for k, v in batch.items():
            if isinstance(v, torch.Tensor):
                batch[k] = v.to(device, non_blocking=True)
            elif hasattr(v, "to_device"):                              # deferred device image transform (RawImageBatch)
                batch[k] = v.to_device(torch.device(device))
            elif hasattr(v, "input_ids") and hasattr(v, "items"):     # BLIP: transformers BatchEncoding
                for kk, vv in v.items():
                    v[kk] = vv.to(device)
        emb, ids = model(batch, encode_mbeir_batch=True)
        chunks.append(emb.half())          # fp16 on disk, like the reference
        id_list.extend(ids)
    dim = chunks[0].shape[1] if chunks else None
    if not dist.is_initialized():
        if not chunks:
            return np.zeros((0, 0), dtype=np.float16), id_list
        return torch.cat(chunks, dim=0).cpu().numpy(), id_list
    world, rank = dist.get_world_size(), dist.get_rank()
    # exchange sizes and the embedding width first (some ranks may hold nothing)
    meta = torch.tensor([sum(c.shape[0] for c in chunks), dim or 0], dtype=torch.long, device=device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    sizes = [int(m[0]) for m in metas]
    dim = max(int(m[1]) for m in metas)
    local = torch.cat(chunks, dim=0) if chunks else torch.zeros(0, dim, dtype=torch.float16, device=device)
    padded = torch.zeros(max(max(sizes), 1), dim, dtype=torch.float16, device=device)
    padded[: local.shape[0]] = local
    gathered = [torch.empty_like(padded) for _ in range(world)] if rank == 0 else None
    ids_gathered = [None] * world if rank == 0 else None
    dist.barrier()
    dist.gather(padded, gather_list=gathered, dst=0)
    dist.gather_object(id_list, object_gather_list=ids_gathered, dst=0)
    embedding_list = None
    if rank == 0:
        embedding_list = torch.cat([g[:n] for g, n in zip(gathered, sizes)], dim=0).cpu().numpy()
        id_list = [i for sub in ids_gathered for i in sub]
        assert len(id_list) == embedding_list.shape[0]
        assert len(id_list) == len(set(id_list)), "Hashed IDs should be unique"
    dist.barrier()
    return embedding_list, id_list


def _splits_to_embed(config):
    data_config, embed_config = config.data_config, config.embed_config
    splits = []
    for split in ("train", "val", "test"):
        ec = embed_config.get(f"{split}_datasets_config")
        if ec and ec.enable_embed:
            assert len(ec.datasets_name) == len(ec.correspond_cand_pools_name), "Mismatch between datasets and candidate pools."
            splits.append((split, data_config[f"{split}_dir_name"], list(ec.datasets_name), list(ec.correspond_cand_pools_name)))
    pc = embed_config.get("cand_pools_config")
    if pc and pc.enable_embed:
        names = list(pc.cand_pools_name_to_embed)
        splits.append(("cand_pool", data_config.cand_pool_dir_name, [None] * len(names), names))
    return splits


def generate_embeds_for_config(model, img_preprocess_fn, tokenizer, config):
    uniir_dir, mbeir_data_dir = config.uniir_dir, config.mbeir_data_dir
    embed_config, data_config = config.embed_config, config.data_config
    out_root = os.path.join(uniir_dir, embed_config.embed_dir_name, config.experiment.path_suffix)
    image_size = tuple(map(int, str(data_config.image_size).split(",")))
    main = dist_utils.is_main_process()
    for split, split_dir, dataset_names, pool_names in _splits_to_embed(config):
        for dataset_name, pool_name in zip(dataset_names, pool_names):
            pool_name = pool_name.lower()
            if split == "cand_pool":
                rel = os.path.join(data_config.cand_pool_dir_name, f"mbeir_{pool_name}_{split}.jsonl")
                dataset = MBEIRCandidatePoolDataset(mbeir_data_dir, rel, img_preprocess_fn, print_config=main)
                collator = MBEIRCandidatePoolCollator(tokenizer, image_size)
                mid = pool_name
            else:
                dataset_name = dataset_name.lower()
                dataset = MBEIRMainDataset(
                    mbeir_data_dir, os.path.join(split_dir, f"mbeir_{dataset_name}_{split}.jsonl"),
                    os.path.join(data_config.cand_pool_dir_name, f"mbeir_{pool_name}_cand_pool.jsonl"),
                    data_config.query_instruct_path, img_preprocess_fn, mode=Mode.EVAL,
                    enable_query_instruct=data_config.enable_query_instruct, shuffle_cand=data_config.shuffle_cand,
                    print_config=main)
                collator = MBEIRMainCollator(tokenizer, image_size, mode=Mode.EVAL)
                mid = dataset_name
            sampler = ContiguousDistributedSampler(dataset, dist_utils.get_world_size(), dist_utils.get_rank())
            loader = DataLoader(dataset, batch_size=config.dataloader_config.batch_size,
                                num_workers=config.dataloader_config.num_workers, pin_memory=True, sampler=sampler,
                                shuffle=False, collate_fn=collator, drop_last=False)
            if main:
                print(f"Embedder Log: Generating embeddings for mbeir_{mid}_{split} ({len(dataset)} items)...")
            emb, ids = generate_embeds_and_ids_for_dataset_with_gather(model, loader, config.dist_config.gpu_id,
                                                                       use_fp16=embed_config.use_fp16)
            if main:
                os.makedirs(os.path.join(out_root, split), exist_ok=True)
                np.save(os.path.join(out_root, split, f"mbeir_{mid}_{split}_embed.npy"), emb)
                np.save(os.path.join(out_root, split, f"mbeir_{mid}_{split}_ids.npy"), ids)
                print(f"Embedder Log: Saved {len(ids)} embeddings to {out_root}/{split}/mbeir_{mid}_{split}_embed.npy")
            if dist.is_initialized():
                dist.barrier()
            del emb, ids, loader, dataset
            gc.collect()
        pc = embed_config.get("cand_pools_config")
        if split == "cand_pool" and pc and pc.get("embed_union_pool"):
            if main:   # union pool = concatenation of the per-dataset pools, in config order
                embs = [np.load(os.path.join(out_root, split, f"mbeir_{n.lower()}_{split}_embed.npy")) for n in pool_names]
                idss = [np.load(os.path.join(out_root, split, f"mbeir_{n.lower()}_{split}_ids.npy")) for n in pool_names]
                np.save(os.path.join(out_root, split, f"mbeir_union_{split}_embed.npy"), np.concatenate(embs, axis=0))
                np.save(os.path.join(out_root, split, f"mbeir_union_{split}_ids.npy"), np.concatenate(idss, axis=0))
                print(f"Embedder Log: union pool written ({sum(len(i) for i in idss)} candidates).")
            if dist.is_initialized():
                dist.barrier()


def main(config):
    set_seed(config.seed + dist_utils.get_rank())
    model = build_model_from_config(config)
    model.eval()
    model = model.to(config.dist_config.gpu_id)
    # no DDP wrapper is needed for inference: every rank holds the full weights and embeds its contiguous shard
    generate_embeds_for_config(model, model.get_img_preprocess_fn(), model.get_tokenizer(), config)


def parse_arguments():
    p = argparse.ArgumentParser(description="Generate Embeddings for MBEIR")
    p.add_argument("--uniir_dir", type=str, default="/data/UniIR")
    p.add_argument("--mbeir_data_dir", type=str, default="/data/UniIR/mbeir_data")
    p.add_argument("--config_path", default="config.yaml", help="Path to the config file.")
    return p.parse_args()


if __name__ == "__main__":
    args = parse_arguments()
    config = OmegaConf.load(args.config_path)
    config.uniir_dir, config.mbeir_data_dir = args.uniir_dir, args.mbeir_data_dir
    args.dist_url = config.dist_config.dist_url
    dist_utils.init_distributed_mode(args)
    config.dist_config.gpu_id = args.gpu
    config.dist_config.distributed_mode = args.distributed
    if dist_utils.is_main_process():
        print(OmegaConf.to_yaml(config, sort_keys=False))
    main(config)
    if config.dist_config.distributed_mode:
        dist.destroy_process_group()
