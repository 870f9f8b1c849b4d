"""Training entry point for CLIPFeatureFusion on MI355X (drop-in for UniIR
src/models/uniir_clip/clip_featurefusion/train.py: same CLI --config_path/--uniir_dir/--mbeir_data_dir, same YAML keys,
same checkpoint dictionary, one process per GPU under torch.distributed.run).

Mirrors: create_optimizer groups :52-66,200-208 -> NativeAdamW (+ the T5 group); save_checkpoint :64-79; train :97-168; main :171-303.
Differences (result-preserving): no DDP wrapper and no GradScaler -- the gradient all-reduce is one RCCL call over the
flat gradient buffer inside NativeAdamW.step(), and bf16 compute needs no loss scaling; wandb / dotenv logging is not
carried (observability only, SURVEY.md section 2 OUT).
"""
import argparse
import logging
import os
import random

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

import models.uniir_clip.utils as utils
from common.config import OmegaConf
from data.mbeir_dataset import MBEIRMainCollator, MBEIRMainDataset, Mode
from models.uniir_clip.clip_featurefusion.clip_ff import CLIPFeatureFusion
from models.uniir_clip.engine import eval_engine, train_one_epoch
from uniir_amd.host_utils import load_checkpoint_file
from uniir_amd.trainer import CosineLR, NativeAdamW

logger = logging.getLogger()


def set_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def build_dataset(config, tokenizer, img_preprocess_fn, train=True):
    dc = config.data_config
    ds = MBEIRMainDataset(
        mbeir_data_dir=config.mbeir_data_dir,
        query_data_path=dc.train_query_data_path if train else dc.val_query_data_path,
        cand_pool_path=dc.train_cand_pool_path if train else dc.val_cand_pool_path,
        query_instruct_path=dc.query_instruct_path, img_preprocess_fn=img_preprocess_fn, mode=Mode.TRAIN,
        enable_query_instruct=dc.enable_query_instruct, shuffle_cand=dc.shuffle_cand,
        hard_neg_num=dc.hard_neg_num if train else 0, returns=dc.get("returns"), print_config=utils.is_main_process())
    image_size = tuple(map(int, str(dc.image_size).split(",")))
    return ds, MBEIRMainCollator(tokenizer=tokenizer, image_size=image_size, mode=Mode.TRAIN)


def save_checkpoint(model, optimizer, scheduler, epoch, scaler, config):
    ckpt = config.model.ckpt_config
    path = os.path.join(config.uniir_dir, ckpt.ckpt_dir, f"{config.model.short_name.lower()}_epoch_{epoch}.pth")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save({"model": model.state_dict(), "optimizer": optimizer.state_dict(), "scheduler": scheduler.state_dict(),
                "config": config.to_dict(), "epoch": epoch, "scaler": {}}, path)
    print(f"Saved checkpoint to {path}")


def main(config):
    distributed = config.dist_config.distributed_mode
    gpu = config.dist_config.gpu_id
    set_seed(config.seed + utils.get_rank())
    mc = config.model
    model = CLIPFeatureFusion(model_name=mc.clip_vision_model_name,
                            download_root=os.path.join(config.uniir_dir, mc.pretrained_clip_model_dir), config=config)
    model.float()
    start_epoch = 0
    ckpt = mc.ckpt_config
    checkpoint = None
    if ckpt.resume_training:
        path = os.path.join(config.uniir_dir, ckpt.ckpt_dir, ckpt.ckpt_name)
        assert os.path.exists(path), f"Checkpoint file {path} does not exist."
        logger.info(f"loading CLIPFeatureFusion checkpoint from {path}")
        checkpoint = load_checkpoint_file(path)
        model.load_state_dict(checkpoint["model"])
    model.train()
    model = model.to(gpu)
    if distributed:          # the reference wraps in DDP here (train.py:217-218): rank 0's parameters and buffers to every replica
        from uniir_amd import comm
        comm.sync_replicas(model)
    # three groups like clip_featurefusion/train.py:52-66,200-208: CLIP gains/biases (wd 0), CLIP rest (wd 0.2), every T5
    # parameter (wd 0.2, its own learning rate trainer_config.t5_learning_rate)
    optimizer = NativeAdamW(model.clip_model, lr=config.trainer_config.learning_rate, betas=(0.9, 0.98), eps=1.0e-6,
                            weight_decay=0.2,
                            extra=[model.t5_optimizer_group(lr=config.trainer_config.t5_learning_rate, weight_decay=0.2)])
    train_ds, collate = build_dataset(config, model.get_tokenizer(), model.get_img_preprocess_fn(), train=True)
    sampler = DistributedSampler(train_ds, num_replicas=utils.get_world_size(), rank=utils.get_rank(), shuffle=True)
    loader = DataLoader(train_ds, batch_size=config.dataloader_config.train_batch_size,
                        num_workers=config.dataloader_config.num_workers, pin_memory=True, sampler=sampler, shuffle=False,
                        collate_fn=collate, drop_last=True)
    val_loader = None
    if config.evaluator.enable_eval:
        val_ds, val_collate = build_dataset(config, model.get_tokenizer(), model.get_img_preprocess_fn(), train=False)
        val_sampler = DistributedSampler(val_ds, num_replicas=utils.get_world_size(), rank=utils.get_rank(), shuffle=True)
        val_loader = DataLoader(val_ds, batch_size=config.dataloader_config.valid_batch_size,
                                num_workers=config.dataloader_config.num_workers, pin_memory=True, sampler=val_sampler,
                                shuffle=False, collate_fn=val_collate, drop_last=True)
    else:
        print("In-batch validation is disabled.")
    t_total = len(loader) // config.trainer_config.gradient_accumulation_steps * config.trainer_config.num_train_epochs
    scheduler = CosineLR(optimizer, t_total)
    if checkpoint is not None:
        optimizer.load_state_dict({k: (v.to(gpu) if isinstance(v, torch.Tensor) else v)
                                   for k, v in checkpoint["optimizer"].items()})
        scheduler.load_state_dict(checkpoint["scheduler"])
        start_epoch = checkpoint["epoch"] + 1
    if distributed:
        dist.barrier()
    global_step = 0
    for epoch in range(start_epoch, config.trainer_config.num_train_epochs):
        if distributed:
            sampler.set_epoch(epoch)
        stats = train_one_epoch(model, loader, optimizer, epoch, gpu, scheduler, global_step, None, config)
        logger.info({f"train_{k}": v for k, v in stats.items()})
        if val_loader is not None and epoch % config.evaluator.eval_freq == 0:
            vstats = eval_engine(model, val_loader, gpu, config)
            logger.info({f"val_{k}": v for k, v in vstats.items()})
        if utils.is_main_process():
            save_checkpoint(model, optimizer, scheduler, epoch, None, config)
        if distributed:
            dist.barrier()
        torch.cuda.empty_cache()


def parse_arguments():
    p = argparse.ArgumentParser()
    p.add_argument("--config_path", default="config.yaml", help="Path to the config file.")
    p.add_argument("--uniir_dir", type=str, default="/data/UniIR", help="Path to UniIR directory to save checkpoints, embeddings, etc.")
    p.add_argument("--mbeir_data_dir", type=str, default="/data/UniIR/mbeir_data", help="Path to mbeir dataset directory")
    return p.parse_args()


if __name__ == "__main__":
    args = parse_arguments()
    config = OmegaConf.load(args.config_path)
    config.uniir_dir, config.mbeir_data_dir = args.uniir_dir, args.mbeir_data_dir
    args.dist_url = config.dist_config.dist_url
    utils.init_distributed_mode(args)
    config.dist_config.gpu_id = args.gpu
    config.dist_config.distributed_mode = args.distributed
    if utils.is_main_process():
        out_dir = os.path.join(config.uniir_dir, config.logger_config.logger_out_dir)
        os.makedirs(out_dir, exist_ok=True)
        logging.basicConfig(filename=os.path.join(out_dir, config.logger_config.logger_out_file_name), level=logging.INFO,
                            format="%(asctime)s - %(levelname)s - %(message)s")
        logging.info(OmegaConf.to_yaml(config, sort_keys=False))
    main(config)
    if config.dist_config.distributed_mode:
        dist.destroy_process_group()
