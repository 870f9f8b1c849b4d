"""Distributed helpers for embedding extraction (drop-in for UniIR src/common/dist_utils.py: init_distributed_mode
:62-91, ContiguousDistributedSampler :94-115, rank/world helpers).  One process per GPU; backend "nccl" is RCCL on
ROCm (xGMI within the node)."""
import math
import os
from datetime import timedelta

import torch
import torch.distributed as dist
from torch.utils.data import Sampler


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def setup_for_distributed(is_master):
    import builtins
    builtin_print = builtins.print

    def quiet_print(*args, **kwargs):
        if is_master or kwargs.pop("force", False):
            builtin_print(*args, **kwargs)

    builtins.print = quiet_print


def init_distributed_mode(args):
    """reads RANK / WORLD_SIZE / LOCAL_RANK (torch.distributed.run) or SLURM_PROCID; sets args.rank/.gpu/.distributed"""
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        args.rank = int(os.environ["RANK"])
        args.world_size = int(os.environ["WORLD_SIZE"])
        args.gpu = int(os.environ.get("LOCAL_RANK", 0))
    elif "SLURM_PROCID" in os.environ:
        args.rank = int(os.environ["SLURM_PROCID"])
        args.gpu = args.rank % max(1, torch.cuda.device_count())
        args.world_size = int(os.environ.get("SLURM_NTASKS", 1))
    else:
        print("Not using distributed mode")
        args.distributed, args.gpu, args.rank, args.world_size = False, 0, 0, 1
        return
    args.distributed = True
    backend = "nccl" if torch.cuda.is_available() else "gloo"
    if torch.cuda.is_available():
        torch.cuda.set_device(args.gpu)
    args.dist_backend = backend
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    print(f"| distributed init (rank {args.rank}): {getattr(args, 'dist_url', 'env://')}", flush=True)
    dist.init_process_group(backend=backend, init_method=getattr(args, "dist_url", "env://"), world_size=args.world_size,
                            rank=args.rank, timeout=timedelta(minutes=60))
    dist.barrier()


class ContiguousDistributedSampler(Sampler):
    """rank r iterates [r*ceil(n/W), min((r+1)*ceil(n/W), n)) in order, no padding (ragged and empty shards allowed)"""

    def __init__(self, dataset, num_replicas=None, rank=None):
        self.dataset = dataset
        self.num_replicas = num_replicas if num_replicas is not None else get_world_size()
        self.rank = rank if rank is not None else get_rank()
        self.epoch = 0
        self.num_samples_per_replica = math.ceil(len(dataset) / self.num_replicas)
        self.total_size = self.num_samples_per_replica * self.num_replicas

    def __iter__(self):
        n = len(self.dataset)
        lo = min(self.rank * self.num_samples_per_replica, n)
        return iter(range(lo, min(lo + self.num_samples_per_replica, n)))

    def __len__(self):
        return self.num_samples_per_replica

    def set_epoch(self, epoch):
        self.epoch = epoch
