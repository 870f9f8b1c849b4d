"""Training-loop utilities (drop-in for UniIR src/models/uniir_clip/utils.py: SmoothedValue :44-104, MetricLogger
:107-200, distributed helpers :203-306).  Host-side bookkeeping only."""
import datetime
import os
import time
from collections import defaultdict, deque

import torch
import torch.distributed as dist


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


class SmoothedValue(object):
    """windowed series (median / avg / max / last value) plus the global running average"""

    def __init__(self, window_size=20, fmt=None):
        self.deque = deque(maxlen=window_size)
        self.total, self.count = 0.0, 0
        self.fmt = fmt or "{median:.4f} ({global_avg:.4f})"

    def update(self, value, n=1):
        self.deque.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self):
        """sums count / total over ranks (one float64[2] all-reduce per meter per epoch); the window is local"""
        if not is_dist_avail_and_initialized():
            return
        dev = "cuda" if torch.cuda.is_available() and dist.get_backend() != "gloo" else "cpu"
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device=dev)
        dist.barrier()
        dist.all_reduce(t)
        self.count, self.total = int(t[0].item()), t[1].item()

    @property
    def median(self):
        return torch.tensor(list(self.deque)).median().item()

    @property
    def avg(self):
        return torch.tensor(list(self.deque), dtype=torch.float32).mean().item()

    @property
    def global_avg(self):
        return self.total / self.count

    @property
    def max(self):
        return max(self.deque)

    @property
    def value(self):
        return self.deque[-1]

    def __str__(self):
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max, value=self.value)


class MetricLogger(object):
    def __init__(self, delimiter="\t"):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def update(self, **kwargs):
        for k, v in kwargs.items():
            if isinstance(v, torch.Tensor):
                v = v.item()
            assert isinstance(v, (float, int))
            self.meters[k].update(v)

    def __getattr__(self, attr):
        if attr in self.__dict__.get("meters", {}):
            return self.meters[attr]
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{attr}'")

    def __str__(self):
        return self.delimiter.join(f"{n}: {m}" for n, m in self.meters.items())

    def global_avg(self):
        return self.delimiter.join(f"{n}: {m.global_avg:.4f}" for n, m in self.meters.items())

    def synchronize_between_processes(self):
        for m in self.meters.values():
            m.synchronize_between_processes()

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def log_every(self, iterable, print_freq, header=None):
        header = header or ""
        start = end = time.time()
        iter_time, data_time = SmoothedValue(fmt="{avg:.4f}"), SmoothedValue(fmt="{avg:.4f}")
        n = len(iterable)
        width = len(str(n))
        for i, obj in enumerate(iterable):
            data_time.update(time.time() - end)
            yield obj
            iter_time.update(time.time() - end)
            if i % print_freq == 0 or i == n - 1:
                eta = str(datetime.timedelta(seconds=int(iter_time.global_avg * (n - i))))
                msg = [header, f"[{i:>{width}}/{n}]", f"eta: {eta}", str(self), f"time: {iter_time}", f"data: {data_time}"]
                if torch.cuda.is_available():
                    msg.append(f"max mem: {torch.cuda.max_memory_allocated() / (1024.0 * 1024.0):.0f}")
                print(self.delimiter.join(msg))
            end = time.time()
        total = time.time() - start
        print(f"{header} Total time: {datetime.timedelta(seconds=int(total))} ({total / max(1, n):.4f} s / it)")


def setup_for_distributed(is_master):
    import builtins
    builtin_print = builtins.print

    def quiet_print(*args, **kwargs):
        if is_master or kwargs.pop("force", False):
            builtin_print(*args, **kwargs)

    builtins.print = quiet_print


def init_distributed_mode(args):
    """env:// rendezvous from RANK / WORLD_SIZE / LOCAL_RANK (or SLURM_PROCID); silences print on non-zero ranks"""
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        args.rank, args.world_size = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
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
    if torch.cuda.is_available():
        torch.cuda.set_device(args.gpu)
    args.dist_backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" == RCCL on ROCm
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    print(f"| distributed init (rank {args.rank}): {args.dist_url}", flush=True)
    dist.init_process_group(backend=args.dist_backend, init_method=args.dist_url, world_size=args.world_size, rank=args.rank)
    dist.barrier()
    setup_for_distributed(args.rank == 0)
