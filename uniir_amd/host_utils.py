"""Host-side bookkeeping shared by the reference-shaped trees under uniir_amd/src (one implementation instead of the
reference's three copies): rank / world helpers and init_distributed_mode (reference src/common/dist_utils.py:62-91,
src/models/uniir_{clip,blip}/utils.py:233-306), ContiguousDistributedSampler (dist_utils.py:94-115), SmoothedValue /
MetricLogger (utils.py:44-200) and the epoch loops of src/models/uniir_{clip,blip}/engine.py.  No arithmetic of the hot
path lives here."""
import datetime
import math
import os
import time
from collections import defaultdict, deque
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




# ------------------------------------------------------------------------------------------------------------
# epoch loops (reference src/models/uniir_clip/engine.py:7-84 and src/models/uniir_blip/engine.py:9-114): what the
# two engines share; `step_fn(model, batch, i, n_batches)` is the only model-specific part (BLIP passes alpha)
# ------------------------------------------------------------------------------------------------------------
def load_checkpoint_file(path):
    """torch.load for UniIR checkpoints ({"model", "optimizer", "scheduler", "config", "epoch", "scaler"}).  The published
    files pickle their `config` as an omegaconf DictConfig; torch >= 2.6 refuses such globals by default and omegaconf is
    not a dependency here (common/config.py replaces it), so: plain tensors-only load first; only when that is refused
    for an unknown global, a second read with an unpickler that resolves torch / numpy / collections globals and turns
    every other class into an inert placeholder (the weights are what is read).  Other failures propagate.
    The second reader resolves an exact allow-list of (module, name) pairs only (a root-module rule let
    torch.utils.collect_env.run through)."""
    import pickle
    import warnings
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as e:      # a global the tensors-only loader refuses (the pickled config object)
        warnings.warn(f"{path}: tensors-only load refused ({str(e).splitlines()[0][:120]}); re-reading with the placeholder "
                      "unpickler (only tensor / storage / container globals are resolved)")

    class _Placeholder(dict):          # accepts whatever the pickle stream does to the object it stands for
        def __init__(self, *a, **k):
            dict.__init__(self)

        def __setstate__(self, state):
            self["_state"] = state

        def append(self, item):
            self.setdefault("_items", []).append(item)

        def extend(self, items):
            self.setdefault("_items", []).extend(items)

    # EXACT (module, name) pairs -- not module roots: torch / numpy themselves ship callables that run shell commands or compile
    # code (torch.utils.collect_env.run, torch.utils.cpp_extension.load_inline, numpy.testing._private.utils.runstring ...),
    # so "anything under torch.*" is not a safe rule.  These are what tensors, storages and plain containers reduce to.
    storages = tuple(f"{t}Storage" for t in ("Float", "Half", "BFloat16", "Double", "Long", "Int", "Short", "Char", "Byte", "Bool",
                                              "ComplexFloat", "ComplexDouble"))
    dtypes = ("float32", "float16", "bfloat16", "float64", "int64", "int32", "int16", "int8", "uint8", "bool", "complex64",
              "complex128", "float", "half", "double", "long", "int", "short")
    allowed = {("torch._utils", n) for n in ("_rebuild_tensor_v2", "_rebuild_tensor", "_rebuild_parameter",
                                             "_rebuild_parameter_with_state")}
    allowed |= {("torch", n) for n in storages + dtypes + ("Size", "device", "Tensor")}
    allowed |= {("torch.storage", "UntypedStorage"), ("torch.storage", "TypedStorage"), ("torch.nn.parameter", "Parameter"),
                ("collections", "OrderedDict"), ("collections", "defaultdict"), ("_codecs", "encode")}
    allowed |= {(m, n) for m in ("numpy.core.multiarray", "numpy._core.multiarray") for n in ("_reconstruct", "scalar")}
    allowed |= {("numpy", "ndarray"), ("numpy", "dtype")}
    allowed |= {("builtins", n) for n in ("set", "frozenset", "list", "dict", "tuple", "int", "float", "bool", "str", "bytes",
                                          "bytearray", "complex", "slice", "range", "object")}

    class _Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            # every global outside the list (config classes, and anything a hostile file might name: os.system,
            # torch.utils.collect_env.run ...) becomes an inert placeholder class instead of being imported
            if (module, name) in allowed:
                try:
                    return super().find_class(module, name)
                except Exception:      # noqa: BLE001
                    pass
            return type(name, (_Placeholder,), {"__module__": module})

    class _Module:
        Unpickler = _Unpickler
        load = staticmethod(lambda f, **kw: _Unpickler(f, **kw).load())
        __name__ = "pickle"

    return torch.load(path, map_location="cpu", weights_only=False, pickle_module=_Module)


def batch_to_device(batch, gpu_id):
    for key, value in batch.items():
        if isinstance(value, torch.Tensor):
            batch[key] = value.to(gpu_id, non_blocking=True)
            if key in ("txt_mask_batched", "image_mask_batched") and not value.is_cuda and batch[key] is not value:
                # the modality masks stay readable on the host (as DevicePrefetcher leaves them): CLIP_SF sizes each tower's launch
                # from the live rows and would otherwise read the mask back from the device
                batch[key]._uniir_host = value
                batch[key]._uniir_host_version = batch[key]._version
        elif hasattr(value, "to_device"):                                   # clip_front.RawImageBatch: transform on the GPU
            batch[key] = value.to_device(torch.device("cuda", gpu_id) if isinstance(gpu_id, int) else torch.device(gpu_id))
        elif hasattr(value, "input_ids") and hasattr(value, "items"):      # transformers BatchEncoding (BLIP tokenizer)
            for k, v in value.items():
                value[k] = v.to(gpu_id)
    return batch


class DevicePrefetcher:
    """Overlaps the host -> device copy of batch i + 1 with the compute of batch i (the
    reference issues `.to(device, non_blocking=True)` on the compute stream right before the step, and its per-step
    `.item()` has already drained the GPU, so the 0.6 GB of a 512-pair batch travels while the GPU idles).  Tensors are
    pinned (if the loader did not) and copied on a side stream; `__next__` makes the compute stream wait for the copy
    event and hands the tensors over with record_stream.  Same batches, same order."""

    def __init__(self, loader, gpu_id):
        self.loader, self.dev = loader, torch.device("cuda", gpu_id) if isinstance(gpu_id, int) else torch.device(gpu_id)
        self.stream = torch.cuda.Stream(self.dev)

    def __len__(self):
        return len(self.loader)

    def _move(self, t):
        if not t.is_cuda and not t.is_pinned():
            t = t.pin_memory()
        return t.to(self.dev, non_blocking=True)

    def _launch(self, batch):
        with torch.cuda.stream(self.stream):
            for key, value in batch.items():
                if isinstance(value, torch.Tensor):
                    lens = None
                    if key == "txt_batched" and not value.is_cuda and value.dim() == 2 and not value.is_floating_point():
                        # CLIP token ids still on the host: the live length of every caption (EOT = arg-max id, upstream's
                        # pooling row) travels with the batch, so that the packed text tower needs no device -> host read
                        lens = (value.argmax(dim=-1) + 1).to(torch.int32)
                    host_mask = value if (key in ("txt_mask_batched", "image_mask_batched") and not value.is_cuda) else None
                    batch[key] = self._move(value)
                    if lens is not None:
                        batch[key]._uniir_lens = lens
                        batch[key]._uniir_lens_version = batch[key]._version     # the hint dies with an in-place edit
                    if host_mask is not None:
                        # the modality masks stay readable on the host: CLIP_SF runs each tower on its live rows only
                        # (clip_sf.encode_multimodal_input) and needs their number to size the launch
                        batch[key]._uniir_host = host_mask
                        batch[key]._uniir_host_version = batch[key]._version     # dies with an in-place edit of the device mask
                elif hasattr(value, "to_device"):                                   # clip_front.RawImageBatch
                    batch[key] = value.pin_memory().to_device(self.dev)
                elif hasattr(value, "input_ids") and hasattr(value, "items"):      # transformers BatchEncoding (BLIP)
                    for k, v in value.items():
                        value[k] = self._move(v)
                        if k == "attention_mask" and not v.is_cuda and v.dim() == 2:
                            # the captions' valid lengths stay readable on the host: BLIP's BERT runs on the rows up to them only
                            # (blip_model.TextPack) and sizes its launches from their sum
                            value[k]._uniir_lens = v.sum(1).to(torch.int32)
                            value[k]._uniir_lens_version = value[k]._version
            done = torch.cuda.Event()
            done.record(self.stream)
        return batch, done

    @staticmethod
    def _tensors(batch):
        for value in batch.values():
            if isinstance(value, torch.Tensor):
                yield value
            elif hasattr(value, "input_ids") and hasattr(value, "items"):
                yield from value.values()

    def __iter__(self):
        it = iter(self.loader)
        nxt = None
        try:
            nxt = self._launch(next(it))
        except StopIteration:
            return
        while nxt is not None:
            batch, done = nxt
            try:
                nxt = self._launch(next(it))          # the next copy is in flight while the caller computes on `batch`
            except StopIteration:
                nxt = None
            cur = torch.cuda.current_stream(self.dev)
            cur.wait_event(done)
            for t in self._tensors(batch):
                if t.is_cuda:
                    t.record_stream(cur)
            yield batch


def run_train_epoch(model, data_loader, optimizer, scheduler, config, gpu_id, epoch, step_fn):
    """forward, loss / accumulation_steps, backward, optimizer + scheduler step every accumulation_steps micro-batches;
    the logged lr is read after scheduler.step() and the logged loss is un-scaled, like the reference"""
    model.train()
    log = MetricLogger(delimiter="  ")
    log.add_meter("lr", SmoothedValue(window_size=1, fmt="{value:.6f}"))
    log.add_meter("loss", SmoothedValue(window_size=1, fmt="{value:.4f}"))
    log.add_meter("inbatch_accuracy", SmoothedValue(window_size=1, fmt="{value:.4f}"))
    accum = config.trainer_config.gradient_accumulation_steps
    pending, n = 0, len(data_loader)
    feed = DevicePrefetcher(data_loader, gpu_id)
    for i, batch in enumerate(log.log_every(feed, config.trainer_config.print_freq, f"Train Epoch: [{epoch}]")):
        outputs = step_fn(model, batch, i, n)
        scaled = outputs["loss"] / accum
        if hasattr(optimizer, "arm_overlap"):          # overlapped gradient all-reduce only on the window's last backward
            optimizer.arm_overlap(pending == accum - 1)
        scaled.backward()
        pending += 1
        if pending == accum:
            optimizer.step()
            model.zero_grad()
            scheduler.step()
            pending = 0
        log.update(loss=scaled.item() * accum)                 # host sync, as in the reference's loop
        log.update(lr=optimizer.param_groups[0]["lr"])
        log.update(inbatch_accuracy=outputs["accuracy"].item())
    log.synchronize_between_processes()
    print("Averaged stats:", log.global_avg())
    return {k: meter.global_avg for k, meter in log.meters.items()}


@torch.no_grad()
def run_eval_epoch(model, data_loader, config, gpu_id, step_fn):
    model.eval()
    log = MetricLogger(delimiter="  ")
    log.add_meter("loss", SmoothedValue(window_size=1, fmt="{value:.4f}"))
    log.add_meter("inbatch_accuracy", SmoothedValue(window_size=1, fmt="{value:.4f}"))
    n = len(data_loader)
    for i, batch in enumerate(log.log_every(DevicePrefetcher(data_loader, gpu_id), config.evaluator.print_freq, "Test:")):
        outputs = step_fn(model, batch, i, n)
        log.update(loss=outputs["loss"].item())
        log.update(inbatch_accuracy=outputs["accuracy"].item())
    log.synchronize_between_processes()
    print("Averaged stats:", log.global_avg())
    return {k: meter.global_avg for k, meter in log.meters.items()}
