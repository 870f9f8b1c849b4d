"""dev tool: wall-clock breakdown of one train step (fwd / bwd / optimizer) with syncs, plus allocator statistics."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "uniir_amd", "src"))
import torch
from types import SimpleNamespace
from bench import synth_batch
from models.uniir_clip.clip_scorefusion.clip_sf import CLIPScoreFusion
from uniir_amd.clip_model import CLIP_CONFIGS
from uniir_amd.trainer import NativeTrainer

pairs = int(os.environ.get("PAIRS", "512"))
dev = torch.device("cuda", 0)
cfg = CLIP_CONFIGS["ViT-L/14"]
config = SimpleNamespace(model=SimpleNamespace(gather_embeddings=True), data_config=SimpleNamespace(in_batch_neg_num=0))
model = CLIPScoreFusion("ViT-L/14", device=dev, config=config)
tr = NativeTrainer(model, t_total=1000)
batch = synth_batch(cfg, pairs, 2023, dev)
def sync(): torch.cuda.synchronize()
for it in range(4):
    sync(); t0 = time.perf_counter()
    tr.opt.zero_grad(); model.train()
    out = model(batch); sync(); t1 = time.perf_counter()
    out["loss"].backward(); sync(); t2 = time.perf_counter()
    tr.opt.step(); tr.sched.step(); sync(); t3 = time.perf_counter()
    ms = torch.cuda.memory_stats()
    print(f"it{it}: fwd {1e3*(t1-t0):.1f} ms  bwd {1e3*(t2-t1):.1f} ms  opt {1e3*(t3-t2):.1f} ms  total {1e3*(t3-t0):.1f} | "
          f"reserved {ms['reserved_bytes.all.current']/2**30:.1f} GiB alloc_retries {ms['num_alloc_retries']} "
          f"device_allocs {ms['num_device_alloc']} device_frees {ms['num_device_free']}")
# host-only enqueue time: run a step without syncing and time how long the host takes to enqueue it
sync(); t0 = time.perf_counter()
tr.opt.zero_grad(); out = model(batch); out["loss"].backward(); tr.opt.step()
t1 = time.perf_counter(); sync(); t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0):.1f} ms, then waited {1e3*(t2-t1):.1f} ms for the GPU")
