"""s_memtime timeline of the persistent pair-tile attention kernels (EXP build): workgroups 64..127, their third head."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from uniir_amd import _lib, ops

dev = "cuda"
raw = C.CDLL(_lib.LIB_PATH)
setf = raw.uniir_exp_attn_set
setf.argtypes = [C.c_void_p, C.c_int]
setf.restype = None
T, H, b = int(os.environ.get("T", 257)), 16, 1024
qkv = torch.randn(b * T, 3 * H * 64, device=dev).bfloat16()
out, lse = ops.attention_fwd(qkv, b, T, H, 0)
do = torch.randn_like(out)
dqkv = torch.empty_like(qkv)
stamps = torch.zeros(64 * 8 * 16, dtype=torch.int64, device=dev)
names = {"bwd": ["top", "dma KV issued", "ph1 odd done", "ph1 pair done", "vm0 (KV landed)", "stores+stats req", "B1 passed", "stats+dma QdO",
                 "ph2 odd done", "ph2 pair done", "vm0 (QdO landed)", "stores+kf req", "B2 passed"],
         "fwd": ["top", "dma issued", "odd done", "pair done", "vm0", "stores+q req", "B passed"]}
for nm, fn in (("bwd", lambda: ops.attention_bwd(qkv, out, do, lse, b, T, H, 0, dqkv=dqkv)),
               ("fwd", lambda: ops.attention_fwd(qkv, b, T, H, 0, out=out, lse=lse))):
    ns = len(names[nm])
    for _ in range(3):
        fn()
    setf(stamps.data_ptr(), 0)
    stamps.zero_()
    fn()
    torch.cuda.synchronize()
    setf(None, 0)
    s = stamps.cpu().numpy().reshape(64, 8, 16)[:, :, :ns].astype(np.int64)
    rel = s - s[:, :, 0].min(axis=1)[:, None, None]
    med = np.median(rel, axis=0)
    print(f"{nm} T={T}: median ticks since the workgroup's first stamp of its third head (64 workgroups), per wave")
    print("   " + " | ".join(f"{i}:{n}" for i, n in enumerate(names[nm])))
    for w in range(8):
        print(f"  wave {w}: " + " ".join(f"{int(x):7d}" for x in med[w]))
    life = rel.max(axis=(1, 2))
    print(f"  head period: median {int(np.median(life))} min {int(life.min())} max {int(life.max())}")
