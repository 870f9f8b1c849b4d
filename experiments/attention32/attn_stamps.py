"""Dev tool: read the s_memtime stamps of a -DA32_STAMP build of the attention kernels (forward, T=257, 1024 items)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uniir_amd import ops, _lib
lib = _lib.load()
b, T, H = int(os.environ.get("MB_ITEMS", "1024")), int(os.environ.get("MB_T", "257")), 16
qkv = torch.randn(b * T, 3 * H * 64, device="cuda").bfloat16()
for _ in range(3):
    out, lse = ops.attention_fwd(qkv, b, T, H, 0)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (64 * 2 * 8))()
fn = C.CDLL(_lib.LIB_PATH).uniir_debug_a32_stamps
assert fn(buf) == 0
import numpy as np
a = np.array(buf, dtype=np.int64).reshape(64, 2, 8)
d = a - a[:, :1, :1]
print("per-WG stamps relative to wave 0 start (ticks), median over 64 WGs; rows: wave 0, wave 7")
print(np.median(d, axis=0).astype(int))
names = ["top", "q issued", "full start", "tail start", "before barrier", "after barrier", "pass1 end", "pieces stored"]
order = [0, 1, 2, 6, 7, 3, 4, 5]
for wv, nm in ((0, "wave0"), (1, "wave7")):
    t = np.median(a[:, wv, :] - a[:, wv, :1], axis=0)
    print(nm, " ".join(f"{names[i]}={int(t[i])}" for i in order))
