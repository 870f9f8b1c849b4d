"""Dev tool: s_memtime stamps of a -DA32_STAMP build of the 32x32 backward (T=257, 1024 items)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from uniir_amd import ops, _lib
lib = _lib.load()
b, T, H = int(os.environ.get("MB_ITEMS", "1024")), int(os.environ.get("MB_T", "257")), 16
qkv = torch.randn(b * T, 3 * H * 64, device="cuda").bfloat16()
out, lse = ops.attention_fwd(qkv, b, T, H, 0)
do = torch.randn_like(out)
for _ in range(3):
    dqkv = ops.attention_bwd(qkv, out, do, lse, b, T, H, 0)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (64 * 2 * 8))()
assert C.CDLL(_lib.LIB_PATH).uniir_debug_a32_stamps(buf) == 0
a = np.array(buf, dtype=np.int64).reshape(64, 2, 8)
names = ["top", "staged", "barrier", "p1 full", "p1 tail", "restaged", "p2 full", "p2 tail"]
for wv, nm in ((0, "wave0"), (1, "wave3")):
    t = np.median(a[:, wv, :] - a[:, wv, :1], axis=0)
    print(nm, " ".join(f"{names[i]}={int(t[i])}" for i in range(8)))
