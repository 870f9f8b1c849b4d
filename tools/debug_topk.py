import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
from oracle import c_oracle
from uniir_amd import retrieval, ops
rng = np.random.default_rng(0)
n, d = 1000, 64
pool = rng.standard_normal((n, d)).astype(np.float16)
inv_o = np.empty(n, dtype=np.float32)
c_oracle.lib().oracle_inv_norms(pool.ctypes.data_as(C.c_void_p), C.c_int64(n), d, inv_o.ctypes.data_as(C.c_void_p))
sh = retrieval.PoolShard(torch.tensor(pool, device="cuda"), torch.arange(n, device="cuda"))
inv_g = sh.inv_norm.cpu().numpy()
print("inv mismatch count", (inv_o != inv_g).sum(), "max rel", np.abs(inv_o-inv_g).max()/inv_o.max())
# numpy emulation of sequential sum
p32 = pool.astype(np.float32)
s = np.zeros(n, dtype=np.float32)
for j in range(d):
    s = (s + p32[:, j] * p32[:, j]).astype(np.float32)
inv_n = (np.float32(1.0) / np.sqrt(s)).astype(np.float32)
print("numpy vs oracle", (inv_n != inv_o).sum(), "numpy vs gpu", (inv_n != inv_g).sum())
