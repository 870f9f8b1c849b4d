import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniir_amd import _lib, ops
lib = _lib.load()
fn = getattr(lib, "_Z18layernorm_fwd_implPKflS0_S0_PvPfiifiS1_")
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_int, C.c_void_p]
dev = "cuda"
for (rows, W) in ((4, 768), (1000, 768), (4, 512), (300, 1024)):
    x = torch.randn(rows, W, device=dev); g = torch.rand(W, device=dev) + 0.5; b = torch.randn(W, device=dev)
    ref = torch.nn.functional.layer_norm(x, (W,), g, b, 1e-5)
    for f16 in (0, 1):
        y = torch.zeros(rows, W, device=dev, dtype=torch.float16 if f16 else torch.bfloat16)
        rc = fn(x.data_ptr(), W, g.data_ptr(), b.data_ptr(), y.data_ptr(), None, rows, W, 1e-5, f16, ops._stream())
        torch.cuda.synchronize()
        print(rows, W, "f16" if f16 else "bf16", rc, float((y.float() - ref).abs().max()))
