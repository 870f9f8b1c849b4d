"""ctypes loader of oracle/liboracle.so (CPU ORACLE -- test infrastructure, never on the product path)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _LIB = C.CDLL(path)
    return _LIB


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def infonce_scores(q, p, scale):
    q, p = _f32(q), _f32(p)
    b, d = q.shape
    B = p.shape[0]
    out = np.empty((b, B), dtype=np.float32)
    lib().oracle_infonce_scores(q.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p), C.c_float(scale), b, B, d,
                                out.ctypes.data_as(C.c_void_p))
    return out


def infonce_loss(score, toff=0):
    score = _f32(score)
    b, B = score.shape
    loss, acc = C.c_double(), C.c_double()
    lse = np.empty(b, dtype=np.float64)
    lib().oracle_infonce_loss(score.ctypes.data_as(C.c_void_p), b, B, toff, C.byref(loss), C.byref(acc),
                              lse.ctypes.data_as(C.c_void_p))
    return loss.value, acc.value, lse


def topk(pool_f16, ids, queries_f16, k):
    pool = np.ascontiguousarray(pool_f16, dtype=np.float16)
    qs = np.ascontiguousarray(queries_f16, dtype=np.float16)
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    n, d = pool.shape
    nq = qs.shape[0]
    s = np.empty((nq, k), dtype=np.float32)
    i = np.empty((nq, k), dtype=np.int64)
    lib().oracle_topk(pool.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p), C.c_int64(n), d,
                      qs.ctypes.data_as(C.c_void_p), nq, k, s.ctypes.data_as(C.c_void_p), i.ctypes.data_as(C.c_void_p))
    return s, i
