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


def resize_geometry(h, w, n_px):
    """torchvision Resize(n_px) + CenterCrop(n_px) on a PIL image (third-party, restated from its published behaviour):
    the short side becomes n_px, the long side int(n_px * long / short) (truncation); the crop offsets are
    int(round((size - n_px) / 2.0)) with Python's round-half-to-even.  -> (oh, ow, top, left)"""
    if w <= h:
        ow, oh = n_px, int(n_px * h / w)
    else:
        oh, ow = n_px, int(n_px * w / h)
    return oh, ow, int(round((oh - n_px) / 2.0)), int(round((ow - n_px) / 2.0))


def resize_bicubic(img_u8, oh, ow):
    img = np.ascontiguousarray(img_u8, dtype=np.uint8)
    h, w, _ = img.shape
    out = np.empty((oh, ow, 3), dtype=np.uint8)
    lib().oracle_resize_bicubic_rgb8(img.ctypes.data_as(C.c_void_p), h, w, out.ctypes.data_as(C.c_void_p), oh, ow)
    return out


def clip_preprocess(img_u8, n_px, mean, std):
    """uint8 [h, w, 3] -> fp32 [3, n_px, n_px]: upstream clip._transform"""
    h, w, _ = img_u8.shape
    oh, ow, top, left = resize_geometry(h, w, n_px)
    r = resize_bicubic(img_u8, oh, ow)
    out = np.empty((3, n_px, n_px), dtype=np.float32)
    m, s = _f32(mean), _f32(std)
    lib().oracle_crop_normalize(r.ctypes.data_as(C.c_void_p), oh, ow, top, left, n_px, m.ctypes.data_as(C.c_void_p),
                                s.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out
