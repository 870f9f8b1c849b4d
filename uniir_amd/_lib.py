"""ctypes binding of libuniir_hip.so (C ABI declared in include/uniir_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol is absent this module raises,
and every wrapper raises RuntimeError(uniir_strerror(code)) on a non-zero return code.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UNIIR_HIP_LIB") or os.path.join(_HERE, "libuniir_hip.so")   # env: kernel experiments only

c_void_p, c_int, c_i64, c_float = C.c_void_p, C.c_int32, C.c_int64, C.c_float
ABI_VERSION = 2          # include/uniir_hip.h as of round 6 (uniir_clip_tower.pool_last_block, uniir_reduce_scratch, ...)


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", c_void_p), ("B", c_void_p), ("C", c_void_p), ("C2", c_void_p),
        ("bias", c_void_p), ("resid", c_void_p), ("aux", c_void_p),
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("lda", c_i64), ("ldb", c_i64), ("ldc", c_i64), ("ldaux", c_i64),
        ("a_tmaj", c_int), ("b_tmaj", c_int),
        ("epilogue", c_int), ("act", c_int), ("dtype", c_int),
        ("k_splits", c_int), ("alpha", c_float),
        ("splitk_ws", c_void_p), ("splitk_ws_bytes", c_i64), ("colsum", c_void_p), ("row_scale", c_void_p),
        ("a_rowsum", c_void_p),
    ]


class ClipBlock(C.Structure):
    """uniir_clip_block (include/uniir_hip.h [TOWER]): one residual block's weights (fp32 / bf16 shadow) and gradients"""
    _fields_ = [(n, c_void_p) for n in (
        "ln1_w", "ln1_b", "ln2_w", "ln2_b", "wqkv16", "wo16", "wfc16", "wproj16", "bqkv", "bo", "bfc", "bproj",
        "g_ln1_w", "g_ln1_b", "g_ln2_w", "g_ln2_b", "g_wqkv", "g_bqkv", "g_wo", "g_bo", "g_wfc", "g_bfc", "g_wproj", "g_bproj")]


class ClipTower(C.Structure):
    """uniir_clip_tower"""
    _fields_ = [
        ("is_text", c_int), ("layers", c_int), ("width", c_int), ("heads", c_int), ("tokens", c_int), ("embed_dim", c_int),
        ("resolution", c_int), ("patch", c_int), ("kpad", c_int), ("vocab", c_int),
        ("blocks", C.POINTER(ClipBlock)),
        ("conv16", c_void_p), ("class_emb", c_void_p), ("pos_emb", c_void_p), ("ln_pre_w", c_void_p), ("ln_pre_b", c_void_p),
        ("token_emb", c_void_p), ("ln_post_w", c_void_p), ("ln_post_b", c_void_p), ("proj16", c_void_p),
        ("g_conv", c_void_p), ("g_class", c_void_p), ("g_pos", c_void_p), ("g_ln_pre_w", c_void_p), ("g_ln_pre_b", c_void_p),
        ("g_token", c_void_p), ("g_ln_post_w", c_void_p), ("g_ln_post_b", c_void_p), ("g_proj", c_void_p),
        ("splitk_ws", c_void_p), ("splitk_ws_bytes", c_i64),
        ("stash_act", c_int), ("dtype16", c_int), ("pool_last_block", c_int),
    ]


# name -> (restype, argtypes); P = device pointer, S = stream
P, S = c_void_p, c_void_p
SIGNATURES = {
    "uniir_strerror": (C.c_char_p, [c_int]),
    "uniir_abi_version": (c_int, []),
    "uniir_reduce_scratch": (c_int, [c_void_p, c_i64, c_void_p]),
    "uniir_gemm": (c_int, [C.POINTER(GemmDesc), S]),
    "uniir_gemm_timing": (c_int, [c_int]),
    "uniir_gemm_timing_on": (c_int, [c_int, c_void_p]),
    "uniir_gemm_timing_read": (c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(c_int)]),
    "uniir_gemm_timing_read_ex": (c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(c_int), C.POINTER(c_int),
                                          C.POINTER(c_int)]),
    "uniir_gemm_timing_filter": (c_int, [C.POINTER(c_float), c_int, C.POINTER(c_float), c_int, c_float, C.POINTER(C.c_uint8),
                                         C.POINTER(c_int)]),
    "uniir_layernorm_fwd": (c_int, [P, c_i64, P, P, P, P, c_int, c_int, c_float, S]),
    "uniir_layernorm_bwd": (c_int, [P, c_i64, P, P, c_int, P, P, c_i64, P, P, P, P, c_int, c_int, c_float, S]),
    "uniir_layernorm_bwd_ex": (c_int, [P, c_i64, P, P, c_int, P, P, c_i64, P, P, P, P, P, c_int, c_int, c_float, S]),
    "uniir_attention_fwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, S]),
    "uniir_attention_bwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, S]),
    "uniir_attention_fwd_packed": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, S]),
    "uniir_attention_bwd_packed": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, S]),
    "uniir_attention_fwd_ex": (c_int, [P, c_i64, P, P, c_i64, P, c_i64, P, P, c_int, c_int, c_int, c_int, c_int, c_float,
                                       C.c_uint32, S]),
    "uniir_attention_bwd_ex": (c_int, [P, c_i64, P, P, c_i64, P, P, c_i64, P, P, P, c_i64, P, P, c_i64, c_int, c_int,
                                       c_int, c_int, c_int, c_float, C.c_uint32, S]),
    "uniir_attention_fwd_rows": (c_int, [P, c_i64, P, P, c_i64, P, c_i64, P, P, c_int, P, c_int, c_int, c_int, c_int, c_float,
                                         C.c_uint32, S]),
    "uniir_attention_bwd_rows": (c_int, [P, c_i64, P, P, c_i64, P, P, c_i64, P, P, c_int, P, P, c_i64, P, P, c_i64, c_int, c_int,
                                         c_int, c_int, c_float, C.c_uint32, S]),
    "uniir_patchify": (c_int, [P, P, c_int, c_int, c_int, c_int, S]),
    "uniir_vit_assemble": (c_int, [P, P, P, P, c_int, c_int, c_int, S]),
    "uniir_vit_assemble_bwd": (c_int, [P, P, P, P, c_int, c_int, c_int, S]),
    "uniir_text_embed": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, S]),
    "uniir_text_embed_bwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, S]),
    "uniir_text_embed_packed": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, S]),
    "uniir_text_embed_bwd_packed": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, S]),
    "uniir_gather_rows": (c_int, [P, P, P, c_int, c_int, c_int, S]),
    "uniir_scatter_rows": (c_int, [P, P, P, c_int, c_int, c_int, S]),
    "uniir_act_fwd": (c_int, [P, P, c_i64, c_int, S]),
    "uniir_colsum_bf16": (c_int, [P, c_i64, P, c_int, c_int, S]),
    "uniir_cast_f32_to_bf16": (c_int, [P, P, c_i64, S]),
    "uniir_cast_f32_to_f16": (c_int, [P, P, c_i64, S]),
    "uniir_cast_bf16_to_f32": (c_int, [P, P, c_i64, S]),
    "uniir_cast_pad_rows": (c_int, [P, P, c_int, c_int, c_int, S]),
    "uniir_cast_pad_rows_f16": (c_int, [P, P, c_int, c_int, c_int, S]),
    "uniir_unpad_add": (c_int, [P, P, c_int, c_int, c_int, S]),
    "uniir_fuse_embeddings": (c_int, [P, P, P, P, P, c_int, c_int, S]),
    "uniir_select_normalize": (c_int, [P, P, P, P, c_int, c_int, S]),
    "uniir_select_normalize_bwd": (c_int, [P, P, P, P, P, c_int, c_int, S]),
    "uniir_fuse_embeddings_bwd": (c_int, [P, P, P, P, P, c_int, c_int, S]),
    "uniir_infonce_fwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P, P, P, P, S]),
    "uniir_infonce_bwd": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, P, P, P, P, S]),
    "uniir_hardneg_fwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P, P, P, P, S]),
    "uniir_hardneg_bwd": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P, P, P, P, S]),
    "uniir_sgemm": (c_int, [P, c_i64, c_i64, P, c_i64, c_i64, P, c_i64, c_int, c_int, c_int, c_float, S]),
    "uniir_adamw_step": (c_int, [P, P, P, P, P, c_i64, c_float, c_float, c_float, c_float, c_float, c_int,
                                 c_float, S]),
    "uniir_rmsnorm_fwd": (c_int, [P, c_i64, P, P, P, c_int, c_int, c_float, S]),
    "uniir_rmsnorm_bwd": (c_int, [P, c_i64, P, P, c_int, P, P, c_i64, P, P, c_int, c_int, c_float, S]),
    "uniir_attention_rel_fwd": (c_int, [P, P, P, P, P, c_int, c_float, c_int, c_int, c_int, c_float, C.c_uint32, S]),
    "uniir_attention_rel_bwd": (c_int, [P, P, P, P, P, P, P, c_int, c_float, P, c_int, c_int, c_int, c_float, C.c_uint32, S]),
    "uniir_meanpool_fwd": (c_int, [P, P, c_int, c_int, c_int, S]),
    "uniir_meanpool_bwd": (c_int, [P, P, c_int, c_int, c_int, S]),
    "uniir_dropout_f32": (c_int, [P, P, P, P, c_i64, c_int, c_float, C.c_uint32, P, c_int, S]),
    "uniir_dropout_bf16": (c_int, [P, P, c_i64, c_int, c_i64, c_float, C.c_uint32, P, c_int, S]),
    "uniir_dropout_mask": (c_int, [P, c_i64, c_float, C.c_uint32, S]),
    "uniir_dropout_f32_rows": (c_int, [P, P, P, P, c_i64, c_int, c_float, C.c_uint32, P, S]),
    "uniir_dropout_bf16_rows": (c_int, [P, P, c_i64, c_int, c_i64, c_float, C.c_uint32, P, S]),
    "uniir_tanh_fwd": (c_int, [P, P, c_i64, S]),
    "uniir_tanh_bwd": (c_int, [P, P, P, c_i64, S]),
    "uniir_ema_update": (c_int, [P, P, P, c_i64, c_float, S]),
    "uniir_softce": (c_int, [P, P, P, P, P, c_int, c_int, c_float, c_float, P, P, P, P, P, S]),
    "uniir_sgemm_acc": (c_int, [P, c_i64, c_i64, P, c_i64, c_i64, P, c_i64, c_int, c_int, c_int, c_float, S]),
    "uniir_sgemm_splitk_workspace_bytes": (c_i64, [c_int, c_int, c_int]),
    "uniir_sgemm_splitk": (c_int, [P, c_i64, c_i64, P, c_i64, c_i64, P, c_i64, c_int, c_int, c_int, c_float, c_int, P, c_i64, S]),
    "uniir_pool_inv_norms": (c_int, [P, c_i64, c_int, P, S]),
    "uniir_topk_workspace_bytes": (c_i64, [c_int, c_int, c_i64]),
    "uniir_topk_ncand": (c_int, [c_int, c_int]),
    "uniir_image_workspace_bytes": (c_i64, [c_int, c_int, c_int, c_int, c_int]),
    "uniir_image_preprocess": (c_int, [P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, c_i64, S]),
    "uniir_topk_coarse": (c_int, [P, P, c_i64, c_int, P, c_int, c_int, P, P, P, c_i64, S]),
    "uniir_topk_rescore": (c_int, [P, P, P, c_i64, c_int, P, P, c_int, P, c_int, c_int, P, P, P, S]),
    "uniir_topk_merge": (c_int, [P, P, c_int, c_int, c_int, P, P, S]),
    "uniir_clip_tower_workspace_bytes": (c_i64, [C.POINTER(ClipTower), c_int, c_int]),
    "uniir_clip_tower_fwd": (c_int, [C.POINTER(ClipTower), P, c_int, P, P, c_i64, c_int, S]),
    "uniir_clip_tower_bwd": (c_int, [C.POINTER(ClipTower), P, P, c_int, P, c_i64, S]),
    "uniir_clip_tower_bwd_head": (c_int, [C.POINTER(ClipTower), P, c_int, P, c_i64, S]),
    "uniir_clip_tower_bwd_blocks": (c_int, [C.POINTER(ClipTower), c_int, c_int, c_int, P, c_i64, S]),
    "uniir_clip_tower_bwd_stem": (c_int, [C.POINTER(ClipTower), P, c_int, P, c_i64, S]),
    "uniir_clip_tower_workspace_bytes_packed": (c_i64, [C.POINTER(ClipTower), c_int, c_int, c_int]),
    "uniir_clip_tower_fwd_packed": (c_int, [C.POINTER(ClipTower), P, c_int, P, c_int, P, P, c_i64, c_int, S]),
    "uniir_clip_tower_bwd_head_packed": (c_int, [C.POINTER(ClipTower), P, c_int, P, c_int, P, c_i64, S]),
    "uniir_clip_tower_bwd_blocks_packed": (c_int, [C.POINTER(ClipTower), c_int, P, c_int, c_int, c_int, P, c_i64, S]),
    "uniir_clip_tower_bwd_stem_packed": (c_int, [C.POINTER(ClipTower), P, c_int, P, c_int, P, c_i64, S]),
    "uniir_bias_act_f32": (c_int, [P, P, P, c_i64, c_int, c_int, S]),
    "uniir_vit_assemble_f32": (c_int, [P, P, P, P, c_int, c_int, c_int, S]),
    "uniir_attention_f32_fwd": (c_int, [P, c_i64, P, P, c_i64, P, c_i64, P, c_int, c_int, c_int, c_int, c_int, c_float, S]),
    "uniir_topk_merge_ex": (c_int, [P, P, c_int, c_int, c_int, c_int, P, P, S]),
    "uniir_topk_ip_workspace_bytes": (c_i64, [c_int, c_int, c_i64]),
    "uniir_topk_ip_workspace_bytes_ex": (c_i64, [c_int, c_int, c_i64, c_int]),
    "uniir_topk_set_chunk": (c_int, [c_int]),
    "uniir_topk_ip_sweep_queries": (c_int, [c_int, c_i64]),
    "uniir_topk_ip": (c_int, [P, P, P, c_i64, c_int, P, c_int, c_int, P, P, P, c_i64, S]),
    "uniir_topk_subshard_rows": (c_i64, [c_i64, c_int]),
    "uniir_topk_ip_multi_workspace_bytes": (c_i64, [c_int, c_int, c_i64, c_int]),
    "uniir_topk_ip_multi": (c_int, [P, P, P, c_i64, c_int, P, c_int, c_int, P, P, P, c_i64, S]),
}

_lib = None


def load():
    """Load the library once; raise (never fall back) when it or one of its symbols is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C uniir_amd/csrc). uniir_amd has no CPU/torch fallback for its HIP path."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing -> loud
        fn.restype = res
        fn.argtypes = args
    if lib.uniir_abi_version() < ABI_VERSION:          # a stale build: struct layouts (ClipTower) would not match
        raise RuntimeError(f"{LIB_PATH} is ABI version {lib.uniir_abi_version()}, this binding needs {ABI_VERSION}: rebuild it "
                           "(python -c 'import __graft_entry__ as g; g.build()')")
    _lib = lib
    return lib


def check(code, what=""):
    if code != 0:
        msg = load().uniir_strerror(int(code)).decode()
        raise RuntimeError(f"uniir_hip {what} failed: {msg} (code {code})")
