"""ctypes binding of liblemon_hip.so (include/lemon_hip.h).

The product path has NO CPU fallback: if the library is missing or a call fails, we raise.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "liblemon_hip.so")

METRIC_IP = 0
METRIC_L2 = 1
MAX_K = 64            # per scan pass and in lemon_neighbors
MAX_K_DEEP = 2048     # lemon_index_search (passes of MAX_K, include/lemon_hip.h)
ALGO_AUTO, ALGO_F32_MFMA, ALGO_BF16_FILTER = 0, 1, 2

# every symbol include/lemon_hip.h declares
EXPORTS = [
    "lemon_last_error", "lemon_version", "lemon_normalize_rows", "lemon_paired_distance",
    "lemon_d1_normalized", "lemon_class_confidence", "lemon_paired_metric", "lemon_preprocess_u8", "lemon_preprocess_u8_f16x3t", "lemon_attention_f32", "lemon_attention_set_f16", "lemon_attention_split3", "lemon_layernorm_f32", "lemon_vision_tokens_ln", "lemon_text_tokens", "lemon_linear_f32", "lemon_linear_bf16x6", "lemon_split3_f32", "lemon_layernorm_split3", "lemon_linear_f16x3", "lemon_pack_weight_f16x3t", "lemon_layernorm_f16x3t", "lemon_linear_f16x3t", "lemon_linear_f16x3t_ln", "lemon_linear_f16x3t_chain", "lemon_ln_finalize", "lemon_rowstats_f16x3t", "lemon_unpack_act_f16x3t", "lemon_linear_f16x3t_set_profiling", "lemon_linear_f16x3t_set_mfma", "lemon_linear_f16x3t_profile_read", "lemon_attention_f16x3t", "lemon_split_f16x3", "lemon_layernorm_f16x3", "lemon_attention_f16x3", "lemon_linear_load_tuned",
    "lemon_linear_dump_tuned", "lemon_linear_set_tuning", "lemon_linear_stamp", "lemon_index_create", "lemon_index_free", "lemon_index_add",
    "lemon_index_ntotal", "lemon_index_dim", "lemon_index_data", "lemon_index_search",
    "lemon_index_set_algo", "lemon_index_set_query_dedup", "lemon_index_last_search_info", "lemon_index_set_profiling",
    "lemon_index_profile_read", "lemon_debug_scan_plan", "lemon_neighbors", "lemon_discrepancy", "lemon_score", "lemon_grid_f1",
]


class LemonHipError(RuntimeError):
    pass


class SearchInfo(ctypes.Structure):
    _fields_ = [("algo", ctypes.c_int), ("grid", ctypes.c_int), ("block", ctypes.c_int),
                ("query_panel", ctypes.c_int), ("db_splits", ctypes.c_int),
                ("nq", ctypes.c_int64), ("n", ctypes.c_int64), ("d", ctypes.c_int), ("k", ctypes.c_int),
                ("nq_distinct", ctypes.c_int64)]


_lib = None


def load():
    """Load the HIP library (after torch, so that both share torch's libamdhip64.so.7)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise LemonHipError(
            f"{SO_PATH} is missing: build it with `python -m lemon_amd.build` "
            "(or __graft_entry__.build()). There is no CPU fallback for the LEMoN hot path.")
    try:
        import torch  # noqa: F401  (loads the ROCm runtime the library resolves against)
    except ImportError:
        pass
    lib = ctypes.CDLL(SO_PATH, mode=ctypes.RTLD_GLOBAL)
    c_i64, c_int, vp = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p
    lib.lemon_last_error.restype = ctypes.c_char_p
    lib.lemon_version.argtypes = [ctypes.POINTER(c_int), ctypes.POINTER(c_int)]
    lib.lemon_normalize_rows.argtypes = [vp, c_i64, c_int, vp, vp]
    lib.lemon_paired_distance.argtypes = [c_int, vp, vp, c_i64, c_int, vp, vp]
    lib.lemon_d1_normalized.argtypes = [c_int, vp, c_i64, c_int, vp, c_int, vp, vp, vp]
    lib.lemon_paired_metric.argtypes = [c_int, vp, vp, c_i64, c_int, vp, vp]
    lib.lemon_class_confidence.argtypes = [c_int, vp, c_i64, c_int, vp, c_int, vp, vp, vp]
    lib.lemon_preprocess_u8.argtypes = [vp, c_i64, c_int, c_int, vp, vp, c_int, vp, vp, c_int, c_int, c_int, c_int,
                                        ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), c_int, vp, vp]
    lib.lemon_preprocess_u8_f16x3t.argtypes = [vp, c_i64, c_int, c_int, vp, vp, c_int, vp, vp, c_int, c_int, c_int, c_int,
                                        ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), c_int, vp, vp]
    lib.lemon_grid_f1.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, c_i64, c_int, vp, c_int, ctypes.c_double, c_int, vp, vp, vp, vp]
    lib.lemon_attention_f32.argtypes = [vp, c_i64, c_int, c_int, c_int, c_int, vp, vp]
    lib.lemon_attention_split3.argtypes = [vp, c_i64, c_int, c_int, c_int, c_int, vp, vp]
    lib.lemon_attention_f16x3.argtypes = [vp, c_i64, c_int, c_int, c_int, c_int, vp, vp]
    lib.lemon_layernorm_f32.argtypes = [vp, vp, vp, ctypes.c_float, c_i64, c_int, vp, vp]
    lib.lemon_vision_tokens_ln.argtypes = [vp, vp, vp, vp, vp, ctypes.c_float, c_i64, c_int, c_int, vp, vp]
    lib.lemon_text_tokens.argtypes = [vp, c_i64, vp, vp, c_i64, c_int, c_int, c_int, vp, vp]
    lib.lemon_linear_f32.argtypes = [vp, vp, vp, vp, c_i64, c_int, c_int, ctypes.c_float, c_int, vp, vp]
    lib.lemon_linear_bf16x6.argtypes = [vp, vp, vp, vp, c_i64, c_int, c_int, ctypes.c_float, c_int, vp, vp]
    lib.lemon_split3_f32.argtypes = [vp, c_i64, c_int, c_int, vp, vp]
    lib.lemon_layernorm_split3.argtypes = [vp, vp, vp, ctypes.c_float, c_i64, c_int, vp, vp]
    lib.lemon_linear_f16x3.argtypes = [vp, vp, vp, vp, c_i64, c_int, c_int, ctypes.c_float, c_int, vp, vp]
    lib.lemon_split_f16x3.argtypes = [vp, c_i64, c_int, c_int, ctypes.c_float, vp, vp]
    lib.lemon_pack_weight_f16x3t.argtypes = [vp, c_int, c_int, ctypes.c_float, vp, vp]
    lib.lemon_layernorm_f16x3t.argtypes = [vp, vp, vp, ctypes.c_float, c_i64, c_int, vp, vp]
    lib.lemon_linear_f16x3t.argtypes = [vp, vp, vp, vp, c_i64, c_int, c_int, ctypes.c_float, c_int, c_int, vp, vp]
    lib.lemon_linear_f16x3t_ln.argtypes = [vp, vp, vp, vp, c_i64, c_int, c_int, ctypes.c_float, c_int, c_int, vp, vp, vp, vp, vp, vp]
    lib.lemon_linear_f16x3t_chain.argtypes = [vp, vp, vp, vp, vp, c_i64, c_int, c_int, ctypes.c_float, vp, vp, vp, vp]
    lib.lemon_ln_finalize.argtypes = [vp, c_i64, c_int, ctypes.c_float, vp, vp]
    lib.lemon_rowstats_f16x3t.argtypes = [vp, ctypes.c_float, c_i64, c_int, vp, vp, vp]
    lib.lemon_unpack_act_f16x3t.argtypes = [vp, c_i64, c_int, vp, vp]
    lib.lemon_attention_set_f16.argtypes = [c_int]
    lib.lemon_linear_f16x3t_set_profiling.argtypes = [c_int]
    lib.lemon_linear_f16x3t_set_mfma.argtypes = [c_int]
    lib.lemon_linear_f16x3t_profile_read.argtypes = [ctypes.POINTER(c_i64), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    lib.lemon_attention_f16x3t.argtypes = [vp, c_i64, c_int, c_int, c_int, c_int, vp, vp]
    lib.lemon_layernorm_f16x3.argtypes = [vp, vp, vp, ctypes.c_float, c_i64, c_int, vp, vp]
    lib.lemon_linear_load_tuned.argtypes = [ctypes.c_char_p]
    lib.lemon_linear_dump_tuned.argtypes = [ctypes.c_char_p]
    lib.lemon_linear_set_tuning.argtypes = [c_int]
    lib.lemon_linear_stamp.argtypes = [ctypes.POINTER(c_int), ctypes.c_char_p, c_int]
    lib.lemon_index_create.argtypes = [c_int, c_int, ctypes.POINTER(vp)]
    lib.lemon_index_free.argtypes = [vp]
    lib.lemon_index_add.argtypes = [vp, vp, c_i64, vp]
    lib.lemon_index_ntotal.argtypes = [vp]
    lib.lemon_index_ntotal.restype = c_i64
    lib.lemon_index_dim.argtypes = [vp]
    lib.lemon_index_data.argtypes = [vp]
    lib.lemon_index_data.restype = vp
    lib.lemon_index_search.argtypes = [vp, vp, c_i64, c_int, vp, vp, vp]
    lib.lemon_index_set_algo.argtypes = [vp, c_int]
    lib.lemon_index_set_query_dedup.argtypes = [vp, c_int]
    lib.lemon_index_last_search_info.argtypes = [vp, ctypes.POINTER(SearchInfo)]
    lib.lemon_index_set_profiling.argtypes = [vp, c_int]
    lib.lemon_index_profile_read.argtypes = [vp, ctypes.POINTER(c_i64)] + [ctypes.POINTER(ctypes.c_double)] * 3
    ip = ctypes.POINTER(c_int)
    lib.lemon_debug_scan_plan.argtypes = [c_int, c_int, ip, ip, ip, c_int, ip, ip, c_int, ip]
    lib.lemon_neighbors.argtypes = [vp, vp, vp, vp, vp, c_i64, c_int, c_int, vp, c_int, vp, vp] + [vp] * 9 + [vp]
    lib.lemon_discrepancy.argtypes = [c_int, vp, vp, vp, vp, c_i64, c_int, c_int, vp, vp]
    lib.lemon_score.argtypes = [vp] * 7 + [c_i64, c_int, ctypes.POINTER(ctypes.c_double), vp, vp, vp, vp]
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().lemon_last_error().decode("utf-8", "replace")
        raise LemonHipError(f"{what} failed (rc={rc}): {msg}")
