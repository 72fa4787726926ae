"""ctypes binding of libzett_hip.so (C ABI: include/zett_hip.h).

The library is the product: if it cannot be loaded there is no fallback — the
import of anything that computes raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

from .build import LIB_PATH

ZETT_OK = 0
E_INVALID, E_HIP, E_STATE, E_INDEX, E_NOT_IMPLEMENTED, E_KEY, E_RANGE = -1, -2, -3, -4, -5, -6, -7
RANGE_SOURCE, RANGE_ACTIVATION, RANGE_OUTPUT, RANGE_WEIGHT = 1, 2, 4, 8      # zett_range_bits
DTYPE_F32, DTYPE_F16, DTYPE_BF16 = 0, 1, 2
PREC_BF16, PREC_F32, PREC_F16 = 0, 1, 2
RETOK_BPE, RETOK_UNIGRAM, RETOK_WORDPIECE = 0, 1, 2
OUT_IN, OUT_BIAS = 0, 1              # zett_output

ABI_SYMBOLS = (
    "zett_last_error", "zett_abi_version", "zett_create", "zett_destroy", "zett_load_weight",
    "zett_finalize", "zett_forward", "zett_get_stats", "zett_workspace_bytes", "zett_set_option",
    "zett_retok_create", "zett_retok_destroy", "zett_retokenize", "zett_check_range", "zett_get_gemm_log",
    "zett_stream_wait_output", "zett_forward_prepare", "zett_retokenize_async", "zett_retok_result", "zett_retok_set_option",
    "zett_partition_rows", "zett_partition_workspace_bytes", "zett_scatter_rows",
    "zett_table_plan", "zett_table_rows", "zett_forward_table",
    # training primitives (zett_amd/autograd.py)
    "zett_op_gemm_f32", "zett_op_transpose_f32", "zett_op_colsum_f32", "zett_op_elementwise_f32", "zett_op_rowdot_f32",
    "zett_op_layernorm_fwd_f32", "zett_op_layernorm_bwd_f32", "zett_op_gelu_fwd_f32", "zett_op_gelu_bwd_f32",
    "zett_op_attention_fwd_f32", "zett_op_attention_bwd_f32", "zett_op_gather_fwd_f32", "zett_op_gather_bwd_f32",
    "zett_op_gather_rows_f32", "zett_op_scatter_add_rows_f32", "zett_op_gemm_lo", "zett_op_convert_lo", "zett_op_transpose_lo", "zett_op_grad_operands_lo", "zett_op_transpose_lo16", "zett_op_gelu_fwd_lo",
)


class ZettConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_embd", "n_in_embd", "hidden", "intermediate", "heads", "layers", "n_extra",
        "original_vocab_size", "pad_token_id", "separate_out", "single_head", "rescale",
        "predict_bias", "embed_lang", "n_langs", "max_positions")] + [
        ("ln_eps_encoder", C.c_float), ("ln_eps_projector", C.c_float)]


ABI_VERSION = 8      # ZETT_ABI_VERSION of include/zett_hip.h


class ZettStats(C.Structure):
    _fields_ = [("rows", C.c_int64), ("packed_tokens", C.c_int64), ("distinct_ids", C.c_int64),
                ("chunks", C.c_int64), ("executed_flops", C.c_double), ("gemm_ms", C.c_double),
                ("gemm_launches", C.c_int64), ("gemm_flops_timed", C.c_double), ("distinct_positions", C.c_int64)]


class ZettGemmRecord(C.Structure):
    _fields_ = [("m", C.c_int32), ("n", C.c_int32), ("k", C.c_int32), ("variant", C.c_int32), ("epilogue", C.c_int32),
                ("ms", C.c_float), ("flops", C.c_double), ("bytes", C.c_double)]


class ZettRetokModel(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("n_pieces", C.c_int32),
        ("piece_bytes", C.c_void_p), ("piece_offsets", C.c_void_p), ("piece_ids", C.c_void_p),
        ("piece_scores", C.c_void_p), ("unigram_min_score", C.c_double),
        ("n_merges", C.c_int32), ("merges", C.c_void_p),
        ("unk_id", C.c_int32), ("fuse_unk", C.c_int32), ("byte_fallback", C.c_int32),
        ("byte_fallback_ids", C.c_void_p), ("ignore_merges", C.c_int32),
        ("n_special", C.c_int32), ("special_bytes", C.c_void_p), ("special_offsets", C.c_void_p),
        ("special_ids", C.c_void_p),
        ("piece_continuing", C.c_void_p), ("max_input_chars_per_word", C.c_int32),
    ]


_lib = None
_lock = threading.Lock()


def lib_path() -> str:
    return os.environ.get("ZETT_HIP_LIB", LIB_PATH)


def load():
    """dlopen the library (once) and declare the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: the HIP extension has not been built. Run "
                "`python -m zett_amd.build` (needs hipcc); zett_amd has no CPU fallback.")
        lib = C.CDLL(path)
        lib.zett_last_error.restype = C.c_char_p
        lib.zett_abi_version.restype = C.c_int
        lib.zett_create.argtypes = [C.POINTER(ZettConfig), C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        lib.zett_destroy.argtypes = [C.c_void_p]
        lib.zett_load_weight.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_int]
        lib.zett_finalize.argtypes = [C.c_void_p]
        lib.zett_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int, C.c_int64,
                                     C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.zett_get_stats.argtypes = [C.c_void_p, C.POINTER(ZettStats)]
        lib.zett_workspace_bytes.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_int64)]
        lib.zett_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        lib.zett_check_range.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
        lib.zett_stream_wait_output.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.zett_get_gemm_log.argtypes = [C.c_void_p, C.POINTER(ZettGemmRecord), C.c_int64, C.POINTER(C.c_int64)]
        lib.zett_retok_create.argtypes = [C.POINTER(ZettRetokModel), C.c_int, C.POINTER(C.c_void_p)]
        lib.zett_retok_destroy.argtypes = [C.c_void_p]
        lib.zett_retok_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        lib.zett_retokenize.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                        C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p]
        lib.zett_retokenize_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                              C.c_void_p, C.c_void_p]
        lib.zett_retok_result.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        lib.zett_forward_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
        lib.zett_table_plan.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p]
        lib.zett_table_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.zett_forward_table.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.zett_partition_workspace_bytes.argtypes = [C.c_int64, C.c_int32, C.POINTER(C.c_int64)]
        lib.zett_scatter_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]
        lib.zett_partition_rows.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32),
                                            C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
        P, I32, I64, F = C.c_void_p, C.c_int32, C.c_int64, C.c_float
        lib.zett_op_gemm_f32.argtypes = [P, I32, P, I32, I64, I32, I32, P, I32, P, I32, P, I32, P]
        lib.zett_op_transpose_f32.argtypes = [P, I32, P, I32, I64, I32, I64, P]
        lib.zett_op_gemm_lo.argtypes = [I32, P, I32, P, I32, I64, I32, I32, P, I32, P, I32, P, I32, P]
        lib.zett_op_convert_lo.argtypes = [I32, P, I32, P, I32, I64, I32, I32, P]
        lib.zett_op_transpose_lo.argtypes = [I32, P, I32, P, I32, I64, I32, I64, P]
        lib.zett_op_grad_operands_lo.argtypes = [I32, P, I32, P, I32, I32, I64, I32, I64, P, I32, P, I32, P, P]
        lib.zett_op_transpose_lo16.argtypes = [I32, P, I32, P, I32, I64, I32, I64, P]
        lib.zett_op_gelu_fwd_lo.argtypes = [I32, P, P, I64, I32, P]
        lib.zett_op_colsum_f32.argtypes = [P, I32, I64, I32, P, I32, P]
        lib.zett_op_elementwise_f32.argtypes = [I32, P, P, P, P, P, P, I64, I32, P]
        lib.zett_op_rowdot_f32.argtypes = [P, I32, P, P, P, I64, I32, P]
        lib.zett_op_layernorm_fwd_f32.argtypes = [P, I32, P, P, F, P, P, I64, I32, P, I32, P]
        lib.zett_op_layernorm_bwd_f32.argtypes = [P, P, P, I32, P, P, P, P, I32, I64, I32, P]
        lib.zett_op_gelu_fwd_f32.argtypes = [P, P, I64, I32, P]
        lib.zett_op_gelu_bwd_f32.argtypes = [P, P, P, I64, I32, P]
        lib.zett_op_attention_fwd_f32.argtypes = [P, I32, P, P, I32, P, P, I64, I32, I32, I32, I32, P, I32, P, P, I32, P]
        lib.zett_op_attention_bwd_f32.argtypes = [P, I32, P, I32, P, P, I32, P, P, I64, I32, I32, I32, I32, P, I32, P, P, I32, P]
        lib.zett_op_gather_rows_f32.argtypes = [P, P, I32, P, P, I64, I32, P]
        lib.zett_op_scatter_add_rows_f32.argtypes = [P, I32, P, P, I64, I32, P]
        lib.zett_op_gather_fwd_f32.argtypes = [P, I64, P, I32, I32, I32, P, P, P, P, P]
        lib.zett_op_gather_bwd_f32.argtypes = [P, I64, P, I32, I32, I32, P, P, P, P, P]
        for name in ABI_SYMBOLS:
            fn = getattr(lib, name)
            if name != "zett_last_error":
                fn.restype = C.c_int
        if lib.zett_abi_version() != ABI_VERSION:      # struct layouts below are those of this version
            raise RuntimeError(f"{path} has ABI version {lib.zett_abi_version()}, this package expects {ABI_VERSION}: "
                               "rebuild it (`python -m zett_amd.build`)")
        _lib = lib
    return _lib


class RangeError(OverflowError):
    """ZETT_E_RANGE: a value left the range of the 16-bit operand type, or an output is not finite."""


def check(rc: int, what: str = "") -> None:
    """Map a negative status to the exception the reference would raise."""
    if rc == ZETT_OK:
        return
    msg = load().zett_last_error().decode("utf-8", "replace")
    if what:
        msg = f"{what}: {msg}"
    if rc == E_INDEX:
        raise IndexError(msg)
    if rc == E_NOT_IMPLEMENTED:
        raise NotImplementedError(msg)
    if rc == E_KEY:
        raise KeyError(msg)
    if rc == E_INVALID:
        raise ValueError(msg)
    if rc == E_RANGE:
        raise RangeError(msg)
    raise RuntimeError(msg)
