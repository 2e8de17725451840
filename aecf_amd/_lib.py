"""ctypes binding of libaecf_hip.so (the C ABI declared in include/aecf_hip.h).

The library is the product: there is no Python / PyTorch fallback for its arithmetic.  If
it is missing or an entry point is absent, loading fails loudly.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# AECF_LIB_PATH: another build of the same library (A/B timing of kernel variants on one box); default = the in-tree build
LIB_PATH = os.environ.get("AECF_LIB_PATH") or os.path.join(_HERE, "lib", "libaecf_hip.so")

AECF_ABI_VERSION = 9
AECF_BF16 = 0
AECF_F32 = 1
AECF_PRECISE = 1
AECF_DRAW_UNIFORMS = 2
AECF_HILO_GRADS = 4
AECF_PREP_READY = 8
AECF_FWD_STAGES = 4
AECF_BWD_STAGES = 8



class PoolDesc(Structure):
    _fields_ = [
        ("batch", c_int64),
        ("modalities", c_int32),
        ("embed_dim", c_int32),
        ("num_heads", c_int32),
        ("dtype", c_int32),
        ("mask_mode", c_int32),
        ("min_active", c_int32),
        ("base_mask_prob", c_float),
        ("entropy_target", c_float),
        ("eps", c_float),
    ]


class PoolFwdArgs(Structure):
    _fields_ = [
        ("x", c_void_p), ("query", c_void_p), ("w_in", c_void_p), ("b_in", c_void_p),
        ("w_out", c_void_p), ("b_out", c_void_p), ("key_padding_mask", c_void_p),
        ("uniforms", c_void_p), ("y", c_void_p), ("attn_w", c_void_p), ("masked_w", c_void_p),
        ("entropy", c_void_p), ("mask_rate", c_void_p), ("saved_probs", c_void_p),
        ("saved_o", c_void_p), ("saved_v", c_void_p), ("workspace", c_void_p), ("workspace_bytes", c_size_t),
        ("stage_events", c_void_p),
        ("info_attn_w", c_void_p), ("info_masked_w", c_void_p), ("info_entropy", c_void_p),
        ("info_mask_rate", c_void_p), ("saved_prep", c_void_p),
        ("info_target_entropy", c_void_p), ("target_entropy_value", c_float), ("flags", c_int32),
        ("ent_loss_partial", c_void_p),
        ("philox_seed", c_uint64), ("philox_offset", c_uint64), ("philox_threads", c_uint32), ("ent_loss", c_void_p),
        ("saved_o_lo", c_void_p), ("philox_element0", c_int64),
    ]


class PoolBwdArgs(Structure):
    _fields_ = [
        ("x", c_void_p), ("query", c_void_p), ("w_in", c_void_p), ("b_in", c_void_p),
        ("w_out", c_void_p), ("dy", c_void_p), ("d_attn_w", c_void_p), ("d_entropy", c_void_p),
        ("attn_w", c_void_p), ("saved_probs", c_void_p), ("saved_o", c_void_p), ("saved_v", c_void_p),
        ("dx", c_void_p),
        ("dquery", c_void_p), ("dw_in", c_void_p), ("db_in", c_void_p), ("dw_out", c_void_p),
        ("db_out", c_void_p), ("workspace", c_void_p), ("workspace_bytes", c_size_t),
        ("stage_events", c_void_p),
        ("grad_dtype", c_int32), ("flags", c_int32), ("saved_prep", c_void_p), ("param_grads_event", c_void_p),
        ("saved_o_lo", c_void_p), ("grad_scale", c_float),
    ]


class MhaDesc(Structure):
    _fields_ = [("batch", c_int64), ("tgt_len", c_int32), ("src_len", c_int32), ("embed_dim", c_int32),
                ("num_heads", c_int32), ("dtype", c_int32), ("dropout_p", c_float)]


class MhaFwdArgs(Structure):
    _fields_ = [("query", c_void_p), ("key", c_void_p), ("value", c_void_p), ("w_in", c_void_p), ("b_in", c_void_p),
                ("w_out", c_void_p), ("b_out", c_void_p), ("attn_mask", c_void_p), ("attn_mask_stride", c_int64),
                ("key_padding_mask", c_void_p), ("dropout_uniforms", c_void_p), ("y", c_void_p), ("attn_w", c_void_p),
                ("saved_q", c_void_p), ("saved_k", c_void_p), ("saved_v", c_void_p), ("saved_o", c_void_p),
                ("saved_probs", c_void_p)]


class MhaBwdArgs(Structure):
    _fields_ = [("query", c_void_p), ("key", c_void_p), ("value", c_void_p), ("w_in", c_void_p), ("w_out", c_void_p),
                ("dropout_uniforms", c_void_p), ("dy", c_void_p), ("d_attn_w", c_void_p), ("saved_q", c_void_p),
                ("saved_k", c_void_p), ("saved_v", c_void_p), ("saved_o", c_void_p), ("saved_probs", c_void_p),
                ("dquery", c_void_p), ("dkey", c_void_p), ("dvalue", c_void_p), ("dw_in", c_void_p), ("db_in", c_void_p),
                ("dw_out", c_void_p), ("db_out", c_void_p), ("workspace", c_void_p), ("workspace_bytes", c_size_t)]


# every symbol include/aecf_hip.h declares: (name, restype, argtypes)
_SYMBOLS = [
    ("aecf_abi_version", c_int, []),
    ("aecf_status_string", c_char_p, [c_int]),
    ("aecf_pool_stage_name", c_char_p, [c_int, c_int]),
    ("aecf_pool_check", c_int, [POINTER(PoolDesc)]),
    ("aecf_pool_fwd_workspace_bytes", c_size_t, [POINTER(PoolDesc)]),
    ("aecf_pool_bwd_workspace_bytes", c_size_t, [POINTER(PoolDesc)]),
    ("aecf_pool_prep_bytes", c_size_t, [POINTER(PoolDesc)]),
    ("aecf_pool_hilo_bwd_workspace_bytes", c_size_t, [POINTER(PoolDesc)]),
    ("aecf_pool_wants_saved_v", c_int, [POINTER(PoolDesc)]),
    ("aecf_pool_precise_workspace_bytes", c_size_t, [POINTER(PoolDesc), c_int]),
    ("aecf_philox_uniforms", c_int, [c_int64, c_uint64, c_uint64, c_uint32, c_int64, c_void_p, c_void_p]),
    ("aecf_philox_host", c_float, [c_uint64, c_uint64, c_uint32, c_int64, c_void_p]),
    ("aecf_pool_forward", c_int, [POINTER(PoolDesc), POINTER(PoolFwdArgs), c_void_p]),
    ("aecf_pool_backward", c_int, [POINTER(PoolDesc), POINTER(PoolBwdArgs), c_void_p]),
    ("aecf_curriculum_mask_forward", c_int,
     [c_int64, c_int32, c_int32, c_int32, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p,
      c_void_p, c_void_p, c_void_p, c_void_p]),
    ("aecf_curriculum_mask_backward", c_int,
     [c_int64, c_int32, c_int32, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    ("aecf_entropy_loss_workspace_bytes", c_size_t, [c_int64]),
    ("aecf_entropy_loss_from_partials", c_int, [c_int64, c_int32, c_void_p, c_void_p, c_void_p]),
    ("aecf_entropy_loss_fwd_bwd", c_int,
     [c_int64, c_int32, c_int32, c_float, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    ("aecf_sdpa_forward", c_int,
     [c_int64, c_int32, c_int32, c_int32, c_int32, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
      c_void_p, c_void_p]),
    ("aecf_sdpa_backward", c_int,
     [c_int64, c_int32, c_int32, c_int32, c_int32, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    ("aecf_mha_check", c_int, [POINTER(MhaDesc)]),
    ("aecf_mha_bwd_workspace_bytes", c_size_t, [POINTER(MhaDesc)]),
    ("aecf_mha_forward", c_int, [POINTER(MhaDesc), POINTER(MhaFwdArgs), c_void_p]),
    ("aecf_mha_backward", c_int, [POINTER(MhaDesc), POINTER(MhaBwdArgs), c_void_p]),
    ("aecf_modality_frontend", c_int, [c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    ("aecf_loss_fwd_bwd", c_int,
     [c_int64, c_int64, c_int64, c_int32, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
      c_int32, c_float, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    ("aecf_route_build", c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    ("aecf_rows_gather", c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    ("aecf_rows_select", c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    ("aecf_front_pair", c_int, [c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    ("aecf_cast_f32_to_bf16", c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    ("aecf_adamw_step", c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float,
                                c_float, c_float, c_float, c_void_p]),
    ("aecf_rows_split", c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    ("aecf_l2norm_forward", c_int, [c_int64, c_int32, c_int32, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    ("aecf_l2norm_backward", c_int, [c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    ("aecf_nce_workspace_bytes", c_size_t, [c_int64, c_int64, c_int32, c_int32]),
    ("aecf_nce_stream_workspace_bytes", c_size_t, [c_int64, c_int64, c_int32, c_int32]),
    ("aecf_nce_sym_workspace_bytes", c_size_t, [c_int64, c_int64, c_int32]),
    ("aecf_nce_sym_pass1", c_int,
     [c_int64, c_int64, c_int32, c_float, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    ("aecf_nce_sym_loss", c_int,
     [c_int64, c_int64, c_int64, c_int32, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_int64,
      c_int32, c_float, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    ("aecf_nce_sym_grads", c_int,
     [c_int64, c_int64, c_int64, c_int32, c_float, c_float, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_int32,
      c_void_p, c_void_p, c_void_p]),
    ("aecf_nce_fwd_bwd", c_int,
     [c_int64, c_int64, c_int64, c_int32, c_int32, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
      c_void_p, c_void_p, c_size_t, c_void_p]),
]
SYMBOL_NAMES = [s[0] for s in _SYMBOLS]

_lib = None


def load() -> ctypes.CDLL:
    """Load libaecf_hip.so once.  Raises RuntimeError (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"aecf_amd: HIP library not found at {LIB_PATH}. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C aecf_amd/csrc`. "
            "There is no CPU / PyTorch fallback for the fusion path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, restype, argtypes in _SYMBOLS:
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover - build error
            raise RuntimeError(f"aecf_amd: {LIB_PATH} does not export {name}") from e
        fn.restype = restype
        fn.argtypes = argtypes
    ver = lib.aecf_abi_version()
    if ver != AECF_ABI_VERSION:
        raise RuntimeError(f"aecf_amd: ABI version mismatch: library {ver}, binding {AECF_ABI_VERSION}")
    _lib = lib
    return lib


def status_string(status: int) -> str:
    return load().aecf_status_string(status).decode()


def check(status: int, what: str) -> None:
    if status != 0:
        raise RuntimeError(f"aecf_amd: {what} failed: {status_string(status)} (status {status})")
