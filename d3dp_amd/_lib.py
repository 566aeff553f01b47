"""ctypes binding of libd3dp_hip.so (C ABI declared in include/d3dp_hip.h).

There is deliberately NO fallback: if the shared library is missing or no MI355X is visible the
import of the library / creation of a context raises.  The CPU oracle under ``oracle/`` is test
infrastructure and is never imported from here.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# D3DP_LIB=path: load another build of the library (A/B runs of differently compiled kernels; test-only)
LIB_PATH = os.environ.get("D3DP_LIB") or os.path.join(_HERE, "lib", "libd3dp_hip.so")

MODE_EXACT, MODE_FAST, MODE_TRAIN = 0, 1, 2
MODE_SPLIT3 = 2   # d3dp_op_linear only: split-bf16 operands
EPI_BIAS, EPI_GELU, EPI_RESID, EPI_QKV_PACK = 0, 1, 2, 4     # (| D << 8: the skewed schedule of epi 1 / 4, include/d3dp_hip.h)
PROFILE_CLASSES = 25


class AdamChunk(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_int32),
                ("pad", C.c_int32)]

ABI_VERSION = 4


class Cfg(C.Structure):
    _fields_ = [("frames", C.c_int32), ("joints", C.c_int32), ("channels", C.c_int32), ("depth", C.c_int32),
                ("heads", C.c_int32), ("hidden", C.c_int32), ("eps_block", C.c_float), ("eps_head", C.c_float),
                ("mode", C.c_int32), ("chunk_seqs", C.c_int32)]


_BLOCK_FIELDS = ["norm1_w", "norm1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "norm2_w", "norm2_b",
                 "fc1_w", "fc1_b", "fc2_w", "fc2_b"]


class BlockWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _BLOCK_FIELDS]


_TOP_FIELDS = ["spatial_pos", "temporal_pos", "embed_w", "embed_b", "time_freq", "time1_w", "time1_b", "time3_w",
               "time3_b", "spatial_norm_w", "spatial_norm_b", "temporal_norm_w", "temporal_norm_b", "head_norm_w",
               "head_norm_b", "head_w", "head_b"]


class Weights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _TOP_FIELDS] + [("ste", C.POINTER(BlockWeights)),
                                                          ("tte", C.POINTER(BlockWeights))]


# name -> (restype, argtypes); every entry must be exported by the library (tests/test_abi.py)
PROTOTYPES = {
    "d3dp_abi_version": (C.c_int, []),
    "d3dp_last_error": (C.c_char_p, []),
    "d3dp_create": (C.c_int, [C.POINTER(Cfg), C.POINTER(C.c_void_p)]),
    "d3dp_destroy": (C.c_int, [C.c_void_p]),
    "d3dp_set_weights": (C.c_int, [C.c_void_p, C.POINTER(Weights), C.c_void_p]),
    "d3dp_set_weights_borrowed": (C.c_int, [C.c_void_p, C.POINTER(Weights)]),
    "d3dp_workspace_bytes": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_size_t)]),
    "d3dp_denoise": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                               C.c_void_p, C.c_size_t, C.c_void_p]),
    "d3dp_ddim_pre": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_int32, C.c_int32,
                                C.c_int32, C.c_void_p]),
    "d3dp_ddim_post": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_double, C.c_double,
                                 C.c_float, C.c_float, C.c_float, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p,
                                 C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "d3dp_q_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int32,
                                C.c_int32, C.c_void_p]),
    "d3dp_train_workspace_bytes": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_size_t)]),
    "d3dp_train_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                     C.c_void_p, C.c_size_t, C.c_void_p]),
    "d3dp_train_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.POINTER(Weights), C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "d3dp_jpma": (C.c_int, [C.c_void_p] * 9 + [C.c_int32] * 6 + [C.c_void_p]),
    "d3dp_jpma_gathered": (C.c_int, [C.c_void_p] * 9 + [C.c_int32] * 7 + [C.c_void_p]),
    "d3dp_clip_count": (C.c_int, [C.c_int32, C.c_int32]),
    "d3dp_clip_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                   C.c_int32, C.c_void_p]),
    "d3dp_clip_scatter": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 7 + [C.c_void_p]),
    "d3dp_jpma_ex": (C.c_int, [C.c_void_p] * 11 + [C.c_int32] * 7 + [C.c_void_p]),
    "d3dp_jpma_winners": (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 7 + [C.c_void_p]),
    "d3dp_jpma_combine": (C.c_int, [C.c_void_p, C.c_int32, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "d3dp_batch_gather": (C.c_int, [C.c_void_p] * 7 + [C.c_int32] * 4 + [C.c_void_p]),
    "d3dp_adamw_step": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                  C.c_int64, C.c_void_p]),
    "d3dp_procrustes": (C.c_int, [C.c_void_p] * 4 + [C.c_int32] * 4 + [C.c_void_p]),
    "d3dp_op_linear": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                 C.c_int32, C.c_int32, C.c_void_p]),
    "d3dp_op_attention": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "d3dp_op_layernorm": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int32,
                                    C.c_int32, C.c_void_p]),
    "d3dp_op_to_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "d3dp_op_split3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "d3dp_op_split2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_void_p]),
    "d3dp_op_linear_x2": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int32,
                                    C.c_int32, C.c_int32, C.c_void_p]),
    "d3dp_profile_enable": (C.c_int, [C.c_void_p, C.c_int32]),
    "d3dp_exact_range_bound": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "d3dp_exact_scales": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    "d3dp_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "d3dp_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "d3dp_profile_class_name": (C.c_char_p, [C.c_int32]),
    # test hooks (include/d3dp_hip.h, last section)
    "d3dp_debug_x2_variants": (C.c_int, []),
    "d3dp_debug_train_linear": (C.c_int, [C.c_void_p] * 4 + [C.c_int32] * 4 + [C.c_void_p, C.c_int32, C.c_void_p]),
}

_lib: Optional[C.CDLL] = None


class D3DPHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libd3dp_hip.so and bind every prototype.  Raises if the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise D3DPHipError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C d3dp_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            if os.environ.get("D3DP_LIB_ANY_ABI") == "1":
                continue
            raise D3DPHipError(f"{LIB_PATH} does not export {name}")
        fn.restype, fn.argtypes = res, args
    if lib.d3dp_abi_version() != ABI_VERSION and os.environ.get("D3DP_LIB_ANY_ABI") != "1":   # (=1: A/B probes of old builds)
        raise D3DPHipError(f"ABI mismatch: library {lib.d3dp_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().d3dp_last_error()
        raise D3DPHipError(f"{what or 'libd3dp_hip'} failed ({rc}): {msg.decode() if msg else ''}")


def ptr(t) -> int:
    """Device pointer of a contiguous torch tensor (0 for None)."""
    if t is None:
        return 0
    assert t.is_contiguous(), "libd3dp_hip takes contiguous buffers"
    return t.data_ptr()


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
