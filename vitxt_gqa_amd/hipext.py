"""ctypes binding of libt2s_hip.so (C ABI declared in include/t2s_hip.h).

The library is built in-tree by ``python -m vitxt_gqa_amd.build`` (also called by
``__graft_entry__.build()``).  There is NO fallback: if the shared object is missing or a call
fails, a RuntimeError is raised -- the product path never routes around the HIP kernels.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int32, c_int64, c_uint64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("T2S_HIP_LIB") or os.path.join(_HERE, "libt2s_hip.so")      # override: kernel build experiments only
ABI_VERSION = 6

T2S_F32, T2S_BF16 = 0, 1

_SIGS = {
    "t2s_abi_version": (c_int, []),
    "t2s_last_error": (c_char_p, []),
    "t2s_compact_keys": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "t2s_attn_fwd": (c_int, [c_void_p] * 7 + [c_int] * 6 + [c_int64] * 6 + [c_float, c_int, c_float, c_uint64, c_void_p]),
    "t2s_attn_dropout_mask": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_float, c_uint64, c_void_p]),
    "t2s_attn_bwd": (c_int, [c_void_p] * 12 + [c_int] * 7 + [c_int64] * 6 + [c_float, c_int, c_float, c_uint64, c_void_p]),
    "t2s_attn_bwd_fill": (c_int, [c_void_p] * 13 + [c_int] * 7 + [c_int64] * 6 + [c_float, c_int, c_float, c_uint64, c_void_p]),
    "t2s_attn_bwd_fused_workspace_bytes": (c_int64, [c_int, c_int, c_int]),
    "t2s_attn_bwd_fused": (c_int, [c_void_p] * 11 + [c_int64, c_int] + [c_void_p] * 3 + [c_int] * 7 + [c_int64] * 6 + [c_float, c_int, c_float, c_uint64, c_void_p]),
    "t2s_add_layernorm_fwd": (c_int, [c_void_p] * 8 + [c_int64, c_float, c_int, c_int, c_float, c_uint64, c_void_p]),
    "t2s_wide_add_layernorm_fwd": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_float, c_void_p]),
    "t2s_add_layernorm_fwd_nres": (c_int, [c_void_p] * 11 + [c_int64, c_float, c_int, c_int, c_float, c_uint64, c_void_p]),
    "t2s_layernorm_bwd_parts": (c_int, [c_int64]),
    "t2s_add_layernorm_bwd": (c_int, [c_void_p] * 8 + [c_int64, c_int, c_int, c_int, c_float, c_uint64, c_void_p]),
    "t2s_add_layernorm_bwd_bias": (c_int, [c_void_p] * 9 + [c_int64, c_int, c_int, c_int, c_float, c_uint64, c_void_p]),
    "t2s_dropout_mask": (c_int, [c_void_p, c_int64, c_float, c_uint64, c_void_p]),
    "t2s_gelu_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "t2s_gelu_bwd_parts": (c_int, [c_int64]),
    "t2s_gelu_bwd": (c_int, [c_void_p] * 4 + [c_int64, c_int, c_int, c_void_p]),
    "t2s_ptr_scores": (c_int, [c_void_p] * 4 + [c_int, c_int, c_int, c_int64, c_int, c_float, c_int, c_int, c_void_p]),
    "t2s_question_pool": (c_int, [c_void_p] * 5 + [c_int, c_int, c_void_p]),
    "t2s_attention_score": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "t2s_ground_select": (c_int, [c_void_p] * 6 + [c_int64, c_int] + [c_void_p] * 11 + [c_int] * 5 + [c_void_p]),
    "t2s_tanh_residual_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "t2s_tanh_residual_bwd": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int64, c_int64, c_void_p]),
    "t2s_ocr_tail_parts": (c_int, [c_int64]),
    "t2s_ocr_tail_fwd": (c_int, [c_void_p, c_int] + [c_void_p] * 9 + [c_int64, c_float, c_float, c_uint64, c_void_p]),
    "t2s_ocr_tail_bwd": (c_int, [c_void_p, c_void_p, c_int] + [c_void_p] * 8 + [c_int64, c_float, c_uint64, c_void_p]),
    "t2s_add_cast": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_void_p, c_int64, c_int64, c_void_p]),
    "t2s_embed_rows": (c_int, [c_void_p, c_int, c_void_p, c_int] + [c_void_p] * 4 + [c_int, c_int, c_void_p, c_int, c_int64, c_int, c_void_p]),
    "t2s_bce_masked": (c_int, [c_void_p] * 5 + [c_int64, c_int, c_void_p]),
    "t2s_infonce_stats": (c_int, [c_void_p] * 4 + [c_int64, c_int, c_void_p]),
    "t2s_infonce_bwd": (c_int, [c_void_p] * 7 + [c_int64, c_int, c_void_p]),
    "t2s_fasttext_rows": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "t2s_optim_chunk_elems": (c_int, []),
    "t2s_status_accumulate": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "t2s_status_gate": (c_int, [c_void_p, c_void_p, c_void_p]),
    "t2s_grad_sqnorm": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "t2s_clip_coef": (c_int, [c_void_p, c_int, c_float, c_void_p, c_void_p]),
    "t2s_adam_step": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_int, c_void_p, c_int, c_void_p]),
    "t2s_phoc": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p]),
    "t2s_gelu_tables": (c_int, [c_void_p, c_void_p, c_void_p]),
    "t2s_gemm_nt_colsum_rows": (c_int, [c_int64]),
    "t2s_gemm_nt": (c_int, [c_void_p] * 4 + [c_int64, c_int, c_int, c_int64, c_int64, c_int64, c_int] + [c_void_p] * 5),
    "t2s_gemm_wgrad_splits": (c_int, [c_int64, c_int, c_int]),
    "t2s_gemm_wgrad": (c_int, [c_void_p] * 4 + [c_int64, c_int, c_int, c_int64, c_int64, c_int, c_int, c_void_p]),
}

_lib = None


def exported_symbols():
    """Names every build of the library must export (mirrors include/t2s_hip.h)."""
    return sorted(_SIGS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libt2s_hip.so not found at %s: build it with `python -m vitxt_gqa_amd.build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU/PyTorch fallback for the T2S hot path." % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)          # AttributeError if the build lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        if l.t2s_abi_version() != ABI_VERSION:
            raise RuntimeError("libt2s_hip.so ABI version %d != expected %d" % (l.t2s_abi_version(), ABI_VERSION))
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (rc=%d): %s" % (what, rc, lib().t2s_last_error().decode()))


def dtype_code(t):
    if t.dtype == torch.bfloat16:
        return T2S_BF16
    if t.dtype == torch.float32:
        return T2S_F32
    raise TypeError("T2S HIP kernels take float32 or bfloat16 tensors, got %s" % t.dtype)


def ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("T2S HIP kernels need device tensors (got a %s tensor); there is no CPU path" % t.device)
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream
