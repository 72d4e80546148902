"""ctypes binding of libperceptor_hip.so (include/perceptor_hip.h).

The product path has no CPU fallback: every wrapper raises if the library is
missing or a tensor is not on a HIP device.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PMI_LIB") or os.path.join(_HERE, "csrc", "libperceptor_hip.so")   # PMI_LIB: diagnostic builds (tools/ only)

DT_F16, DT_BF16, DT_F16X2 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_SILU, ACT_GELU, ACT_QUICKGELU = 0, 1, 2, 3, 4
ACT_GEGLU = 5      # pmi_igemm (weights-direct GEMM only): value * gelu(gate) over (16 value | 16 gate) column groups, N / 2 output columns
TORCH_DTYPE = {DT_F16: torch.float16, DT_BF16: torch.bfloat16, DT_F16X2: torch.float16}


def dtype_code(name) -> int:
    if name in (DT_F16, "f16", "fp16", torch.float16):
        return DT_F16
    if name in (DT_BF16, "bf16", torch.bfloat16):
        return DT_BF16
    if name in (DT_F16X2, "precise", "f16x2"):
        return DT_F16X2
    raise ValueError(f"unsupported compute dtype {name!r} (f16, bf16 or precise)")


class IgemmArgs(C.Structure):
    _fields_ = [
        ("A0", C.c_void_p), ("A1", C.c_void_p), ("B", C.c_void_p), ("bias", C.c_void_p), ("nbias", C.c_void_p),
        ("R", C.c_void_p), ("D", C.c_void_p), ("pro_a", C.c_void_p), ("pro_b", C.c_void_p), ("ws", C.c_void_p), ("stats", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("C0", C.c_int32), ("C1", C.c_int32),
        ("lda0", C.c_int32), ("lda1", C.c_int32), ("ldb", C.c_int32), ("ldd", C.c_int32), ("ldr", C.c_int32),
        ("H", C.c_int32), ("W", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32),
        ("taps", C.c_int32), ("stride", C.c_int32), ("up", C.c_int32), ("res_up", C.c_int32), ("act", C.c_int32),
        ("out_f32", C.c_int32), ("res_f32", C.c_int32), ("hw", C.c_int32), ("alpha", C.c_float),
        ("batch", C.c_int32), ("batch_inner", C.c_int32),
        ("sA_o", C.c_int64), ("sA_i", C.c_int64), ("sB_o", C.c_int64), ("sB_i", C.c_int64),
        ("sD_o", C.c_int64), ("sD_i", C.c_int64), ("sR_o", C.c_int64), ("sR_i", C.c_int64),
        ("dtype", C.c_int32), ("ldnb", C.c_int32), ("pro_act", C.c_int32), ("stats_p", C.c_int32), ("splitk", C.c_int32), ("reserved", C.c_int32),
        ("Bf", C.c_void_p), ("split_out", C.c_int32), ("split_in", C.c_int32),
        ("D2", C.c_void_p), ("aux", C.c_void_p), ("aux_act", C.c_int32), ("reserved3", C.c_int32),
    ]


class GemmF32Args(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("bias", C.c_void_p), ("D", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("lda", C.c_int32), ("ldb", C.c_int32), ("ldd", C.c_int32),
                ("transB", C.c_int32), ("act", C.c_int32), ("alpha", C.c_float), ("batch", C.c_int32), ("batch_inner", C.c_int32),
                ("sA_o", C.c_int64), ("sA_i", C.c_int64), ("sB_o", C.c_int64), ("sB_i", C.c_int64), ("sD_o", C.c_int64), ("sD_i", C.c_int64), ("R", C.c_void_p)]


_lib = None

_I, _F, _P, _L = C.c_int, C.c_float, C.c_void_p, C.c_int64
_PROTOS = {
    "pmi_abi_version": ([], ),
    "pmi_igemm": ([C.POINTER(IgemmArgs), _P],),
    "pmi_conv3x3_halo_config": ([C.POINTER(IgemmArgs)],),
    "pmi_igemm_stats_rows": ([C.POINTER(IgemmArgs)],),
    "pmi_igemm_splitk": ([C.POINTER(IgemmArgs)],),
    "pmi_set_option": ([_I, _I],),
    "pmi_gemm_wd_eligible": ([C.POINTER(IgemmArgs)],),
    "pmi_gemm_f32": ([C.POINTER(GemmF32Args), _P],),
    "pmi_softmax_f32": ([_P, _I, _I, _I, _F, _P],),
    "pmi_split_from_f32": ([_P, _I, _P, _L, _I, _P],),
    "pmi_split_to_f32": ([_P, _P, _L, _I, _P],),
    "pmi_split_convert": ([_P, _P, _L, _I, _I, _P],),
    "pmi_gn_stats": ([_P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P],),
    "pmi_gn_finalize": ([_P, _I, _I, _P, _I, _I, _P, _P, _P, _I, _P, _P, _I, _I, _I, _F, _P],),
    "pmi_gn_apply_pool_skip": ([_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],),
    "pmi_gn_apply": ([_P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],),
    "pmi_attn_flash_workspace": ([_I, _I, _I, _I, _I],),
    "pmi_attn_flash": ([_P, _I, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P],),
    "pmi_qkv_split": ([_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],),
    "pmi_attn_d64": ([_P, _P, _P, _P, _I, _I, _I, _F, _I, _P],),
    "pmi_vit_attn_fwd": ([_P, _P, _P, _P, _I, _I, _I, _F, _I, _P],),
    "pmi_vit_attn_bwd": ([_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _P],),
    "pmi_prep_input": ([_P, _P, _I, _P, _I, _I, _I, _I, _I, _P],),
    "pmi_finish_output": ([_P, _I, _P, _I, _I, _I, _I, _P],),
    "pmi_nchw_to_nhwc": ([_P, _P, _I, _I, _I, _I, _I, _F, _F, _I, _P],),
    "pmi_nhwc_to_nchw": ([_P, _I, _P, _I, _I, _I, _I, _F, _F, _P],),
    "pmi_geglu": ([_P, _P, _L, _I, _I, _I, _P],),
    "pmi_avgpool2": ([_P, _P, _I, _I, _I, _I, _I, _P],),
    "pmi_upsample_bilinear2": ([_P, _P, _I, _I, _I, _I, _I, _P],),
    "pmi_upsample_nearest2": ([_P, _P, _I, _I, _I, _I, _P],),
    "pmi_timestep_embedding": ([_P, _P, _I, _I, _F, _I, _P],),
    "pmi_fourier_features": ([_P, _P, _P, _I, _I, _P],),
    "pmi_cast_f32_to_16": ([_P, _P, _L, _I, _I, _P],),
    "pmi_ddim_eps_step": ([_P, _P, _P, _P, _P, _P, _P, _P, _I, _L, _P],),
    "pmi_ddim_v_step": ([_P, _P, _P, _P, _P, _P, _P, _P, _I, _L, _P],),
    "pmi_guided_update": ([_P, _P, _P, _F, _F, _P, _I, _L, _P],),
    "pmi_lincomb2": ([_P, _P, _P, _P, _P, _P, _I, _L, _P],),
    "pmi_clamp": ([_P, _P, _P, _P, _I, _L, _P],),
    # Predictions variants, clamp_with_grad (sampling.hip)
    "pmi_quantile_abs": ([_P, _P, _I, _L, _F, _P],),
    "pmi_randn": ([_P, _L, _L, _L, _L, _P],),
    "pmi_philox4x32_10": ([_P, _L, _L, _L, _P],),
    "pmi_sort_rows_padded": ([_L],),
    "pmi_sort_rows": ([_P, _P, _I, _L, _P],),
    "pmi_wasserstein": ([_P, _I, _L, _I, _P, _P, _P],),
    "pmi_clamp_grad": ([_P, _P, _P, _P, _P, _I, _L, _P],),
    # UNet input-gradient adjoints (backward.hip)
    "pmi_add16": ([_P, _P, _P, _L, _I, _P],),
    "pmi_avgpool2_bwd": ([_P, _P, _I, _I, _I, _I, _I, _P],),
    "pmi_upsample_bilinear2_bwd": ([_P, _P, _I, _I, _I, _I, _I, _P],),
    "pmi_upsample_nearest2_bwd": ([_P, _P, _I, _I, _I, _I, _I, _P],),
    "pmi_gn_bwd_stats": ([_P, _P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P],),
    "pmi_gn_bwd_finalize": ([_P, _I, _I, _P, _I, _I, _P, _I, _P, _P, _I, _P, _P, _I, _I, _I, _F, _P],),
    "pmi_gn_bwd_apply": ([_P, _P, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P],),
    "pmi_gn1_bwd_partials": ([_L, _I],),
    "pmi_gn1_bwd": ([_P, _P, _P, _I, _F, _P, _P, _P, _I, _L, _I, _F, _I, _P],),
    # CLIP path (clip.hip)
    "pmi_layernorm_fwd": ([_P, _I, _P, _P, _P, _P, _P, _I, _I, _F, _I, _P],),
    "pmi_layernorm_bwd": ([_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],),
    "pmi_layernorm_fwd_slabs": ([_P, _I, _L, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _I, _P],),
    "pmi_layernorm_bwd_slabs": ([_P, _I, _L, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],),
    "pmi_softmax_fwd": ([_P, _P, _I, _I, _I, _I, _F, _I, _P],),
    "pmi_softmax_causal_fwd": ([_P, _P, _I, _I, _I, _I, _F, _I, _P],),
    "pmi_embed_tokens": ([_P, _P, _P, _P, _I, _I, _I, _I, _P],),
    "pmi_gather_rows": ([_P, _P, _P, _I, _I, _I, _L, _P],),
    "pmi_softmax_bwd": ([_P, _P, _P, _I, _I, _I, _I, _F, _I, _P],),
    "pmi_transpose_16": ([_P, _P, _I, _I, _I, _L, _L, _I, _I, _P],),
    "pmi_act_bwd": ([_P, _P, _P, _L, _I, _I, _P],),
    "pmi_resize_apply": ([_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],),
    "pmi_patchify": ([_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],),
    "pmi_unpatchify": ([_P, _P, _P, _I, _I, _I, _I, _F, _P],),
    "pmi_act_fwd": ([_P, _P, _L, _I, _I, _P],),
    "pmi_l2norm_rows": ([_P, _P, _I, _I, _F, _P],),
    "pmi_vit_assemble": ([_P, _P, _P, _P, _I, _I, _I, _I, _P],),
    "pmi_spherical_loss": ([_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _P],),
}


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -m perceptor_amd.csrc.build` "
                "(perceptor_amd has no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        for name, (args,) in _PROTOS.items():
            if not hasattr(L, name):
                continue
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = C.c_int
        _lib = L
    return _lib


def exported_symbols():
    L = lib()
    return [n for n in _PROTOS if hasattr(L, n)]


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("perceptor_amd kernels need tensors on a HIP device (no CPU fallback)")
    return t.data_ptr()


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc} "
                           f"({'bad argument' if rc == -1 else 'launch error' if rc == -2 else 'unknown'})")


def call(name: str, *args) -> None:
    fn = getattr(lib(), name)
    check(fn(*args, stream_ptr()), name)
