"""ctypes binding of libvitmi.so (include/vitmi.h).

The product path has no CPU or PyTorch fallback: if the shared object is
missing or a call fails, this module raises.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import os

HERE = Path(__file__).resolve().parent
# VITMI_LIB: another build of the same library (A/B of two builds in separate processes on one box: tools/lib_ab.sh)
LIB_PATH = Path(os.environ["VITMI_LIB"]).resolve() if os.environ.get("VITMI_LIB") else HERE / "libvitmi.so"

F32, BF16 = 0, 1
EPI_STORE, EPI_BIAS_GELU, EPI_RESIDUAL, EPI_DGELU, EPI_PATCH_POS = 0, 1, 2, 3, 4
GEMM_AUTO, GEMM_GENERIC, GEMM_FAST = 0, 1, 2
LAUNCH_SHARED_DEVICE = 1
LAUNCH_ROWS_PADDED = 2

c_i64, c_i32, c_f32, c_vp, c_sz = C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_size_t


class GemmDesc(C.Structure):
    _fields_ = [
        ("struct_size", c_i64),
        ("M", c_i64), ("N", c_i64), ("K", c_i64),
        ("A", c_vp), ("lda", c_i64), ("a_kmajor", c_i32),
        ("B", c_vp), ("ldb", c_i64), ("b_kmajor", c_i32),
        ("in_dtype", c_i32), ("epilogue", c_i32),
        ("C", c_vp), ("ldc", c_i64), ("c_dtype", c_i32),
        ("C2", c_vp), ("ldc2", c_i64),
        ("bias", c_vp),
        ("R", c_vp), ("ldr", c_i64), ("r_dtype", c_i32),
        ("gamma", c_vp),
        ("AUX", c_vp), ("ldaux", c_i64),
        ("pos", c_vp), ("n_tok", c_i64), ("cls", c_vp),
        ("alpha", c_f32), ("accumulate", c_i32), ("impl", c_i32),
        ("workspace", c_vp), ("workspace_bytes", c_sz),
        ("batch", c_i64), ("batch_inner", c_i64),
        ("a_bs", c_i64 * 2), ("b_bs", c_i64 * 2), ("c_bs", c_i64 * 2),
        ("rowscale", c_vp), ("rows_per_group", c_i64),
        ("colsum_part", c_vp),
        ("aux_is_derivative", c_i32),
        ("launch_flags", c_i32),
    ]

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.struct_size = C.sizeof(GemmDesc)      # the library rejects any other layout


class FoldDesc(C.Structure):
    """vitmi_fold_desc: one deferred fold of fp32 partial column sums (include/vitmi.h)."""
    _fields_ = [
        ("struct_size", c_i64),
        ("part", c_vp), ("S", c_i32), ("nseg", c_i32),
        ("N", c_i64), ("ld", c_i64),
        ("out", c_vp * 3),
    ]

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.struct_size = C.sizeof(FoldDesc)


# name -> (restype, argtypes); every symbol declared in include/vitmi.h
SIGNATURES = {
    "vitmi_version": (C.c_int, []),
    "vitmi_last_error_string": (C.c_char_p, []),
    "vitmi_gemm": (C.c_int, [C.POINTER(GemmDesc), c_vp]),
    "vitmi_gemm_pair_workspace": (c_sz, [C.POINTER(GemmDesc), C.POINTER(GemmDesc)]),
    "vitmi_gemm_pair": (C.c_int, [C.POINTER(GemmDesc), C.POINTER(GemmDesc), c_vp, c_sz, c_vp]),
    "vitmi_gemm_uses_fast": (C.c_int, [C.POINTER(GemmDesc)]),
    "vitmi_gemm_workspace": (c_sz, [C.POINTER(GemmDesc)]),
    "vitmi_layernorm_fwd": (C.c_int, [c_vp, C.c_int, c_i64, c_vp, c_vp, c_vp, C.c_int, c_i64,
                                      c_vp, c_vp, c_i64, c_i64, c_f32, c_vp]),
    "vitmi_layernorm_bwd_workspace": (c_sz, [c_i64, c_i64]),
    "vitmi_layernorm_bwd": (C.c_int, [c_vp, C.c_int, c_i64, c_vp, C.c_int, c_i64, c_vp, c_vp, c_vp,
                                      c_vp, c_vp, C.c_int, c_i64, c_vp, C.c_int, c_i64, c_vp, c_vp, c_vp, c_vp,
                                      c_vp, c_i64, c_i64, c_i64, c_vp, c_sz, c_vp]),
    "vitmi_layernorm_bwd_deferred": (C.c_int, [c_vp, C.c_int, c_i64, c_vp, C.c_int, c_i64, c_vp, c_vp, c_vp,
                                               c_vp, c_vp, C.c_int, c_i64, c_vp, C.c_int, c_i64, c_vp, c_vp, c_vp, c_vp,
                                               c_vp, c_i64, c_i64, c_i64, c_vp, c_sz, C.POINTER(FoldDesc), c_vp]),
    "vitmi_fold_many": (C.c_int, [C.POINTER(FoldDesc), C.c_int, c_vp]),
    "vitmi_attn_fwd": (C.c_int, [c_vp, c_vp, c_vp, C.c_int, c_i64, c_i64, c_i64, c_i64, c_f32, c_vp]),
    "vitmi_attn_bwd_workspace": (c_sz, [c_i64, c_i64, c_i64]),
    "vitmi_attn_bwd_dbias_rows": (c_i64, [c_i64, c_i64]),
    "vitmi_attn_bwd": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int, c_i64, c_i64, c_i64, c_i64,
                                 c_f32, c_vp, c_i32, c_vp, c_sz, c_vp]),
    "vitmi_th_softmax_fwd": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int, c_i64, c_i64, c_i64, c_i64, c_i64, c_vp]),
    "vitmi_th_softmax_bwd_workspace": (c_sz, [c_i64, c_i64, c_i64]),
    "vitmi_th_softmax_bwd": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int,
                                       c_i64, c_i64, c_i64, c_i64, c_i64, c_vp, c_sz, c_vp]),
    "vitmi_th_attn_supported": (C.c_int, [C.c_int, c_i64, c_i64, c_i64]),
    "vitmi_th_attn_workspace": (c_sz, [c_i64, c_i64, c_i64, c_i64]),
    "vitmi_th_attn_fwd": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int, c_i64, c_i64, c_i64, c_i64, c_f32, c_vp, c_sz, c_vp]),
    "vitmi_th_attn_bwd": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, C.c_int,
                                    c_i64, c_i64, c_i64, c_i64, c_f32, c_vp, c_sz, c_vp]),
    "vitmi_class_attn_fwd": (C.c_int, [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, C.c_int, c_i64, c_i64, c_i64, c_i64, c_f32, c_vp]),
    "vitmi_class_attn_bwd": (C.c_int, [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, C.c_int,
                                       c_i64, c_i64, c_i64, c_i64, c_f32, c_vp]),
    "vitmi_colsum_mul_workspace": (c_sz, [c_i64, c_i64]),
    "vitmi_colsum_mul": (C.c_int, [c_vp, C.c_int, c_i64, c_vp, C.c_int, c_i64, c_i64, c_i64, c_vp, c_vp, c_sz, c_vp]),
    "vitmi_win_attn_fwd": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64,
                                     c_i64, c_i64, c_f32, c_vp]),
    "vitmi_win_attn_bwd_workspace": (c_sz, [c_i64, c_i64, c_i64]),
    "vitmi_win_attn_bwd_fuses_qkv_bias": (C.c_int, [C.c_int, c_i64]),
    "vitmi_win_attn_bwd": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int, c_i64, c_i64, c_i64, c_i64,
                                     c_i64, c_i64, c_i64, c_i64, c_f32, c_vp, c_sz, c_vp]),
    "vitmi_relpos_bias": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp]),
    "vitmi_patch_merge": (C.c_int, [c_vp, c_vp, C.c_int, c_i64, c_i64, c_i64, c_i64, C.c_int, c_vp]),
    "vitmi_token_mean": (C.c_int, [c_vp, c_vp, c_vp, c_vp, C.c_int, c_i64, c_i64, c_i64, c_vp]),
    "vitmi_split3": (C.c_int, [c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, C.c_int, C.c_int, c_vp]),
    "vitmi_gelu_fwd": (C.c_int, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp]),
    "vitmi_gelu_bwd": (C.c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp]),
    "vitmi_cast": (C.c_int, [c_vp, C.c_int, c_vp, C.c_int, c_i64, c_vp]),
    "vitmi_axpy": (C.c_int, [c_vp, c_vp, c_f32, c_i64, c_vp]),
    "vitmi_scale_cast": (C.c_int, [c_vp, C.c_int, c_i64, c_vp, c_vp, c_i64, c_vp, C.c_int, c_i64, c_i64, c_i64, c_vp]),
    "vitmi_patchify": (C.c_int, [c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, C.c_int, c_i64, c_i64, c_i64,
                                 c_i64, c_i64, c_i64, C.c_int, c_vp]),
    "vitmi_pos_resample": (C.c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp]),
    "vitmi_colsum_workspace": (c_sz, [c_i64, c_i64]),
    "vitmi_colsum": (C.c_int, [c_vp, C.c_int, c_i64, c_i64, c_i64, c_vp, c_vp, c_sz, c_vp]),
    "vitmi_softmax_xent": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp]),
    "vitmi_sgd_momentum": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_vp]),
    "vitmi_image_ingest": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64,
                                     c_i64, c_vp]),
    "vitmi_ingest_patchify": (C.c_int, [c_vp, c_vp, C.c_int, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64,
                                        c_i64, c_i64, c_i64, c_i64, C.c_int, c_vp]),
    "vitmi_adam": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, C.c_int,
                             c_f32, c_vp]),
    "vitmi_adagrad": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_vp]),
    "vitmi_adadelta": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_vp]),
    "vitmi_adabelief": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, C.c_int,
                                  C.c_int, c_f32, c_vp]),
}

_lib = None


class VitmiError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libvitmi.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise VitmiError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -m vit_torch_amd.build). There is no CPU/PyTorch fallback.")
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().vitmi_last_error_string()
        raise VitmiError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")
