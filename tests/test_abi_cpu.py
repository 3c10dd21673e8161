"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and
exports every symbol include/vitmi.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vitmi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vitmi_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_something():
    syms = declared_symbols()
    assert "vitmi_gemm" in syms and "vitmi_attn_fwd" in syms and len(syms) >= 15


def test_library_exports_every_declared_symbol(lib):
    from vit_torch_amd import _lib
    raw = ctypes.CDLL(str(_lib.LIB_PATH))
    for s in declared_symbols():
        assert hasattr(raw, s), f"libvitmi.so does not export {s}"
        assert s in _lib.SIGNATURES, f"_lib.SIGNATURES has no binding for {s}"
    assert set(_lib.SIGNATURES) == set(declared_symbols())


def test_version_and_error_string(lib):
    assert lib.vitmi_version() == 100
    assert isinstance(lib.vitmi_last_error_string(), bytes)


def test_bad_arguments_are_rejected_without_a_gpu(lib):
    from vit_torch_amd._lib import GemmDesc
    d = GemmDesc()
    rc = lib.vitmi_gemm(ctypes.byref(d), None)       # M=N=K=0 -> rejected before any launch
    assert rc == -1
    assert b"M,N,K" in lib.vitmi_last_error_string()
    assert lib.vitmi_layernorm_fwd(None, 0, 0, None, None, None, 0, 0, None, None, 4, 8, 1e-6, None) == -1
    ws = lib.vitmi_layernorm_bwd_workspace(50432, 768)       # one partial row of 3*D floats per block
    assert ws % (3 * 768 * 4) == 0 and 256 <= ws // (3 * 768 * 4) <= 2048


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from vit_torch_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(_lib.VitmiError):
        _lib.load()


def test_cpu_tensor_is_refused():
    import torch
    from vit_torch_amd import VisionTransformer, VitmiError
    m = VisionTransformer(img_size=32, patch_size=16, embed_dim=64, depth=1, num_heads=1)
    with pytest.raises(VitmiError):
        m(torch.zeros(1, 3, 32, 32))


def test_hot_shapes_take_the_fast_gemm(lib):
    """ViT-B/16, 256 images x 197 tokens: every large GEMM of the step
    must be eligible for the LDS-DMA kernel, in all three orientations."""
    from vit_torch_amd import ops
    from vit_torch_amd._lib import BF16, EPI_BIAS_GELU, EPI_DGELU, EPI_PATCH_POS, EPI_RESIDUAL, F32
    M, D = 256 * 197, 768
    assert ops.gemm_uses_fast(M, 3 * D, D)                                           # qkv
    assert ops.gemm_uses_fast(M, D, D, epilogue=EPI_RESIDUAL, c_dtype=F32)           # proj
    assert ops.gemm_uses_fast(M, 4 * D, D, epilogue=EPI_BIAS_GELU)                   # fc1
    assert ops.gemm_uses_fast(M, D, 4 * D, epilogue=EPI_RESIDUAL, c_dtype=F32)       # fc2
    assert ops.gemm_uses_fast(M, D, D, epilogue=EPI_PATCH_POS, c_dtype=F32)          # patch embed
    assert ops.gemm_uses_fast(M, 4 * D, D, b_kmajor=False, epilogue=EPI_DGELU)       # fc2 dgrad
    assert ops.gemm_uses_fast(M, D, 3 * D, b_kmajor=False)                           # qkv dgrad
    assert ops.gemm_uses_fast(3 * D, D, M, a_kmajor=False, b_kmajor=False, c_dtype=F32)   # qkv wgrad
    assert ops.gemm_uses_fast(D, 4 * D, M, a_kmajor=False, b_kmajor=False, c_dtype=F32)   # fc2 wgrad
    # fp32 operands, N % 8 != 0 and K % 32 != 0 go to the generic kernel
    assert not ops.gemm_uses_fast(256, 10, D, in_dtype=F32, c_dtype=F32)
    assert not ops.gemm_uses_fast(640, 10, 384)
    assert not ops.gemm_uses_fast(401408, 96, 48)         # Swin patch embed: K = 3*4*4
    # ragged bf16 shapes take the 256x128-tile kernel (clamped loads, masked stores)
    assert ops.gemm_uses_fast(640, 384, 384)              # M not a multiple of 256
    assert ops.gemm_uses_fast(401408, 288, 96)            # Swin-T stage 0 qkv
    assert ops.gemm_uses_fast(96, 96, 401408, a_kmajor=False, b_kmajor=False, c_dtype=F32)
    assert ops.gemm_uses_fast(256 * 5, 1152, 384)         # ViT-S qkv: N % 128 == 0 -> 256x128 tiles
