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
    assert lib.vitmi_version() == 109
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


# ---- the three statements of struct vitmi_gemm_desc must agree: header, _lib.GemmDesc, and the
# stand-alone ctypes stub a maintainer would copy from INTEGRATION.md (a short struct would
# make the library read past the caller's memory; the library also checks struct_size)
_CTYPE_OF = {"int64_t": "c_int64", "int32_t": "c_int32", "float": "c_float", "size_t": "c_size_t"}


def _header_gemm_fields():
    text = open(os.path.join(ROOT, "include", "vitmi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    body = re.search(r"typedef struct vitmi_gemm_desc \{(.*?)\} vitmi_gemm_desc;", text, flags=re.S).group(1)
    out = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        m = re.match(r"(const )?(\w+)(\*?) (.*)", decl)
        base, ptr, names = m.group(2), m.group(3), m.group(4)
        for nm in names.split(","):
            nm = nm.strip()
            arr = re.match(r"(\w+)\[(\d+)\]", nm)
            is_ptr = bool(ptr) or nm.startswith("*")
            nm = nm.lstrip("* ")
            if arr:
                out.append((arr.group(1), f"{_CTYPE_OF[base]} * {arr.group(2)}"))
            else:
                out.append((nm, "c_void_p" if is_ptr else _CTYPE_OF[base]))
    return out


def _integration_stub_fields():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = text[text.index("class GemmDesc(C.Structure):"):text.index("lib.vitmi_gemm.argtypes")]
    return [(n, t.replace("C.", "")) for n, t in re.findall(r'\("(\w+)", ([^)]+)\)', stub)]


def test_gemm_desc_header_ctypes_and_doc_stub_agree(lib):
    from vit_torch_amd._lib import GemmDesc
    hdr = _header_gemm_fields()
    doc = _integration_stub_fields()
    assert [n for n, _ in hdr] == [n for n, _ in GemmDesc._fields_]
    assert hdr == doc, "INTEGRATION.md ctypes stub differs from struct vitmi_gemm_desc in include/vitmi.h"
    # same layout: build the doc's structure and compare size and every offset
    ns = {"c_int64": ctypes.c_int64, "c_int32": ctypes.c_int32, "c_float": ctypes.c_float,
          "c_size_t": ctypes.c_size_t, "c_void_p": ctypes.c_void_p}

    def ctype(t):
        if "*" in t:
            b, n = t.split("*")
            return ns[b.strip()] * int(n)
        return ns[t]

    Doc = type("Doc", (ctypes.Structure,), {"_fields_": [(n, ctype(t)) for n, t in doc]})
    assert ctypes.sizeof(Doc) == ctypes.sizeof(GemmDesc)
    for n, _ in doc:
        assert getattr(Doc, n).offset == getattr(GemmDesc, n).offset, n


def test_short_gemm_descriptor_is_rejected(lib):
    from vit_torch_amd._lib import GemmDesc
    d = GemmDesc()
    d.M = d.N = d.K = 256
    d.struct_size = 232              # the layout that ended at workspace_bytes
    assert lib.vitmi_gemm(ctypes.byref(d), None) == -1
    assert b"struct_size" in lib.vitmi_last_error_string()


def test_gemm_kernels_keep_their_accumulators_in_registers(lib):
    """hipcc's resource remarks of the build: no gemm_fast_kernel instantiation may own more than a few spilled dwords
    of scratch.  A comparison chain over an unrolled accumulator index once came back as a dynamically indexed array
    (24 accumulator quads = 400 B of scratch in the fp32-residual fold loop): numerically identical, 10 x slower."""
    from vit_torch_amd import build
    build.build()
    res = build.kernel_resources("gemm_fast.hip")
    kernels = {k: v for k, v in res.items() if "gemm_fast_kernel" in k}
    assert len(kernels) > 40
    for name, r in kernels.items():
        assert int(r["ScratchSize [bytes/lane]"]) <= 192, (name, r["ScratchSize [bytes/lane]"])
        assert int(r["Occupancy [waves/SIMD]"]) >= 2, name


# ------------------------------------------------------------------ libvitmi_comm.so (include/vitmi_comm.h) ---
def _comm_declared():
    text = open(os.path.join(ROOT, "include", "vitmi_comm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vitmi_comm_[a-z0-9_]+)\s*\(", text)))


def test_comm_library_exports_every_declared_symbol_and_binds_rccl():
    """The gradient-exchange library: header <-> exports <-> ctypes table agree; RCCL is bound at run time from the copy
    the process already has (PyTorch's), and reports its version.  No communicator is created without a GPU."""
    import ctypes
    from vit_torch_amd import build, comm
    if not comm.LIB_PATH.exists():
        build.build_comm()
    declared = _comm_declared()
    assert len(declared) >= 10
    assert sorted(comm.SIGNATURES) == declared
    raw = ctypes.CDLL(str(comm.LIB_PATH))
    for name in declared:
        assert hasattr(raw, name), f"libvitmi_comm.so does not export {name}"
    lib = comm.load()
    assert lib.vitmi_comm_version() == 1
    v = comm.rccl_version()
    assert 20000 <= v < 30000, v
    # calls on a null communicator are refused with a message, not a crash
    assert lib.vitmi_comm_join(None, None) == -1 and b"null communicator" in lib.vitmi_comm_last_error()
    assert lib.vitmi_comm_allreduce_sum_f32_async(None, None, 0, None) == -1


def test_comm_library_is_not_linked_against_a_second_rccl():
    """libvitmi_comm.so must not carry a DT_NEEDED on librccl: a second copy beside PyTorch's would own a second set of
    device-side state.  (readelf is part of binutils; skip if it is not there.)"""
    import shutil
    import subprocess
    from vit_torch_amd import build, comm
    if shutil.which("readelf") is None:
        import pytest
        pytest.skip("readelf not installed")
    if not comm.LIB_PATH.exists():
        build.build_comm()
    out = subprocess.run(["readelf", "-d", str(comm.LIB_PATH)], capture_output=True, text=True).stdout
    needed = re.findall(r"\(NEEDED\)\s+Shared library: \[(.*?)\]", out)
    assert needed and not any("rccl" in n or "nccl" in n for n in needed), needed
