"""A/B of the epilogue side-input prefetch depth (vitmi_debug_gemm_side_depth 1 | 3) on the ViT-B/16 shapes whose epilogue
reads a per-element side input, as the step issues them (automatic store policy, gelu' saved as the derivative, column sums);
interleaved rounds in one process.  usage: python tools/side_depth_ab.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_torch_amd import _lib, ops  # noqa: E402
from vit_torch_amd._lib import EPI_BIAS_GELU, EPI_DGELU, EPI_RESIDUAL  # noqa: E402

raw = ctypes.CDLL(str(_lib.LIB_PATH))
bt = torch.bfloat16
M = 50432


def make(layout, N, K, epi):
    akm, bkm = {"nt": (True, True), "nn": (True, False)}[layout]
    A = torch.randn((M, K), device="cuda").to(bt)
    B = (torch.randn((N, K) if bkm else (K, N), device="cuda") * 0.05).to(bt)
    C = torch.empty((M, N), device="cuda", dtype=bt)
    if epi == "gelu":
        C2 = torch.empty((M, N), device="cuda", dtype=bt)
        kw = dict(epilogue=EPI_BIAS_GELU, bias=torch.randn(N, device="cuda"), C2=C2, aux_deriv=True)
        return lambda: ops.gemm(A, B, C, a_kmajor=akm, b_kmajor=bkm, **kw), (C, C2)
    if epi == "store":
        bias = torch.randn(N, device="cuda") if layout == "nt" else None
        return lambda: ops.gemm(A, B, C, a_kmajor=akm, b_kmajor=bkm, bias=bias), (C,)
    if epi == "res":
        kw = dict(epilogue=EPI_RESIDUAL, bias=torch.randn(N, device="cuda"), R=torch.randn((M, N), device="cuda").to(bt))
    else:
        kw = dict(epilogue=EPI_DGELU, aux=torch.randn((M, N), device="cuda").to(bt), aux_deriv=True,
                  colsum_part=torch.empty((M // 128, N), device="cuda"))
    return lambda: ops.gemm(A, B, C, a_kmajor=akm, b_kmajor=bkm, **kw), (C,)


def timed(f, n=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, (layout, N, K, epi) in {"fc1 + gelu, gelu' (nt 3072 x 768)": ("nt", 3072, 768, "gelu"),
                                  "qkv forward (nt 2304 x 768)": ("nt", 2304, 768, "store"),
                                  "fc1 dgrad (nn 768 x 3072)": ("nn", 768, 3072, "store"),
                                  "proj dgrad (nn 768 x 768)": ("nn", 768, 768, "store"),
                                  "fc2 dgrad x gelu' (nn 3072 x 768)": ("nn", 3072, 768, "dgelu"),
                                  "proj + bf16 residual (nt 768 x 768)": ("nt", 768, 768, "res"),
                                  "fc2 + bf16 residual (nt 768 x 3072)": ("nt", 768, 3072, "res")}.items():
    f, C = make(layout, N, K, epi)
    outs = {}
    for rnd in range(3):
        for d in (1, 3):
            raw.vitmi_debug_gemm_side_depth(d)
            us = timed(f)
            outs[d] = [c.clone() for c in C]
            print(f"{name:40s} depth {d}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
    print("   bit-identical outputs:", all(torch.equal(a, b) for a, b in zip(outs[1], outs[3])),
          " max |diff|:", [float((a.float() - b.float()).abs().max()) for a, b in zip(outs[1], outs[3])],
          " differing elements:", [int((a != b).sum()) for a, b in zip(outs[1], outs[3])], "of", outs[1][0].numel())
raw.vitmi_debug_gemm_side_depth(3)
