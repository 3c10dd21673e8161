"""The twelve GEMMs of a ViT-B/16 block at batch 256 as the step issues them (bf16 stream, automatic store policy, gelu'
saved as the derivative, column sums, paired / split-K weight gradients): microseconds per launch.  Meant for A/B of two
builds of the library on one box: VITMI_LIB=<other .so> python tools/step_shapes_bench.py   (tools/lib_ab.sh interleaves)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_torch_amd import _lib, ops  # noqa: E402
from vit_torch_amd._lib import EPI_BIAS_GELU, EPI_DGELU, EPI_RESIDUAL  # noqa: E402

bt = torch.bfloat16
M = 50432
dev = "cuda"


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(bt)


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


cases = []
x768, x3072 = rnd(M, 768), rnd(M, 3072)
for name, N, K, kw in [
        ("qkv fwd", 2304, 768, dict(bias=torch.randn(2304, device=dev))),
        ("proj + res", 768, 768, dict(epilogue=EPI_RESIDUAL, bias=torch.randn(768, device=dev), R=rnd(M, 768))),
        ("fc1 + gelu", 3072, 768, dict(epilogue=EPI_BIAS_GELU, bias=torch.randn(3072, device=dev), C2=torch.empty(M, 3072, device=dev, dtype=bt), aux_deriv=True)),
        ("fc2 + res", 768, 3072, dict(epilogue=EPI_RESIDUAL, bias=torch.randn(768, device=dev), R=rnd(M, 768)))]:
    A = x768 if K == 768 else x3072
    W = rnd(N, K, scale=0.05)
    C = torch.empty(M, N, device=dev, dtype=bt)
    cases.append((f"nt {name:12s} {N:5d} x {K:5d}", 2.0 * M * N * K, (lambda A=A, W=W, C=C, kw=kw: ops.gemm(A, W, C, **kw))))
for name, N, K, kw in [
        ("fc2 dgrad x g'", 3072, 768, dict(epilogue=EPI_DGELU, aux=rnd(M, 3072), aux_deriv=True, colsum_part=torch.empty(M // 128, 3072, device=dev))),
        ("fc1 dgrad", 768, 3072, {}), ("qkv dgrad", 768, 2304, {}), ("proj dgrad", 768, 768, {})]:
    A = rnd(M, K)
    W = rnd(K, N, scale=0.05)
    C = torch.empty(M, N, device=dev, dtype=bt)
    cases.append((f"nn {name:12s} {N:5d} x {K:5d}", 2.0 * M * N * K, (lambda A=A, W=W, C=C, kw=kw: ops.gemm(A, W, C, a_kmajor=True, b_kmajor=False, **kw))))
for name, Mo, No in [("fc2 wgrad", 768, 3072), ("fc1 wgrad", 3072, 768)]:
    A, B = rnd(M, Mo), rnd(M, No)
    C = torch.empty(Mo, No, device=dev)
    cases.append((f"tn {name:12s} {Mo:5d} x {No:5d}", 2.0 * M * Mo * No, (lambda A=A, B=B, C=C: ops.gemm(A, B, C, a_kmajor=False, b_kmajor=False))))
A0, B0, C0 = rnd(M, 768), rnd(M, 768), torch.empty(768, 768, device=dev)
A1, B1, C1 = rnd(M, 2304), rnd(M, 768), torch.empty(2304, 768, device=dev)
cases.append(("tn proj + qkv wgrad (pair)      ", 2.0 * M * 768 * (768 + 2304), lambda: ops.gemm_pair(A0, B0, C0, A1, B1, C1)))

import ctypes  # noqa: E402
raw = ctypes.CDLL(str(_lib.LIB_PATH))
qkv = cases[0]
for pm in (1, 2):                                     # the qkv forward on the other main loops (default: the two-stage loop, PIPE 0)
    def forced(pm=pm, f=qkv[2]):
        raw.vitmi_debug_gemm_pipe(pm)
        f()
        raw.vitmi_debug_gemm_pipe(-1)
    cases.append((f"nt qkv fwd, loop {pm} forced       ", qkv[1], forced))
tag = os.path.basename(str(_lib.LIB_PATH))
tot = 0.0
for rnd_i in range(2):
    tot = 0.0
    for name, fl, f in cases:
        us = timed(f)
        tot += us
        if rnd_i == 1:
            print(f"{tag:18s} {name} {us:7.1f} us  {fl / us / 1e6:7.1f} TFLOP/s", flush=True)
print(f"{tag:18s} block total (incl. the two extra qkv lines) {tot:8.1f} us")
