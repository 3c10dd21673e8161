"""A/B of the residual fold (gemm_fast.hip ResFold) on the two residual GEMMs of a ViT-B block,
interleaved rounds in one process (guide §5.4 rule 24)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_torch_amd import _lib, ops  # noqa: E402
from vit_torch_amd._lib import EPI_RESIDUAL  # noqa: E402

raw = ctypes.CDLL(str(_lib.LIB_PATH))
M = 50432


def case(N, K):
    A = torch.randn(M, K, device="cuda").bfloat16()
    B = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    R = torch.randn(M, N, device="cuda")
    C = torch.empty(M, N, device="cuda")
    bias = torch.randn(N, device="cuda")
    return lambda: ops.gemm(A, B, C, epilogue=EPI_RESIDUAL, bias=bias, R=R)


def timeit(f, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for N, K in ((768, 768), (768, 3072)):
    f = case(N, K)
    modes = {"epilogue-read (PIPE 2)": (0, -1), "ring-3, epilogue-read": (0, 3), "ring-3 + fold (PIPE 3)": (1, -1)}
    res = {k: [] for k in modes}
    for rnd in range(5):
        for k, (rfold, pipe) in modes.items():
            raw.vitmi_debug_gemm_rfold(rfold)
            raw.vitmi_debug_gemm_pipe(pipe)
            f(); f()
            res[k].append(timeit(f))
    raw.vitmi_debug_gemm_rfold(-1)
    raw.vitmi_debug_gemm_pipe(-1)
    fl = 2.0 * M * N * K
    for k in modes:
        med = sorted(res[k])[len(res[k]) // 2]
        print(f"nt:{M}:{N}:{K}:res:f32 {k:24s}: median {med:7.1f} us  min {min(res[k]):7.1f} us  {fl / med / 1e6:7.1f} TFLOP/s", flush=True)
