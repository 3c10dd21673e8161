"""What the fc1 epilogue costs beyond a plain store (ViT-B/16 bs 256 shape): bias only, GELU with one output
(inference), GELU + pre-activation, GELU + derivative.  One process, interleaved rounds."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_torch_amd import ops
from vit_torch_amd._lib import EPI_BIAS_GELU

M, N, K = 50432, 3072, 768
bt = torch.bfloat16
A = torch.randn(M, K, device="cuda").to(bt)
B = (torch.randn(N, K, device="cuda") * 0.05).to(bt)
bias = torch.randn(N, device="cuda")
C, C2 = torch.empty(M, N, device="cuda", dtype=bt), torch.empty(M, N, device="cuda", dtype=bt)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


cases = {
    "store + bias": lambda: ops.gemm(A, B, C, bias=bias),
    "gelu, one output": lambda: ops.gemm(A, B, C, epilogue=EPI_BIAS_GELU, bias=bias),
    "gelu + pre": lambda: ops.gemm(A, B, C, epilogue=EPI_BIAS_GELU, bias=bias, C2=C2),
    "gelu + gelu'": lambda: ops.gemm(A, B, C, epilogue=EPI_BIAS_GELU, bias=bias, C2=C2, aux_deriv=True),
}
for rnd in range(3):
    print("round", rnd, "  ".join(f"{k}: {timed(f):6.1f} us" for k, f in cases.items()), flush=True)
