"""The three fp32 products of the classifier head (ViT-B/16: 256 x 10 x 768) on the skinny kernels and on the 64x64-tile kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_torch_amd import ops
from vit_torch_amd._lib import GEMM_AUTO, GEMM_GENERIC
B, D, K = 256, 768, 10
feat, W, d = torch.randn(B, D, device="cuda"), torch.randn(K, D, device="cuda"), torch.randn(B, K, device="cuda")
out, dW, dx = torch.empty(B, K, device="cuda"), torch.empty(K, D, device="cuda"), torch.empty(B, D, device="cuda")
cases = {"logits = feat W^T": lambda impl: ops.gemm(feat, W, out, impl=impl),
         "dW = d^T feat": lambda impl: ops.gemm(d, feat, dW, a_kmajor=False, b_kmajor=False, impl=impl),
         "dfeat = d W": lambda impl: ops.gemm(d, W, dx, b_kmajor=False, impl=impl)}
for name, f in cases.items():
    for impl, tag in ((GEMM_AUTO, "skinny"), (GEMM_GENERIC, "generic")):
        for _ in range(5):
            f(impl)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            f(impl)
        e1.record(); torch.cuda.synchronize()
        print(f"{name:20s} {tag:8s} {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us")
