"""Time the six batched per-(image, head) products of CaiT's talking-heads attention
(cait_S24_224 shapes: B 64, N 196, H 8, hd 48) as the engine issues them."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_torch_amd import ops  # noqa: E402

B, Np, H, hd = int(os.environ.get("BATCH", "64")), 196, 8, 48
D, D3 = H * hd, 3 * H * hd
NS = (Np + 7) // 8 * 8
bt = torch.bfloat16
qkv = torch.randn(B * Np, D3, device="cuda").to(bt)
dO = torch.randn(B * Np, D, device="cuda").to(bt)
S = torch.empty(B * H * Np * NS, device="cuda", dtype=bt)
Pm = torch.randn(B * H * Np * NS, device="cuda").to(bt)
O = torch.empty(B * Np, D, device="cuda", dtype=bt)
dqkv = torch.empty(B * Np, D3, device="cuda", dtype=bt)
cases = {
    "S = q k^T": lambda: ops.gemm_batched(qkv, qkv, S, M=Np, N=Np, K=hd, lda=D3, ldb=D3, ldc=NS, a_kmajor=True, b_kmajor=True,
                                          batch=B * H, batch_inner=H, a_bs=(Np * D3, hd), b_bs=(Np * D3, hd),
                                          c_bs=(H * Np * NS, Np * NS), b_off=H * hd, alpha=0.1),
    "O = P v": lambda: ops.gemm_batched(Pm, qkv, O, M=Np, N=hd, K=Np, lda=NS, ldb=D3, ldc=D, a_kmajor=True, b_kmajor=False,
                                        batch=B * H, batch_inner=H, a_bs=(H * Np * NS, Np * NS), b_bs=(Np * D3, hd),
                                        c_bs=(Np * D, hd), b_off=2 * D),
    "dP = dO v^T": lambda: ops.gemm_batched(dO, qkv, S, M=Np, N=Np, K=hd, lda=D, ldb=D3, ldc=NS, a_kmajor=True, b_kmajor=True,
                                            batch=B * H, batch_inner=H, a_bs=(Np * D, hd), b_bs=(Np * D3, hd),
                                            c_bs=(H * Np * NS, Np * NS), b_off=2 * D),
    "dV = P^T dO": lambda: ops.gemm_batched(Pm, dO, dqkv, M=Np, N=hd, K=Np, lda=NS, ldb=D, ldc=D3, a_kmajor=False, b_kmajor=False,
                                            batch=B * H, batch_inner=H, a_bs=(H * Np * NS, Np * NS), b_bs=(Np * D, hd),
                                            c_bs=(Np * D3, hd), c_off=2 * D),
    "dQ = dS k": lambda: ops.gemm_batched(Pm, qkv, dqkv, M=Np, N=hd, K=Np, lda=NS, ldb=D3, ldc=D3, a_kmajor=True, b_kmajor=False,
                                          batch=B * H, batch_inner=H, a_bs=(H * Np * NS, Np * NS), b_bs=(Np * D3, hd),
                                          c_bs=(Np * D3, hd), b_off=D, alpha=0.1),
    "dK = dS^T q": lambda: ops.gemm_batched(Pm, qkv, dqkv, M=Np, N=hd, K=Np, lda=NS, ldb=D3, ldc=D3, a_kmajor=False, b_kmajor=False,
                                            batch=B * H, batch_inner=H, a_bs=(H * Np * NS, Np * NS), b_bs=(Np * D3, hd),
                                            c_bs=(Np * D3, hd), c_off=D, alpha=0.1),
}
import ctypes
from vit_torch_amd import _lib
raw = ctypes.CDLL(str(_lib.LIB_PATH))
for parts in [int(v) for v in os.environ.get("PARTS", "1,2,4").split(",")]:   # workgroups per problem
    raw.vitmi_debug_gemm_small_parts(parts)
    for name, fn in cases.items():
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        print(f"parts {parts}: {name:14s} {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us")
raw.vitmi_debug_gemm_small_parts(0)
