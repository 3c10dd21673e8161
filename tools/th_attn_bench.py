"""Micro-benchmark: fused talking-heads attention (vitmi_th_attn_fwd / _bwd + the two remaining batched products) against
the three-call form, at cait_S24_224's shape (B images, N = 196, H = 8, hd = 48).  usage: python tools/th_attn_bench.py [B]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_torch_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N, H, hd = 196, 8, 48
D, D3, NS = H * hd, 3 * H * hd, 224
bt = torch.bfloat16
dev = "cuda"
qkv = (torch.randn(B * N, D3, device=dev) * 0.7).to(bt)
dO = torch.randn(B * N, D, device=dev).to(bt)
O = torch.empty(B * N, D, device=dev, dtype=bt)
Wl = (torch.eye(H) + 0.3 * torch.randn(H, H)).to(dev); Ww = (torch.eye(H) + 0.3 * torch.randn(H, H)).to(dev)
bl, bw = (0.2 * torch.randn(H)).to(dev), (0.05 * torch.randn(H)).to(dev)
S = torch.empty(B, H, N, NS, device=dev, dtype=bt); P = torch.empty_like(S); Pm = torch.empty_like(S)
dPm = torch.empty_like(S); dS = torch.empty_like(S)
dqkv = torch.empty_like(qkv)
g = [torch.empty(H, H, device=dev), torch.empty(H, device=dev), torch.empty(H, H, device=dev), torch.empty(H, device=dev)]
scale = hd ** -0.5
bs = dict(batch=B * H, batch_inner=H)


def timed(f, n=10):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def fwd3():
    ops.gemm_batched(qkv, qkv, S, M=N, N=N, K=hd, lda=D3, ldb=D3, ldc=NS, a_kmajor=True, b_kmajor=True, a_bs=(N * D3, hd), b_bs=(N * D3, hd), c_bs=(H * N * NS, N * NS), b_off=H * hd, alpha=scale, **bs)
    ops.th_softmax_fwd(S, Wl, bl, Ww, bw, P, Pm, B, H, N, N, NS)
    ops.gemm_batched(Pm, qkv, O, M=N, N=hd, K=N, lda=NS, ldb=D3, ldc=D, a_kmajor=True, b_kmajor=False, a_bs=(H * N * NS, N * NS), b_bs=(N * D3, hd), c_bs=(N * D, hd), b_off=2 * D, **bs)


def dkdv():
    ops.gemm_batched(Pm, dO, dqkv, M=N, N=hd, K=N, lda=NS, ldb=D, ldc=D3, a_kmajor=False, b_kmajor=False, a_bs=(H * N * NS, N * NS), b_bs=(N * D, hd), c_bs=(N * D3, hd), c_off=2 * D, **bs)
    ops.gemm_batched(dS, qkv, dqkv, M=N, N=hd, K=N, lda=NS, ldb=D3, ldc=D3, a_kmajor=False, b_kmajor=False, a_bs=(H * N * NS, N * NS), b_bs=(N * D3, hd), c_bs=(N * D3, hd), c_off=D, alpha=scale, **bs)


def bwd3():
    ops.gemm_batched(dO, qkv, dPm, M=N, N=N, K=hd, lda=D, ldb=D3, ldc=NS, a_kmajor=True, b_kmajor=True, a_bs=(N * D, hd), b_bs=(N * D3, hd), c_bs=(H * N * NS, N * NS), b_off=2 * D, **bs)
    ops.th_softmax_bwd(S, P, dPm, Wl, Ww, dS, *g, B, H, N, N, NS)
    ops.gemm_batched(dS, qkv, dqkv, M=N, N=hd, K=N, lda=NS, ldb=D3, ldc=D3, a_kmajor=True, b_kmajor=False, a_bs=(H * N * NS, N * NS), b_bs=(N * D3, hd), c_bs=(N * D3, hd), b_off=D, alpha=scale, **bs)
    dkdv()


def fwdf():
    ops.th_attn_fwd(qkv, Wl, bl, Ww, bw, O, B, H, N, hd, scale)


def bwdf_k():
    ops.th_attn_bwd(qkv, dO, Wl, bl, Ww, bw, dqkv, dS, Pm, NS, *g, B, H, N, hd, scale)


def bwdf():
    bwdf_k()


print(f"B={B} N={N} H={H} hd={hd}: us per layer")
print(f"  forward : three calls {timed(fwd3):7.1f}   fused {timed(fwdf):7.1f}")
print(f"  backward: three calls + 4 products {timed(bwd3):7.1f}   fused (pack + row kernel + products kernel) {timed(bwdf):7.1f}")

import ctypes, numpy as np
from vit_torch_amd import _lib
raw = ctypes.CDLL(str(_lib.LIB_PATH))
raw.vitmi_debug_th_attn_stamps.argtypes = [ctypes.c_void_p]
for name, f, nwg, cols in (("forward", fwdf, B, [1, 2, 3, 4, 5, 6]), ("backward", bwdf_k, 8 * ((B + 7) // 8) * 13, [1, 2, 3, 4, 5, 7])):
    buf = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
    raw.vitmi_debug_th_attn_stamps(buf.data_ptr())
    f()
    torch.cuda.synchronize()
    raw.vitmi_debug_th_attn_stamps(None)
    t = buf.cpu().numpy().reshape(nwg, 8).astype(np.float64)
    t = t[nwg // 3: 2 * nwg // 3]
    d = np.diff(t[:, cols], axis=1)
    print(f"  {name}, first block of a workgroup, cycles (S | wait | R | wait | last phase): " + "  ".join(f"{x:.0f}" for x in np.median(d, axis=0))
          + f"   whole workgroup {np.median(t[:, 7] - t[:, 0]):.0f}")
