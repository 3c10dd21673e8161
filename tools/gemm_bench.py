"""GPU micro-benchmark of vitmi_gemm on chosen shapes (random bf16 operands).
usage: python tools/gemm_bench.py [shape ...]   shape = layout:M:N:K[:epi[:cdt]]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_torch_amd import ops  # noqa: E402
from vit_torch_amd._lib import EPI_BIAS_GELU, EPI_DGELU, EPI_RESIDUAL, EPI_STORE  # noqa: E402

EPI = {"store": EPI_STORE, "gelu": EPI_BIAS_GELU, "res": EPI_RESIDUAL, "dgelu": EPI_DGELU}
DEFAULT = ["nt:4096:4096:4096", "nt:8192:8192:8192", "nn:4096:4096:4096", "tn:4096:4096:4096",
           "nt:50432:2304:768", "nt:50432:3072:768:gelu", "nt:50432:768:768:res:f32",
           "nt:50432:768:3072:res:f32", "nt:50432:768:3072:res:bf16",
           "nn:50432:3072:768:dgelu", "nn:50432:768:3072", "nn:50432:768:2304", "nn:50432:768:768",
           "tn:768:768:50432:store:f32", "tn:3072:768:50432:store:f32", "tn:768:3072:50432:store:f32",
           "tn:2304:768:50432:store:f32"]


def run(spec, iters=20):
    parts = spec.split(":")
    layout, M, N, K = parts[0], int(parts[1]), int(parts[2]), int(parts[3])
    epi = parts[4] if len(parts) > 4 else "store"
    cdt = torch.float32 if (len(parts) > 5 and parts[5] == "f32") else torch.bfloat16
    akm, bkm = {"nt": (True, True), "nn": (True, False), "tn": (False, False)}[layout]
    bt = torch.bfloat16
    A = torch.randn((M, K) if akm else (K, M), device="cuda").to(bt)
    B = (torch.randn((N, K) if bkm else (K, N), device="cuda") * 0.05).to(bt)
    C = torch.empty((M, N), device="cuda", dtype=cdt)
    kw = {}
    if epi == "gelu":
        kw = dict(bias=torch.randn(N, device="cuda"), C2=torch.empty_like(C))
    elif epi == "res":
        kw = dict(bias=torch.randn(N, device="cuda"), R=torch.randn((M, N), device="cuda").to(cdt))
    elif epi == "dgelu":
        kw = dict(aux=torch.randn((M, N), device="cuda").to(bt))
    f = lambda: ops.gemm(A, B, C, a_kmajor=akm, b_kmajor=bkm, epilogue=EPI[epi], **kw)
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{spec:36s} {ms*1e3:9.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    import ctypes
    from vit_torch_amd import _lib
    raw = ctypes.CDLL(str(_lib.LIB_PATH))
    specs = [a for a in sys.argv[1:] if not a.startswith("pipe=")]
    specs = [a for a in specs if not a.startswith("tile=")]
    persists = [int(a[8:]) for a in specs if a.startswith("persist=")] or [1]
    specs = [a for a in specs if not a.startswith("persist=")]
    bands = [int(a[5:]) for a in specs if a.startswith("band=")] or [-1]
    specs = [a for a in specs if not a.startswith("band=")]
    pols = [int(a[4:]) for a in specs if a.startswith("pol=")] or [0]
    specs = [a for a in specs if not a.startswith("pol=")]
    pipes = [int(a[5:]) for a in sys.argv[1:] if a.startswith("pipe=")] or [-1]
    tiles = [int(a[5:]) for a in sys.argv[1:] if a.startswith("tile=")] or [-1]
    for tm, pm, ps, bd, pol in [(t, p, q, b, c) for t in tiles for p in pipes for q in persists for b in bands for c in pols]:
        raw.vitmi_debug_gemm_store_policy(pol)
        print(f"--- store policy {pol}")
        raw.vitmi_debug_gemm_tile(tm)
        raw.vitmi_debug_gemm_pipe(pm)
        raw.vitmi_debug_gemm_persist(ps)
        raw.vitmi_debug_gemm_band(bd)
        print(f"--- tile {tm} pipe {pm} persistent {ps} band {bd}")
        for s in (specs or DEFAULT):
            run(s)
