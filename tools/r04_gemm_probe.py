"""Round-4 probe: is the K = 768 main loop of gemm_fast_kernel paced by the misses of the STREAMED operand?
Every launch is timed as built and with vitmi_debug_gemm_alias (every tile stages operand panel 0: bit 0 = A, bit 1 = B,
so that operand comes out of the XCD's L2 after the first round) — interleaved rounds in one process — and the first
tile's entry -> epilogue span (shader cycles, wave 0) is read from the timeline stamps.
usage: python tools/r04_gemm_probe.py [layout:M:N:K[:epi] ...]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_torch_amd import ops, _lib  # noqa: E402

lib = _lib.load()
raw = ctypes.CDLL(str(_lib.LIB_PATH))
raw.vitmi_debug_gemm_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
EPI = {"store": _lib.EPI_STORE, "gelu": _lib.EPI_BIAS_GELU, "res": _lib.EPI_RESIDUAL, "dgelu": _lib.EPI_DGELU}
specs = sys.argv[1:] or ["nt:50432:768:768", "nt:50432:2304:768", "nt:50432:3072:768:gelu", "nt:50432:768:768:res",
                         "nn:50432:768:768", "nn:50432:3072:768:dgelu", "nt:50432:768:3072:res", "nn:50432:768:3072"]
pool = torch.empty(300 << 20, dtype=torch.uint8, device="cuda")       # flush the Infinity Cache between timed launches


def run(spec, alias, n=10):
    parts = spec.split(":")
    layout, M, N, K = parts[0], *map(int, parts[1:4])
    akm, bkm = {"nt": (True, True), "nn": (True, False), "tn": (False, False)}[layout]
    A = torch.randn((M, K) if akm else (K, M), device="cuda").to(torch.bfloat16)
    B = (torch.randn((N, K) if bkm else (K, N), device="cuda") * 0.05).to(torch.bfloat16)
    epi = parts[4] if len(parts) > 4 else "store"
    C = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    kw = dict(a_kmajor=akm, b_kmajor=bkm, epilogue=EPI[epi])
    if epi == "gelu":
        kw.update(bias=torch.randn(N, device="cuda"), C2=torch.empty_like(C), aux_deriv=True)
    elif epi == "res":
        kw.update(bias=torch.randn(N, device="cuda"), R=torch.randn((M, N), device="cuda").to(torch.bfloat16))
    elif epi == "dgelu":
        kw.update(aux=torch.randn((M, N), device="cuda").to(torch.bfloat16), aux_deriv=True)
    res = {}
    for _ in range(3):
        ops.gemm(A, B, C, **kw)
    for rnd in range(2):
        for al in alias:
            raw.vitmi_debug_gemm_alias(al)
            ops.gemm(A, B, C, **kw)
            ts = []
            for _ in range(n):
                pool.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ops.gemm(A, B, C, **kw); e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            res.setdefault(al, []).append(float(np.median(ts)))
    tl = {}
    nb = 256
    for al in alias:
        raw.vitmi_debug_gemm_alias(al)
        buf = torch.zeros(64 + 8 * nb, dtype=torch.int64, device="cuda")
        raw.vitmi_debug_gemm_timeline(buf.data_ptr(), nb)
        ops.gemm(A, B, C, **kw)
        torch.cuda.synchronize()
        raw.vitmi_debug_gemm_timeline(None, 64)
        t = buf.cpu().numpy()[64:].reshape(nb, 8).astype(np.float64)
        ok = t[:, 0] > 0
        tl[al] = (np.median(t[ok, 1] - t[ok, 0]), np.median(t[ok, 2] - t[ok, 1]))
    raw.vitmi_debug_gemm_alias(0)
    flops = 2.0 * M * N * K
    print(spec)
    for al in alias:
        us = min(res[al])
        print(f"   alias {al}: {res[al][0]:7.1f} / {res[al][1]:7.1f} us  ({flops / us / 1e6:6.0f} TFLOP/s)   first tile: main {tl[al][0]:7.0f}  epilogue {tl[al][1]:6.0f} cycles")


for s in specs:
    run(s, (0, 1, 2, 3))
