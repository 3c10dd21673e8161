"""Shader clock under the GEMM kernels, two independent methods on the SAME dispatches:
 (1) in-kernel: every workgroup stamps s_memtime (shader cycles) and s_memrealtime (100 MHz wall clock) at entry and
     exit; clock = d(cycles) / d(wall), median over the workgroups (guide: DVFS give-back, item 6);
 (2) counters: GRBM_GUI_ACTIVE / 8 XCDs / kernel wall time from `rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE
     SQ_VALU_MFMA_BUSY_CYCLES` of this script (tools/clock_check.sh parses the CSVs); valid for dispatches of >= 10 ms
     (the guide: the quotient reads high below ~0.3 ms), so the shapes here are the ViT-B/16 GEMM shapes with M x 16.
Prints one line per shape: in-kernel clock and the launch's wall time; under rocprofv3 the dispatch ids follow in order.
usage: python tools/clock_probe.py [scale]      (scale 16: ~10-25 ms per dispatch; scale 1: the real step shapes)"""
import ctypes, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_torch_amd import ops, _lib  # noqa: E402

_lib.load()
raw = ctypes.CDLL(str(_lib.LIB_PATH))
raw.vitmi_debug_gemm_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
raw.vitmi_debug_gemm_tail(0)
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 16
EPI = {"store": _lib.EPI_STORE, "gelu": _lib.EPI_BIAS_GELU, "res": _lib.EPI_RESIDUAL, "dgelu": _lib.EPI_DGELU}
SHAPES = ["nt:50432:3072:768:gelu", "nt:50432:2304:768", "nn:50432:3072:768:dgelu", "nn:50432:768:3072", "nt:50432:768:3072:res"]
# >= 2 s of back-to-back GEMM launches first: the power management needs that long to settle at the clock it holds
# under sustained load (the first dispatches of a cold process ran 30 % faster than the steady state)
_A = torch.randn(8192, 8192, device="cuda").to(torch.bfloat16)
_C = torch.empty(8192, 8192, device="cuda", dtype=torch.bfloat16)
import time
_t = time.time()
while time.time() - _t < 2.5:
    for _ in range(20):
        ops.gemm(_A, _A, _C)
    torch.cuda.synchronize()
del _A, _C
out = []
for spec in SHAPES:
    parts = spec.split(":")
    layout, M, N, K = parts[0], int(parts[1]) * scale, int(parts[2]), int(parts[3])
    epi = parts[4] if len(parts) > 4 else "store"
    akm, bkm = {"nt": (True, True), "nn": (True, False)}[layout]
    bt = torch.bfloat16
    A = torch.randn((M, K), device="cuda").to(bt)
    B = (torch.randn((N, K) if bkm else (K, N), device="cuda") * 0.05).to(bt)
    C = torch.empty((M, N), device="cuda", dtype=bt)
    kw = dict(a_kmajor=akm, b_kmajor=bkm, epilogue=EPI[epi])
    if epi == "gelu":
        kw.update(bias=torch.randn(N, device="cuda"), C2=torch.empty_like(C), aux_deriv=True)
    elif epi == "res":
        kw.update(bias=torch.randn(N, device="cuda"), R=torch.randn((M, N), device="cuda").to(bt))
    elif epi == "dgelu":
        kw.update(aux=torch.randn((M, N), device="cuda").to(bt), aux_deriv=True)
    for _ in range(3):
        ops.gemm(A, B, C, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        ops.gemm(A, B, C, **kw)                      # dispatches the counters are read on
    e1.record()
    torch.cuda.synchronize()
    wall_ms = e0.elapsed_time(e1) / 3
    nb = 256
    buf = torch.zeros(64 + 8 * nb, dtype=torch.int64, device="cuda")
    raw.vitmi_debug_gemm_timeline(buf.data_ptr(), nb)
    ops.gemm(A, B, C, **kw)                          # the stamped dispatch
    torch.cuda.synchronize()
    raw.vitmi_debug_gemm_timeline(None, 64)
    t = buf.cpu().numpy()[64:].reshape(nb, 8).astype(np.float64)
    ok = (t[:, 6] > t[:, 4]) & (t[:, 5] > t[:, 0])
    ghz = (t[ok, 5] - t[ok, 0]) / ((t[ok, 6] - t[ok, 4]) * 10.0)
    rec = {"shape": spec, "M": M, "wall_ms": round(wall_ms, 3), "tflops": round(2.0 * M * N * K / wall_ms / 1e9, 1),
           "clock_mhz_in_kernel": round(float(np.median(ghz)) * 1e3, 1), "workgroups": int(ok.sum())}
    out.append(rec)
    print(json.dumps(rec), flush=True)
    del A, B, C, kw
    torch.cuda.empty_cache()
