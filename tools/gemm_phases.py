"""Diagnostic: where do the cycles of the PIPE=1 GEMM main loop go?  Block 0's waves
accumulate s_memtime stamps (R phase, wait at the barrier after R, M phase, wait at
the barrier after M); stamps are only executed when this tool arms them."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_torch_amd import ops, _lib  # noqa: E402

lib = _lib.load()
raw = ctypes.CDLL(str(_lib.LIB_PATH))
raw.vitmi_debug_gemm_stamps.argtypes = [ctypes.c_void_p]
buf = torch.zeros(64 + 64 * 4, dtype=torch.int64, device="cuda")
raw.vitmi_debug_gemm_pipe(int(os.environ.get("PIPE", "-1")))
EPI = {"store": _lib.EPI_STORE, "gelu": _lib.EPI_BIAS_GELU, "res": _lib.EPI_RESIDUAL, "dgelu": _lib.EPI_DGELU}
for spec in (sys.argv[1:] or ["nt:8192:8192:8192", "tn:4096:4096:4096", "nt:50432:2304:768", "nn:50432:768:3072"]):
    layout, M, N, K = spec.split(":")[0], *map(int, spec.split(":")[1:4])
    akm, bkm = {"nt": (True, True), "nn": (True, False), "tn": (False, False)}[layout]
    A = torch.randn((M, K) if akm else (K, M), device="cuda").to(torch.bfloat16)
    B = torch.randn((N, K) if bkm else (K, N), device="cuda").to(torch.bfloat16)
    parts = spec.split(":")
    epi = parts[4] if len(parts) > 4 else "store"
    cdt = torch.float32 if (len(parts) > 5 and parts[5] == "f32") else torch.bfloat16
    C = torch.empty((M, N), device="cuda", dtype=cdt)
    kw = dict(a_kmajor=akm, b_kmajor=bkm, epilogue=EPI[epi])
    if epi == "gelu":
        kw.update(bias=torch.randn(N, device="cuda"), C2=torch.empty_like(C))
    elif epi == "res":
        kw.update(bias=torch.randn(N, device="cuda"), R=torch.randn((M, N), device="cuda").to(cdt))
    elif epi == "dgelu":
        kw.update(aux=torch.randn((M, N), device="cuda").to(torch.bfloat16))
    for _ in range(2):
        ops.gemm(A, B, C, **kw)
    buf.zero_()
    raw.vitmi_debug_gemm_stamps(buf.data_ptr())
    ops.gemm(A, B, C, **kw)
    torch.cuda.synchronize()
    raw.vitmi_debug_gemm_stamps(None)
    b = buf.cpu().tolist()
    ns = b[32]
    print(f"{spec}: slabs={ns}; cycles per slab (R work, wait after R, M work, wait after M) — ideal M = 512")
    for w in range(8):
        r, wr, m, wm = (b[w * 4 + i] / max(ns, 1) for i in range(4))
        print(f"  wave {w} (group {w >> 2}): R {r:7.1f}  waitR {wr:7.1f}  M {m:7.1f}  waitM {wm:7.1f}  sum {r + wr + m + wm:7.1f}")
    tl = [(b[64 + i * 4], b[64 + i * 4 + 1], b[64 + i * 4 + 2]) for i in range(64) if b[64 + i * 4]]
    if tl:
        main = sorted(t[1] - t[0] for t in tl)
        epi_c = sorted(t[2] - t[1] for t in tl)
        print(f"  timeline of {len(tl)} blocks (cycles): entry->epilogue median {main[len(main)//2]} "
              f"(min {main[0]}, max {main[-1]}); epilogue median {epi_c[len(epi_c)//2]} (min {epi_c[0]}, max {epi_c[-1]})")
