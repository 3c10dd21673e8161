"""A/B of the start stagger (vitmi_debug_gemm_stagger) on the ViT-B/16 GEMM shapes, interleaved rounds in one process.
usage: python tools/stagger_ab.py [permille:phases ...]   (default 0:1 500:1 500:2 660:3)"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_torch_amd import ops, _lib  # noqa: E402

_lib.load()
raw = ctypes.CDLL(str(_lib.LIB_PATH))
EPI = {"store": _lib.EPI_STORE, "gelu": _lib.EPI_BIAS_GELU, "res": _lib.EPI_RESIDUAL, "dgelu": _lib.EPI_DGELU}
SHAPES = ["nt:50432:3072:768:gelu", "nn:50432:3072:768:dgelu", "nt:50432:768:768:res", "nt:50432:768:768:res:f32",
          "nn:50432:768:768", "nt:50432:2304:768", "nt:50432:768:3072:res", "nn:50432:768:3072", "nn:50432:768:2304"]
settings = [tuple(map(int, a.split(":"))) for a in sys.argv[1:]] or [(0, 1), (500, 1), (500, 2), (660, 3)]


def make(spec):
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
        kw = dict(bias=torch.randn(N, device="cuda"), C2=torch.empty_like(C), aux_deriv=True)
    elif epi == "res":
        kw = dict(bias=torch.randn(N, device="cuda"), R=torch.randn((M, N), device="cuda").to(cdt))
    elif epi == "dgelu":
        kw = dict(aux=torch.randn((M, N), device="cuda").to(bt), aux_deriv=True)
    return lambda: ops.gemm(A, B, C, a_kmajor=akm, b_kmajor=bkm, epilogue=EPI[epi], **kw), 2.0 * M * N * K


def timed(f, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for spec in SHAPES:
    f, flop = make(spec)
    for _ in range(5):
        f()
    res = {s: [] for s in settings}
    for rnd in range(4):
        for s in settings:
            raw.vitmi_debug_gemm_stagger(s[0])
            raw.vitmi_debug_gemm_stagger_phases(s[1])
            res[s].append(timed(f))
    raw.vitmi_debug_gemm_stagger(0)
    print(f"{spec:32s} " + "  ".join(f"{s[0]}:{s[1]} {min(v):6.1f}/{sorted(v)[len(v)//2]:6.1f} us" for s, v in res.items()), flush=True)
