"""A/B of any integer vitmi_debug_* switch on GEMM shapes, interleaved rounds in one process.
usage: python tools/hook_ab.py <hook> <v0,v1,...> [layout:M:N:K[:epi[:f32]] ...]
e.g.   python tools/hook_ab.py gemm_side_prefetch 0,1,2          (default shapes: the ViT-B/16 epilogues with a side input)
The micro-benchmark cycles each shape's side input through a pool of buffers larger than the Infinity Cache, as the step
does (a single buffer would sit in the 256-MB cache and hide what a prefetch buys)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_torch_amd import ops, _lib  # noqa: E402

_lib.load()
raw = ctypes.CDLL(str(_lib.LIB_PATH))
EPI = {"store": _lib.EPI_STORE, "gelu": _lib.EPI_BIAS_GELU, "res": _lib.EPI_RESIDUAL, "dgelu": _lib.EPI_DGELU}
hook = getattr(raw, "vitmi_debug_" + sys.argv[1])
values = [int(v) for v in sys.argv[2].split(",")]
SHAPES = sys.argv[3:] or ["nt:50432:768:768:res", "nt:50432:768:3072:res", "nn:50432:3072:768:dgelu"]
POOL = 6


def make(spec):
    parts = spec.split(":")
    layout, M, N, K = parts[0], int(parts[1]), int(parts[2]), int(parts[3])
    epi = parts[4] if len(parts) > 4 else "store"
    cdt = torch.float32 if (len(parts) > 5 and parts[5] == "f32") else torch.bfloat16
    akm, bkm = {"nt": (True, True), "nn": (True, False), "tn": (False, False)}[layout]
    bt = torch.bfloat16
    As = [torch.randn((M, K) if akm else (K, M), device="cuda").to(bt) for _ in range(POOL)]
    B = (torch.randn((N, K) if bkm else (K, N), device="cuda") * 0.05).to(bt)
    Cs = [torch.empty((M, N), device="cuda", dtype=cdt) for _ in range(POOL)]
    sides = [torch.randn((M, N), device="cuda").to(cdt if epi == "res" else bt) for _ in range(POOL)]
    bias = torch.randn(N, device="cuda")
    state = {"i": 0}

    def f():
        i = state["i"] = (state["i"] + 1) % POOL
        kw = {}
        if epi == "gelu":
            kw = dict(bias=bias, C2=sides[i], aux_deriv=True)
        elif epi == "res":
            kw = dict(bias=bias, R=sides[i])
        elif epi == "dgelu":
            kw = dict(aux=sides[i], aux_deriv=True)
        ops.gemm(As[i], B, Cs[i], a_kmajor=akm, b_kmajor=bkm, epilogue=EPI[epi], **kw)
    return f, 2.0 * M * N * K


def timed(f, n=24):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for spec in SHAPES:
    f, flop = make(spec)
    for _ in range(6):
        f()
    res = {v: [] for v in values}
    for rnd in range(4):
        for v in values:
            hook(v)
            res[v].append(timed(f))
    hook(values[0])
    print(f"{spec:30s} " + "  ".join(f"{v}: {min(r):6.1f}/{sorted(r)[len(r)//2]:6.1f} us" for v, r in res.items()), flush=True)
    del f
    torch.cuda.empty_cache()
