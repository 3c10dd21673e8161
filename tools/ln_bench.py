"""LayerNorm forward / backward micro-benchmark: 4-element kernels (ln8 = 0) against the 8-element form (ln8 = 1),
interleaved rounds in one process; bytes counted as the kernels' header does.  usage: python tools/ln_bench.py [M:D:stream ...]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_torch_amd import ops, _lib  # noqa: E402

_lib.load()
raw = ctypes.CDLL(str(_lib.LIB_PATH))
specs = sys.argv[1:] or ["50432:768:bf16", "50432:768:fp32", "802816:96:bf16", "200704:192:bf16", "50176:384:bf16", "50432:384:fp32"]
bt = torch.bfloat16


def timed(f, n=30):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for sp in specs:
    M, D, stream = sp.split(":")
    M, D = int(M), int(D)
    R = bt if stream == "bf16" else torch.float32
    x = torch.randn(M, D, device="cuda").to(R)
    g, b = torch.randn(D, device="cuda"), torch.randn(D, device="cuda")
    y = torch.empty(M, D, device="cuda", dtype=bt)
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    dy = torch.randn(M, D, device="cuda").to(bt)
    G = torch.randn(M, D, device="cuda").to(R)
    Gb = None if R == bt else torch.empty(M, D, device="cuda", dtype=bt)
    dg, db, gsum = (torch.empty(D, device="cuda") for _ in range(3))
    fwd = lambda: ops.layernorm_fwd(x, g, b, y, mean, rstd, 1e-6, M=M, D=D)
    bwd = lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, G, G, Gb, dg, db, gsum=gsum, M=M, D=D)
    es = 2 if R == bt else 4
    fb = M * D * (es + 2)
    bb = M * D * (2 + es + 2 * es + (0 if R == bt else 2))
    res = {}
    for rnd in range(3):
        for mode in (0, 1):
            raw.vitmi_debug_ln8(mode)
            res.setdefault(("fwd", mode), []).append(timed(fwd))
            res.setdefault(("bwd", mode), []).append(timed(bwd))
    raw.vitmi_debug_ln8(1)
    print(f"{sp:18s} fwd {fb/1e6:6.0f} MB: 4-elt {min(res[('fwd',0)]):6.1f} us ({fb/min(res[('fwd',0)])/1e6:4.2f} TB/s)  8-elt {min(res[('fwd',1)]):6.1f} us ({fb/min(res[('fwd',1)])/1e6:4.2f} TB/s)"
          f"   bwd {bb/1e6:6.0f} MB: 4-elt {min(res[('bwd',0)]):6.1f} us ({bb/min(res[('bwd',0)])/1e6:4.2f} TB/s)  8-elt {min(res[('bwd',1)]):6.1f} us ({bb/min(res[('bwd',1)])/1e6:4.2f} TB/s)", flush=True)
