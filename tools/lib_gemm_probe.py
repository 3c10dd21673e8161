"""How fast does the vendor BLAS (through torch.mm: hipBLASLt / rocBLAS) run the ViT-B/16
GEMM shapes, beside vitmi_gemm, on the same random operands?  Diagnostic only: the product
never calls torch.mm."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_torch_amd import ops  # noqa: E402

def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

M = 50432
bt = torch.bfloat16
for layout, m, n, k in [("nt", M, 2304, 768), ("nt", M, 768, 768), ("nt", M, 3072, 768), ("nt", M, 768, 3072),
                        ("nn", M, 768, 768), ("nn", M, 768, 2304), ("nn", M, 768, 3072), ("nn", M, 3072, 768),
                        ("tn", 768, 768, M), ("tn", 2304, 768, M), ("tn", 3072, 768, M), ("tn", 768, 3072, M),
                        ("nt", 4096, 4096, 4096), ("nt", 8192, 8192, 8192)]:
    akm, bkm = {"nt": (True, True), "nn": (True, False), "tn": (False, False)}[layout]
    A = torch.randn((m, k) if akm else (k, m), device="cuda").to(bt)
    B = (torch.randn((n, k) if bkm else (k, n), device="cuda") * 0.05).to(bt)
    cdt = torch.float32 if layout == "tn" else bt
    C = torch.empty((m, n), device="cuda", dtype=cdt)
    t_mine = timed(lambda: ops.gemm(A, B, C, a_kmajor=akm, b_kmajor=bkm))
    Am = A if akm else A.t()
    Bm = B.t() if bkm else B
    Cl = torch.empty((m, n), device="cuda", dtype=bt)
    t_lib = timed(lambda: torch.mm(Am, Bm, out=Cl))
    fl = 2.0 * m * n * k
    print(f"{layout} {m:6d} {n:5d} {k:6d}: vitmi {t_mine*1e3:8.1f} us {fl/t_mine/1e9:7.1f} TF | torch.mm {t_lib*1e3:8.1f} us {fl/t_lib/1e9:7.1f} TF", flush=True)
