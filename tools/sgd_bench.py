"""sgd_momentum_kernel on the ViT-B/16 parameter count (85.8 M): time and bytes/s."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_torch_amd import ops
n = 85_806_346 // 4 * 4
p, g, b = (torch.randn(n, device="cuda") for _ in range(3))
sh = torch.empty(n, device="cuda", dtype=torch.bfloat16)
big = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
def run():
    ops.sgd_momentum(p, g, b, sh, 1e-3, 0.9)
for _ in range(3): run()
ts = []
for _ in range(10):
    big.zero_()                                   # evict: the step's other 30 ms do the same
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
print(f"sgd_momentum {n/1e6:.1f} M params: median {ts[5]:.1f} us, min {ts[0]:.1f} us = {n*22/ts[5]/1e6:.2f} TB/s (22 B / param)")
