"""Per-GEMM-call durations of one ViT-B/16 step in the bf16x3 mode (eager, HIP events): finds calls that miss the tile kernels."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vit_torch_amd import CrossEntropyLoss  # noqa: E402

m = bench.build_model("dino_vitb16", 224, "bf16x3", "fp32").cuda()
m.train()
x = torch.randn(256, 3, 224, 224, device="cuda")
y = torch.randint(0, 10, (256,), device="cuda")
crit = CrossEntropyLoss()
eng = m.engine()
for it in range(2):
    eng.profile = [] if it == 1 else None
    loss = crit(m(x), y)
    loss.backward()
    torch.cuda.synchronize()
rows = {}
for name, shape, fl, e0, e1 in eng.profile:
    r = rows.setdefault((name, shape), [0, 0.0])
    r[0] += 1
    r[1] += e0.elapsed_time(e1)
for (name, shape), (n, ms) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:10s} {str(shape):24s} x{n:3d}  {ms / n * 1e3:9.1f} us each  {2.0 * shape[0] * shape[1] * shape[2] * n / ms / 1e9:8.1f} TFLOP/s (algorithmic)")
