"""Swin-T bs 256 step (bf16, HIP graph) with the configuration's DropPath rates against rate 0: what the per-sample row scale
of the residual epilogue (and the per-step draw) costs."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vit_torch_amd import CrossEntropyLoss, FusedSGD  # noqa: E402

x = torch.randn(256, 3, 224, 224, device="cuda")
y = torch.randint(0, 10, (256,), device="cuda")
for rnd in range(2):
    for rate in (None, 0.0):
        extra = {} if rate is None else {"drop_path_rate": rate}
        torch.manual_seed(1)
        m = bench.build_model("swin_tiny_patch4_window7_224", 224, "bf16", "bf16", **extra).cuda()
        m.train()
        m.engine()
        opt = FusedSGD(m.parameters(), lr=1e-3, momentum=0.9)
        step, graphed, err = bench.runner(m, CrossEntropyLoss(), opt, x, y, "on")
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(15):
            step()
        torch.cuda.synchronize()
        print(f"round {rnd} drop_path {'config' if rate is None else rate}: {(time.perf_counter() - t) / 15 * 1e3:.3f} ms/step (graph {graphed})", flush=True)
        del m, opt, step
        torch.cuda.empty_cache()
