"""Host-side op census of one eager training step (torch.profiler): which ATen ops and kernels a step launches, by count.
usage: python tools/host_ops_profile.py <arch> <batch>"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--arch", sys.argv[1], "--batch", sys.argv[2]]
import bench
from torch.profiler import profile, ProfilerActivity
a = bench.parse()
dev = torch.device("cuda", 0)
from vit_torch_amd import CrossEntropyLoss, FusedSGD
torch.manual_seed(1)
model = bench.build_model(a.arch, a.img, a.compute, a.residual).to(dev)
x = torch.randn(a.batch, 3, a.img, a.img, device=dev); y = torch.randint(0, 10, (a.batch,), device=dev)
crit = CrossEntropyLoss(); eng = model.engine()
opt = FusedSGD(model.parameters(), lr=1e-3, momentum=0.9)
def step():
    opt.zero_grad(); loss = crit(model(x), y); loss.backward(); opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
import collections
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::contiguous", "aten::clone", "aten::fill_", "aten::zero_", "aten::zeros", "aten::cat"):
        st = [s for s in ev.stack if "vit_torch_amd" in s or "bench" in s]
        cnt[(ev.name, st[0] if st else "?")] += 1
for k, v in cnt.most_common(25): print(v, k)
c2 = collections.Counter(ev.name for ev in prof.events())
for k, v in c2.most_common(60): print(v, k[:100])
