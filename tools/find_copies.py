"""Which Python lines launch device copies (hipMemcpy* / blit kernels) inside one eager training step?
usage: python tools/find_copies.py arch [batch]"""
import os, sys
import torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_torch_amd import CrossEntropyLoss, FusedSGD, VisionModelZoo  # noqa: E402

arch = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
extra = {"num_classes": 10} if arch.startswith("swin") else {}
m = VisionModelZoo.get_model(arch, pretrained=False, classifier=10 if not arch.startswith("swin") else None, compute_dtype="bf16", **extra).cuda()
m.train()
x = torch.randn(B, 3, 224, 224, device="cuda")
y = torch.randint(0, 10, (B,), device="cuda")
crit = CrossEntropyLoss()
m.engine()
opt = FusedSGD(m.parameters(), lr=1e-3, momentum=0.9)


def step():
    opt.zero_grad()
    loss = crit(m(x), y)
    loss.backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
n = 0
for e in prof.events():
    nm = e.name
    if "emcpy" in nm or "copyBuffer" in nm or "emset" in nm or nm in ("aten::copy_", "aten::clone", "aten::zero_", "aten::fill_"):
        n += 1
        st = [s for s in (e.stack or []) if "vit_torch_amd" in s or "bench" in s or "tools/" in s][:2]
        print(f"{nm:40s} {st}")
print("total", n)
