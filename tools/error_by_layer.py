"""Where does the bf16 deviation come from?  Residual stream after every block of the HIP bf16 path against the fp32
oracle's, same seeded weights and inputs: RMS error relative to the RMS of the stream, and the same for the final
features / logits / loss.  usage: python tools/error_by_layer.py arch img batch [residual]   """
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vit_ref  # noqa: E402
from vit_torch_amd import CrossEntropyLoss, VisionModelZoo  # noqa: E402

arch, img, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
residual = sys.argv[4] if len(sys.argv) > 4 else "bf16"
ref = vit_ref.build(arch, classifier=10)
vit_ref.seeded_init_(ref, 1)
g = torch.Generator("cpu").manual_seed(0)
x, y = torch.randn(B, 3, img, img, generator=g), torch.randint(0, 10, (B,), generator=g)
stream = []
hooks = [blk.register_forward_hook(lambda m, i, o: stream.append(o.detach())) for blk in ref.blocks]
lo = ref(x)
lr = F.cross_entropy(lo, y)
for h in hooks:
    h.remove()
for mode, res in (("fp32", "fp32"), ("bf16", "fp32"), ("bf16", residual)):
    m = VisionModelZoo.get_model(arch, pretrained=False, classifier=10, compute_dtype=mode, residual_dtype=res)
    m.load_state_dict(ref.state_dict(), strict=True)
    m = m.cuda()
    out = m(x.cuda())
    loss = CrossEntropyLoss()(out, y.cuda())
    sv = m.engine().saved
    xs = [b[0] for b in sv["blocks"]][1:] + [sv["Xf"]]          # input of block i+1 = output of block i
    print(f"--- {arch}@{img} batch {B}: compute {mode}, residual stream {res}")
    for i, (a, r) in enumerate(zip(xs, stream)):
        a = a.float().cpu().view_as(r)
        e = (a - r)
        print(f"  after block {i:2d}: rms err / rms {e.pow(2).mean().sqrt().item() / r.pow(2).mean().sqrt().item():.3e}   "
              f"max|err| / max|x| {e.abs().max().item() / r.abs().max().item():.3e}   rms(x) {r.pow(2).mean().sqrt().item():.3f}")
    d = out.float().cpu() - lo
    print(f"  logits: max|err| {d.abs().max().item():.3e}  max|logit| {lo.abs().max().item():.3f}  rel {d.abs().max().item() / lo.abs().max().item():.3e}; "
          f"per-sample mean err {[round(v, 5) for v in d.mean(1).tolist()]}")
    lp_r, lp_m = F.log_softmax(lo, -1), F.log_softmax(out.float().cpu(), -1)
    per = (lp_m - lp_r)[torch.arange(B), y]
    print(f"  loss {loss.item():.5f} vs {lr.item():.5f}: diff {loss.item() - lr.item():+.3e}; per-sample -log p diff {[round(-v, 5) for v in per.tolist()]}")
    m.engine().saved = None
    del m
