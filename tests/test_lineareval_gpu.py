"""Linear evaluation (/root/reference/main.py:184-201): frozen backbone under
no_grad, a stand-alone classifier head trained on its features — both on libvitmi kernels."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from tests.util import assert_close

pytestmark = pytest.mark.gpu


def _data(n, B=48, S=32):
    g = torch.Generator("cpu").manual_seed(11)
    return [(torch.randn(B, 3, S, S, generator=g), torch.randint(0, 10, (B,), generator=g)) for _ in range(n)]


def test_classifier_head_matches_torch_and_has_no_cpu_path(lib):
    from vit_torch_amd import VisionModelZoo, VitmiError
    from vit_torch_amd.head import ClassifierHead
    head = VisionModelZoo.get_classifier_head(64, [32, 24, 10])
    assert isinstance(head, ClassifierHead) and isinstance(head, nn.Sequential)
    ref = nn.Sequential(nn.Linear(64, 32), nn.GELU(), nn.Linear(32, 24), nn.GELU(), nn.Linear(24, 10, bias=False))
    assert list(head.state_dict().keys()) == list(ref.state_dict().keys())
    ref.load_state_dict(head.state_dict())
    with pytest.raises(VitmiError):
        head(torch.zeros(2, 64))
    head = head.cuda()
    g = torch.Generator("cpu").manual_seed(0)
    x = torch.randn(37, 64, generator=g)
    xr = x.clone().requires_grad_(True)
    xg = x.cuda().requires_grad_(True)
    out_r = ref(xr)
    out = head(xg)
    assert_close("head.out", out, out_r.detach(), 2e-5)
    dy = torch.randn(37, 10, generator=g)
    out_r.backward(dy)
    out.backward(dy.cuda())
    assert_close("head.dx", xg.grad, xr.grad, 5e-5)
    for (n, p), (_, q) in zip(head.named_parameters(), ref.named_parameters()):
        assert_close(f"head.grad[{n}]", p.grad, q.grad, 5e-5)


def test_lineareval_harness_matches_the_reference_loop(lib):
    from oracle import vit_ref
    from vit_torch_amd import VisionModelZoo, VisionTransformer
    from vit_torch_amd.network import Network
    cfg = dict(img_size=32, patch_size=8, embed_dim=64, depth=2, num_heads=2)
    ref_bb = vit_ref.VisionTransformer(**cfg)
    vit_ref.seeded_init_(ref_bb, 7)
    bb = VisionTransformer(**cfg, apply_head=False, compute_dtype="fp32")
    bb.load_state_dict(ref_bb.state_dict(), strict=False)
    out_dim = VisionModelZoo.get_output_shape(bb, [1, 3, 32, 32], "cuda")[-1]          # main.py:194
    assert out_dim == 64
    head = VisionModelZoo.get_classifier_head(out_dim, [32, 10])
    ref_head = nn.Sequential(nn.Linear(64, 32), nn.GELU(), nn.Linear(32, 10, bias=False))
    ref_head.load_state_dict(head.state_dict())
    data = _data(3)
    # reference loop: frozen bottom under no_grad, SGD momentum 0.9 on the head only
    opt = torch.optim.SGD(ref_head.parameters(), lr=5e-2, momentum=0.9)
    ref_losses = []
    for x, y in data:
        with torch.no_grad():
            feat = ref_bb(x)
        loss = F.cross_entropy(ref_head(feat), y)
        opt.zero_grad(); loss.backward(); opt.step()
        ref_losses.append(loss.item())
    before = {k: v.clone() for k, v in bb.state_dict().items()}
    net = Network(head, opt="sgd", lr=5e-2, lr_type="step", lr_step=100, frozen_model_bottom=[bb])
    rec = net.run_one_epoch(data, training=True)
    assert rec["loss"] == pytest.approx(ref_losses, rel=2e-4, abs=2e-5)
    for (n, p), (_, q) in zip(head.named_parameters(), ref_head.named_parameters()):
        assert_close(f"head[{n}] after 3 steps", p.detach(), q.detach(), 2e-4)
    for k, v in bb.state_dict().items():
        assert torch.equal(v.cpu(), before[k].cpu()), f"frozen backbone parameter {k} changed"
    assert bb.engine().saved is None, "the frozen forward must not keep activations"
