"""vitmi_pos_resample (posembed.hip) against torch's F.interpolate on the CPU and the engine's use of it
(VitEngine._pos_for / backward): the bicubic resize of pos_embed, forward and gradient, with no ATen math."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("side,gh,gw,D", [(14, 2, 2, 384), (28, 12, 12, 768), (14, 37, 37, 64), (14, 9, 5, 16)])
def test_pos_resample_kernel_matches_interpolate(side, gh, gw, D):
    from vit_torch_amd import ops
    from vit_torch_amd.posembed import tables_for
    g = torch.Generator().manual_seed(gh)
    pos = torch.randn(1, 1 + side * side, D, generator=g)
    leaf = pos.clone().requires_grad_(True)
    patch = leaf[:, 1:].reshape(1, side, side, D).permute(0, 3, 1, 2)
    ref = F.interpolate(patch, scale_factor=((gh + 0.1) / side, (gw + 0.1) / side), mode="bicubic")
    eff = torch.cat((leaf[:, :1], ref.permute(0, 2, 3, 1).reshape(1, -1, D)), 1)[0]
    gout = torch.randn(eff.shape, generator=g)
    (gl,) = torch.autograd.grad(eff, leaf, gout)
    tabs = tables_for(side * side, gh, gw, "cuda")
    out = ops.pos_resample(pos[0].cuda(), tabs.fwd).cpu()
    assert (out - eff.detach()).abs().max() <= 1e-5 * eff.detach().abs().max()      # VERDICT r02 item 5 bar
    gin = ops.pos_resample(gout.cuda(), tabs.bwd).cpu()
    assert (gin - gl[0]).abs().max() <= 1e-5 * gl.abs().max()


def test_engine_resize_runs_on_the_kernel_and_matches_oracle():
    """dino_vits16 at 32x32 (BASELINE config 0's shape, 14x14 -> 2x2): logits and d pos_embed against the
    oracle, with the resize and its backward on vitmi_pos_resample (no autograd graph inside the engine)."""
    from oracle import vit_ref
    from vit_torch_amd import VisionModelZoo
    ref = vit_ref.build("dino_vits16", classifier=10)
    vit_ref.seeded_init_(ref, 1)
    m = VisionModelZoo.get_model("dino_vits16", pretrained=False, classifier=10, compute_dtype="fp32")
    m.load_state_dict(ref.state_dict(), strict=True)
    m = m.cuda()
    g = torch.Generator("cpu").manual_seed(0)
    x, y = torch.randn(4, 3, 32, 32, generator=g), torch.randint(0, 10, (4,), generator=g)
    out = m(x.cuda())
    saved = m.engine().saved
    assert saved["pos_tabs"] is not None and "pos_graph" not in saved
    F.cross_entropy(out, y.cuda()).backward()
    out_r = ref(x)
    F.cross_entropy(out_r, y).backward()
    assert (out.detach().cpu() - out_r.detach()).abs().max() <= 1e-4 * out_r.detach().abs().max()
    gp, gr = m.pos_embed.grad.cpu(), ref.pos_embed.grad
    assert (gp - gr).abs().max() <= 1e-4 * gr.abs().max()
