"""Fused optimizers of the reference's table (/root/reference/utils_network.py:119-126) against
torch.optim on the CPU (adagrad, adadelta, adam, adamw, sgd) and against the oracle's restatement of
AdaBelief (oracle/optim_ref.py; the adabelief_pytorch package is absent: parity unpinned), on
IDENTICAL injected gradients, so that only the update rule is compared.  12 steps: AdaBelief's
rectification switches from the SGD form to the adaptive form at step 6 (rho_t >= 5)."""
import pytest
import torch

from util import assert_close

pytestmark = pytest.mark.gpu

SMALL = dict(img_size=32, patch_size=16, in_chans=3, embed_dim=64, depth=1, num_heads=2)


def _model(compute="fp32"):
    from vit_torch_amd import VisionTransformer
    torch.manual_seed(3)
    m = VisionTransformer(**SMALL, num_classes=10, apply_head=True, compute_dtype=compute, residual_dtype="auto").cuda()
    m.engine()
    return m


def _cpu_twin(m):
    return [p.detach().cpu().clone().requires_grad_(True) for p in m.parameters()]


def _inject(m, twin, step, scale=1.0):
    pack = m.engine().pack
    g = torch.Generator().manual_seed(1000 + step)
    for p, q in zip(m.parameters(), twin):
        gr = torch.randn(q.shape, generator=g) * scale
        q.grad = gr.clone()
        if p.requires_grad:
            pack.g(p).copy_(gr.cuda())


CASES = {
    "adagrad": (lambda ps: __import__("vit_torch_amd").FusedAdagrad(ps, lr=1e-2),
                lambda ps: torch.optim.Adagrad(ps, lr=1e-2)),
    "adagrad_decay": (lambda ps: __import__("vit_torch_amd").FusedAdagrad(ps, lr=1e-2, lr_decay=0.1, weight_decay=0.01, initial_accumulator_value=0.5),
                      lambda ps: torch.optim.Adagrad(ps, lr=1e-2, lr_decay=0.1, weight_decay=0.01, initial_accumulator_value=0.5)),
    "adadelta": (lambda ps: __import__("vit_torch_amd").FusedAdadelta(ps, lr=1.0),
                 lambda ps: torch.optim.Adadelta(ps, lr=1.0)),
    "adam": (lambda ps: __import__("vit_torch_amd").FusedAdamW(ps, lr=1e-3, weight_decay=0.0, decoupled=False),
             lambda ps: torch.optim.Adam(ps, lr=1e-3)),
    "adamw": (lambda ps: __import__("vit_torch_amd").FusedAdamW(ps, lr=1e-3),
              lambda ps: torch.optim.AdamW(ps, lr=1e-3)),
    "sgd": (lambda ps: __import__("vit_torch_amd").FusedSGD(ps, lr=1e-2, momentum=0.9),
            lambda ps: torch.optim.SGD(ps, lr=1e-2, momentum=0.9)),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_fused_optimizer_matches_torch(name):
    m = _model()
    twin = _cpu_twin(m)
    fused, ref = CASES[name][0](m.parameters()), CASES[name][1](twin)
    for s in range(12):
        _inject(m, twin, s)
        fused.step(); ref.step()
    for (n, p), q in zip(m.named_parameters(), twin):
        assert_close(f"{name} param[{n}]", p.data, q.data, 2e-5)


@pytest.mark.parametrize("rectify,decouple,wd", [(True, True, 0.0), (True, True, 0.05), (False, False, 0.05)])
def test_fused_adabelief_matches_the_restated_algorithm(rectify, decouple, wd):
    from oracle.optim_ref import AdaBeliefRef
    from vit_torch_amd import FusedAdaBelief
    m = _model()
    twin = _cpu_twin(m)
    fused = FusedAdaBelief(m.parameters(), lr=1e-3, weight_decay=wd, weight_decouple=decouple, rectify=rectify)
    ref = AdaBeliefRef(twin, lr=1e-3, weight_decay=wd, weight_decouple=decouple, rectify=rectify)
    for s in range(12):
        _inject(m, twin, s)
        fused.step(); ref.step()
        if s in (3, 11):                       # both sides of the rho_t >= 5 switch (t = 6)
            for (n, p), q in zip(m.named_parameters(), twin):
                assert_close(f"adabelief step {s + 1} param[{n}]", p.data, q.data, 5e-5)


def test_network_table_has_every_reference_optimizer():
    """utils_network.py:119-126: sgd, adam, adadelta, adagrad, adamw, adabelief."""
    from vit_torch_amd.network import Network
    for k in ("sgd", "adam", "adadelta", "adagrad", "adamw", "adabelief"):
        assert k in Network.optimizer_fns
    net = Network(_model(), opt="adabelief", lr=1e-3)
    assert type(net.optimizer).__name__ == "FusedAdaBelief"


def test_step_counts_follow_each_parameter_when_the_trainable_set_changes():
    """a span that appears later (unfrozen parameter) starts its bias corrections at step 1
    with zero moments, the others continue theirs — as torch.optim.AdamW, whose `step` is per parameter."""
    from vit_torch_amd import FusedAdamW
    m = _model()
    twin = _cpu_twin(m)
    names = [n for n, _ in m.named_parameters()]
    frozen = [i for i, n in enumerate(names) if n.startswith("blocks.0.mlp.")]
    for i, p in enumerate(m.parameters()):
        if i in frozen:
            p.requires_grad_(False)
    for i in frozen:
        twin[i].requires_grad_(False)
    fused, ref = FusedAdamW(m.parameters(), lr=1e-2), torch.optim.AdamW(twin, lr=1e-2)

    def run(s):
        _inject(m, twin, s)
        for i in frozen:
            if not twin[i].requires_grad:
                twin[i].grad = None
        fused.step(); ref.step()

    for s in range(3):
        run(s)
    for i, p in enumerate(m.parameters()):
        if i in frozen:
            p.requires_grad_(True)
            twin[i].requires_grad_(True)
    for s in range(3, 6):
        run(s)
    for (n, p), q in zip(m.named_parameters(), twin):
        assert_close(f"param[{n}]", p.data, q.data, 2e-5)
