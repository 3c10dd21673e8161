"""Model-level parity on the BASELINE.json configurations that round 1 only touched at op level:
C1 dino_vits16 at 32x32, batch 128 (N = 5 tokens, D = 384, 6 heads; pos_embed bicubic-resized from
the stored 14x14 grid to 2x2) and C3 dino_vitb8 at 96x96 (N = 145, patch 8; 28x28 -> 12x12), built
through the factory (`/root/reference/models/vision_all.py:37-43,154-182`), in the fp32 parity
mode and in the bf16 perf mode, against the CPU oracle on identical seeded weights and inputs.

Also: the bf16 weight shadow must follow every way torch can change a parameter, and
the fused optimizers must touch exactly the parameters they were given.

Tolerances: fp32 logits 1e-3 rel (north-star bar; measured 1.4e-6 / 2.6e-6).  bf16 bounds are
~1.6-2x what was measured on the MI355X (printed by the tests; round 2: logits 7.4e-3 (vits16@32),
9.3e-3 (vitb8@96, max|logit| only 1.1), loss 1.1e-4 / 4.7e-3, grad-norm 1.8e-3): logits 1.5e-2,
loss 1e-2, grad-norm 8e-3.  bf16 operands cannot meet the 1e-3 bar; the fp32 mode does."""
import pytest
import torch
import torch.nn.functional as F

from util import assert_close, grad_agreement

pytestmark = pytest.mark.gpu

BF16_LOGITS, BF16_LOSS, BF16_GRADNORM = 1.5e-2, 1e-2, 8e-3
BF16_COS = 0.9995           # worst per-parameter gradient cosine, bf16 mode vs the fp32 oracle (printed by the test)


def data(B, S, seed=0):
    g = torch.Generator("cpu").manual_seed(seed)
    return torch.randn(B, 3, S, S, generator=g), torch.randint(0, 10, (B,), generator=g)


def pair(arch, img, compute, residual="fp32", seed=1):
    from oracle import vit_ref
    from vit_torch_amd import VisionModelZoo
    ref = vit_ref.build(arch, classifier=10)          # stored 224-grid pos_embed, resized per input
    vit_ref.seeded_init_(ref, seed)
    m = VisionModelZoo.get_model(arch, pretrained=False, classifier=10, compute_dtype=compute,
                                 residual_dtype=residual)
    res = m.load_state_dict(ref.state_dict(), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return ref, m.cuda()


def step(ref, m, x, y):
    from vit_torch_amd import CrossEntropyLoss
    lo = ref(x)
    lr = F.cross_entropy(lo, y)
    ref.zero_grad(); lr.backward()
    crit = CrossEntropyLoss()
    out = m(x.cuda())
    loss = crit(out, y.cuda())
    m.zero_grad(); loss.backward()
    return lo.detach(), lr.detach(), out.detach(), loss.detach(), crit


def gradnorm_worst(ref, m):
    worst, name = 0.0, ""
    for (n, pr), (n2, pm) in zip(ref.named_parameters(), m.named_parameters()):
        assert n == n2
        gr, gm = pr.grad.double().norm().item(), pm.grad.double().norm().item()
        rel = abs(gm - gr) / max(gr, 1e-12)
        if rel > worst:
            worst, name = rel, n
    return worst, name


CONFIGS = [("dino_vits16", 32, 128), ("dino_vitb8", 96, 4)]


@pytest.mark.parametrize("arch,img,B", CONFIGS)
def test_config_fp32_parity(arch, img, B):
    ref, m = pair(arch, img, "fp32")
    x, y = data(B, img)
    lo, lr, out, loss, crit = step(ref, m, x, y)
    e = assert_close(f"{arch} logits", out, lo, 1e-3)
    assert abs(loss.item() - lr.item()) < 1e-3
    worst, name = gradnorm_worst(ref, m)
    assert worst < 1e-3, (name, worst)
    for (n, pr), (_, pm) in zip(ref.named_parameters(), m.named_parameters()):      # every entry, and the direction
        if pr.grad.abs().max().item() > 1e-9:
            assert_close(f"grad[{n}]", pm.grad, pr.grad, 1e-3)
    assert grad_agreement(ref, m)[2] > 1 - 1e-6
    assert crit.last_correct.item() == (lo.argmax(-1) == y).sum().item()
    print(f"\n{arch}@{img} bs{B} fp32: logits rel {e:.2e}, loss diff {abs(loss.item() - lr.item()):.2e}, "
          f"worst grad-norm rel {worst:.2e} ({name})")


@pytest.mark.parametrize("arch,img,B", CONFIGS)
def test_config_bf16_deviation_is_bounded(arch, img, B):
    ref, m = pair(arch, img, "bf16")
    x, y = data(B, img)
    lo, lr, out, loss, _ = step(ref, m, x, y)
    e = assert_close(f"{arch} bf16 logits", out, lo, BF16_LOGITS)
    assert abs(loss.item() - lr.item()) < BF16_LOSS
    worst, name = gradnorm_worst(ref, m)
    assert worst < BF16_GRADNORM, (name, worst)
    _, _, cmin, cname = grad_agreement(ref, m)
    assert cmin > BF16_COS, (cname, cmin)
    print(f"\n{arch}@{img} bs{B} bf16: logits rel {e:.2e}, loss diff {abs(loss.item() - lr.item()):.2e}, "
          f"worst grad-norm rel {worst:.2e} ({name}), worst gradient cosine {cmin:.6f} ({cname})")


# ------------------------------------------------------------- bf16 shadow freshness ---
SMALL = dict(img_size=48, patch_size=16, in_chans=3, embed_dim=128, depth=2, num_heads=2)


def small_pair(seed, compute="bf16"):
    from oracle import vit_ref
    from vit_torch_amd import VisionModelZoo, VisionTransformer
    ref = vit_ref.VisionTransformer(**SMALL, apply_head=True)
    ref.head = vit_ref.get_classifier_head(SMALL["embed_dim"], 10)
    vit_ref.seeded_init_(ref, seed)
    m = VisionTransformer(**SMALL, apply_head=True, compute_dtype=compute, residual_dtype="auto")
    m.head = VisionModelZoo.get_classifier_head(SMALL["embed_dim"], 10)
    m.load_state_dict(ref.state_dict(), strict=True)
    return ref, m.cuda()


def test_load_state_dict_on_a_live_bf16_engine_reaches_the_gemm_weights():
    """forward, load other weights into the SAME module, forward: must equal a fresh model."""
    ref_a, m = small_pair(1)
    ref_b, fresh = small_pair(2)
    x, _ = data(4, 48)
    with torch.no_grad():
        out_a = m(x.cuda()).clone()
        m.load_state_dict(ref_b.state_dict(), strict=True)
        out_b = m(x.cuda())
        want = fresh(x.cuda())
    assert not torch.allclose(out_a, out_b)
    assert torch.equal(out_b, want), "GEMMs still read the old bf16 weights after load_state_dict"


def test_in_place_parameter_writes_reach_the_gemm_weights():
    """p.add_(), nn.init and reset_parameters() after the first forward."""
    _, m = small_pair(1)
    _, twin = small_pair(1)
    x, _ = data(4, 48)
    with torch.no_grad():
        m(x.cuda())
        for mod in (m, twin):
            torch.manual_seed(7)
            mod.blocks[0].mlp.fc1.weight.add_(0.05 * torch.randn_like(mod.blocks[0].mlp.fc1.weight))
            torch.nn.init.trunc_normal_(mod.blocks[1].attn.qkv.weight, std=0.05,
                                        generator=torch.Generator("cuda").manual_seed(3))
        assert torch.equal(m(x.cuda()), twin(x.cuda()))


@pytest.mark.parametrize("opt_name", ["sgd", "adamw"])
def test_stock_torch_optimizer_trains_the_bf16_model(opt_name):
    """Two steps of torch.optim.* on the module: its updates must reach the bf16 GEMM operands
    (they did not in round 1).  (a) the trained module's forward equals, bit for bit, the forward
    of a FRESH module loaded with its state dict; (b) for SGD the trajectory equals FusedSGD's
    to fp32 rounding (Adam divides by |g|, so near-zero gradients make element-wise trajectories
    of two bf16 runs incomparable: (a) is the property that matters there)."""
    from vit_torch_amd import CrossEntropyLoss, FusedSGD
    _, a = small_pair(1)
    _, b = small_pair(1)
    a.engine(); b.engine()
    oa = (torch.optim.SGD(a.parameters(), lr=0.05, momentum=0.9) if opt_name == "sgd"
          else torch.optim.AdamW(a.parameters(), lr=1e-2))
    ob = FusedSGD(b.parameters(), lr=0.05, momentum=0.9)
    crit = CrossEntropyLoss()
    x0, _ = data(4, 48, seed=70)
    with torch.no_grad():
        before = a(x0.cuda()).clone()
    for s in range(2):
        x, y = data(4, 48, seed=60 + s)
        for mod, opt in ((a, oa), (b, ob)):
            opt.zero_grad()
            crit(mod(x.cuda()), y.cuda()).backward()
            opt.step()
    _, fresh = small_pair(5)
    fresh.load_state_dict(a.state_dict(), strict=True)
    with torch.no_grad():
        out = a(x0.cuda())
        assert not torch.allclose(out, before), "the optimizer did not change the function at all"
        assert torch.equal(out, fresh(x0.cuda())), "GEMMs read stale bf16 weights after torch.optim steps"
    if opt_name == "sgd":
        for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
            assert_close(f"param[{n}]", pa.data, pb.data, 2e-5)


# ------------------------------------------------------ fused optimizers: param groups ---
@pytest.mark.parametrize("kind", ["sgd", "adamw"])
def test_fused_optimizer_updates_only_its_param_groups(kind):
    """An optimizer over the head alone must leave the backbone bit-identical (AdamW's decoupled
    decay included); two groups over one model apply each group's lr to its own span only."""
    from vit_torch_amd import CrossEntropyLoss, FusedAdamW, FusedSGD
    _, m = small_pair(1, "fp32")
    _, twin = small_pair(1, "fp32")
    m.engine(); twin.engine()
    mk = (lambda ps, **k: FusedSGD(ps, momentum=0.9, **k)) if kind == "sgd" else (lambda ps, **k: FusedAdamW(ps, **k))
    ref_mk = ((lambda ps, **k: torch.optim.SGD(ps, momentum=0.9, **k)) if kind == "sgd"
              else (lambda ps, **k: torch.optim.AdamW(ps, **k)))
    head = list(m.head.parameters())
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    opt = mk(head, lr=0.05)
    crit = CrossEntropyLoss()
    x, y = data(4, 48)
    opt.zero_grad(); crit(m(x.cuda()), y.cuda()).backward(); opt.step()
    for n, p in m.named_parameters():
        if n.startswith("head."):
            assert not torch.equal(p, before[n]), n
        else:
            assert torch.equal(p, before[n]), f"{n} changed although it is not in the optimizer"
    # two groups, different learning rates, vs torch on the twin
    def groups(mod, lr_body, lr_head):
        hp = list(mod.head.parameters())
        ids = {id(p) for p in hp}
        return [{"params": [p for p in mod.parameters() if id(p) not in ids], "lr": lr_body},
                {"params": hp, "lr": lr_head}]
    _, m2 = small_pair(1, "fp32")
    m2.engine()
    o1, o2 = mk(groups(m2, 1e-3, 5e-2), lr=1e-3), ref_mk(groups(twin, 1e-3, 5e-2), lr=1e-3)
    for s in range(2):
        x, y = data(4, 48, seed=80 + s)
        for mod, o in ((m2, o1), (twin, o2)):
            o.zero_grad(); crit(mod(x.cuda()), y.cuda()).backward(); o.step()
    for (n, pa), (_, pb) in zip(m2.named_parameters(), twin.named_parameters()):
        # the key third of qkv.bias has an analytically ZERO gradient (softmax shift invariance):
        # Adam divides rounding noise by its own magnitude there, so only its bound is checked
        noise = kind == "adamw" and n.endswith("attn.qkv.bias")
        assert_close(f"param[{n}]", pa.data, pb.data, 5e-2 if noise else 2e-5)


def test_frozen_parameters_are_not_updated_by_fused_adamw():
    from vit_torch_amd import CrossEntropyLoss, FusedAdamW
    _, m = small_pair(1, "fp32")
    for p in m.blocks[0].parameters():
        p.requires_grad_(False)
    m.engine()
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    opt = FusedAdamW(m.parameters(), lr=1e-2, weight_decay=0.1)
    x, y = data(4, 48)
    opt.zero_grad(); CrossEntropyLoss()(m(x.cuda()), y.cuda()).backward(); opt.step()
    for n, p in m.named_parameters():
        if n.startswith("blocks.0."):
            assert torch.equal(p, before[n]), f"frozen {n} was decayed / updated"
        else:
            assert not torch.equal(p, before[n]), n


def test_out_of_range_label_gives_nan_loss_not_a_fault():
    from vit_torch_amd import CrossEntropyLoss
    crit = CrossEntropyLoss()
    logits = torch.randn(4, 10, device="cuda")
    loss = crit(logits, torch.tensor([1, 2, 10, 3], device="cuda"))
    assert torch.isnan(loss).item()
    with pytest.raises(TypeError):
        crit(logits, torch.tensor([1, 2, 3, 4], device="cuda", dtype=torch.int32))
