"""End-to-end parity of the HIP ViT (vit_torch_amd.VisionTransformer) against the
CPU oracle (oracle/vit_ref.py) on identical seeded weights and inputs.

Tolerances (max|diff| / max|ref|):
  fp32 mode (parity mode): logits 1e-3 is the north-star bar; we assert 1e-4.
  bf16 mode (perf mode: bf16 GEMM/attention operands, fp32 accumulation): bounds ~2x the
  round-2 measurements on the MI355X (small model 4.2e-3 / 1.5e-4 / 1.8e-3 with the fp32
  residual stream, 4.9e-3 / 4.2e-4 / 3.3e-3 with the bf16 one; ViT-B/16 @224 5.1e-3 / 4.0e-4 /
  1.8e-3 and 8.8e-3 / 4.4e-3 / 2.6e-3): logits 1.2e-2 (fp32 residual) / 2e-2 (bf16 residual),
  loss 2e-3 / 1e-2, per-parameter grad-norm 8e-3.  The tests print the achieved numbers.
"""
import pytest
import torch
import torch.nn.functional as F

from util import assert_close, rel_err, grad_agreement

pytestmark = pytest.mark.gpu

# (logits rel-to-max, |loss diff|, per-parameter grad-norm rel) by residual-stream dtype
BF16_TOL = {"fp32": (1.2e-2, 1e-2, 8e-3), "bf16": (2e-2, 1e-2, 8e-3)}
# worst per-parameter cosine between bf16-mode gradients and the fp32 oracle's (small 2-block model / full ViT-B/16);
# the tests print the measured value
BF16_COS, BF16_COS_FULL = 0.9998, 0.9995        # measured 0.99996 / 0.99991 (round 3)


def make_pair(cfg, classifier, compute, residual="fp32", seed=1):
    from oracle import vit_ref
    from vit_torch_amd import VisionTransformer
    ref = vit_ref.VisionTransformer(**cfg, apply_head=classifier is not None)
    if classifier is not None:
        ref.head = vit_ref.get_classifier_head(cfg["embed_dim"], classifier)
    vit_ref.seeded_init_(ref, seed)
    m = VisionTransformer(**cfg, apply_head=classifier is not None, compute_dtype=compute,
                          residual_dtype=residual)
    if classifier is not None:
        from vit_torch_amd import VisionModelZoo
        m.head = VisionModelZoo.get_classifier_head(cfg["embed_dim"], classifier)
    missing = m.load_state_dict(ref.state_dict(), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return ref, m.cuda()


def data(B, C, S, K, seed=0):
    g = torch.Generator("cpu").manual_seed(seed)
    return torch.randn(B, C, S, S, generator=g), torch.randint(0, K, (B,), generator=g)


TINY = dict(img_size=32, patch_size=8, in_chans=3, embed_dim=64, depth=2, num_heads=2)          # N=17, hd=32
SMALL = dict(img_size=48, patch_size=16, in_chans=3, embed_dim=128, depth=3, num_heads=2)       # N=10, hd=64


def run_step(ref, m, x, y, K):
    from vit_torch_amd import CrossEntropyLoss
    out_ref = ref(x)
    loss_ref = F.cross_entropy(out_ref[:, :K] if out_ref.shape[1] != K else out_ref, y)
    ref.zero_grad()
    loss_ref.backward()
    crit = CrossEntropyLoss()
    out = m(x.cuda())
    loss = crit(out, y.cuda())
    m.zero_grad()
    loss.backward()
    return out_ref.detach(), loss_ref.detach(), out.detach(), loss.detach(), crit


@pytest.mark.parametrize("cfg,classifier", [(TINY, 10), (SMALL, [24, 10]), (TINY, None)])
def test_fp32_mode_matches_oracle(cfg, classifier):
    K = 10 if classifier is not None else cfg["embed_dim"]
    ref, m = make_pair(cfg, classifier, "fp32")
    x, y = data(6, 3, cfg["img_size"], K)
    out_ref, loss_ref, out, loss, crit = run_step(ref, m, x, y, K)
    e = assert_close("logits", out, out_ref, 1e-4)
    assert abs(loss.item() - loss_ref.item()) < 1e-4
    worst = 0.0
    for (n, pr), (n2, pm) in zip(ref.named_parameters(), m.named_parameters()):
        assert n == n2
        if pr.grad is None:
            assert pm.grad is None or pm.grad.abs().max().item() == 0
            continue
        assert pm.grad is not None, f"no grad for {n}"
        worst = max(worst, assert_close(f"grad[{n}]", pm.grad, pr.grad, 2e-4))
    assert crit.last_correct.item() == (out_ref[:, :K].argmax(-1) == y).sum().item()
    print(f"\nfp32 mode: logits rel err {e:.2e}, loss diff {abs(loss.item()-loss_ref.item()):.2e}, worst grad rel err {worst:.2e}")


@pytest.mark.parametrize("residual", ["fp32", "bf16"])
def test_bf16_mode_close_to_oracle(residual):
    cfg, classifier, K = SMALL, 10, 10
    ref, m = make_pair(cfg, classifier, "bf16", residual)
    x, y = data(8, 3, cfg["img_size"], K)
    out_ref, loss_ref, out, loss, _ = run_step(ref, m, x, y, K)
    e = assert_close("logits", out, out_ref, BF16_TOL[residual][0])
    assert abs(loss.item() - loss_ref.item()) < BF16_TOL[residual][1]
    worst = 0.0
    for (n, pr), (_, pm) in zip(ref.named_parameters(), m.named_parameters()):
        gn_ref, gn = pr.grad.norm().item(), pm.grad.float().norm().item()
        rel = abs(gn - gn_ref) / max(gn_ref, 1e-12)
        worst = max(worst, rel)
        assert rel < BF16_TOL[residual][2], f"grad-norm[{n}]: {gn:.4g} vs {gn_ref:.4g}"
    _, _, cmin, cname = grad_agreement(ref, m)
    assert cmin > BF16_COS, (cname, cmin)            # direction of every parameter's gradient, not only its length
    print(f"\nbf16 mode (residual {residual}): logits rel err {e:.2e}, loss diff "
          f"{abs(loss.item()-loss_ref.item()):.2e}, worst grad-norm rel err {worst:.2e}, worst cosine {cmin:.6f} ({cname})")


def test_two_sgd_steps_match_torch_sgd_fp32():
    """Harness parity (utils_network.py:440-442): zero_grad -> backward -> SGD(momentum 0.9)."""
    from vit_torch_amd import CrossEntropyLoss, FusedSGD
    cfg, K = TINY, 10
    ref, m = make_pair(cfg, 10, "fp32")
    opt_ref = torch.optim.SGD(ref.parameters(), lr=0.05, momentum=0.9)
    crit = CrossEntropyLoss()
    opt = None
    for step in range(2):
        x, y = data(4, 3, cfg["img_size"], K, seed=10 + step)
        opt_ref.zero_grad()
        F.cross_entropy(ref(x), y).backward()
        opt_ref.step()
        if opt is None:
            m(x.cuda())                      # builds the engine / flat buffers
            opt = FusedSGD(m.parameters(), lr=0.05, momentum=0.9)
        opt.zero_grad()
        crit(m(x.cuda()), y.cuda()).backward()
        opt.step()
    for (n, pr), (_, pm) in zip(ref.named_parameters(), m.named_parameters()):
        assert_close(f"param[{n}] after 2 steps", pm.data, pr.data, 1e-4)


def test_stock_torch_optimizer_also_works():
    """The module is a drop-in: torch.optim.SGD on its parameters must train it too."""
    cfg, K = TINY, 10
    ref, m = make_pair(cfg, 10, "fp32")
    opt_ref = torch.optim.SGD(ref.parameters(), lr=0.05, momentum=0.9)
    opt = torch.optim.SGD(m.parameters(), lr=0.05, momentum=0.9)
    for step in range(2):
        x, y = data(4, 3, cfg["img_size"], K, seed=20 + step)
        opt_ref.zero_grad(); F.cross_entropy(ref(x), y).backward(); opt_ref.step()
        opt.zero_grad(); F.cross_entropy(m(x.cuda()), y.cuda()).backward(); opt.step()
    for (n, pr), (_, pm) in zip(ref.named_parameters(), m.named_parameters()):
        assert_close(f"param[{n}]", pm.data, pr.data, 1e-4)


def test_no_grad_forward_and_eval_path():
    ref, m = make_pair(TINY, 10, "fp32")
    x, _ = data(3, 3, 32, 10)
    with torch.no_grad():
        assert_close("logits(no_grad)", m(x.cuda()), ref(x), 1e-4)


def test_pos_embed_interpolation_matches_oracle():
    """C1/C3 style: pos_embed stored for 32x32 (4x4 grid), input 16x16 (2x2 grid)."""
    cfg = dict(TINY)
    ref, m = make_pair(cfg, 10, "fp32")
    x, y = data(4, 3, 16, 10)
    out_ref, loss_ref, out, loss, _ = run_step(ref, m, x, y, 10)
    assert_close("logits(interp)", out, out_ref, 1e-4)
    assert_close("grad pos_embed", m.pos_embed.grad, ref.pos_embed.grad, 2e-4)


def test_vitb16_full_size_fp32_logits_within_1e3():
    """The north-star parity bar on the real architecture: dino_vitb16 @224, batch 2."""
    from oracle import vit_ref
    from vit_torch_amd import VisionModelZoo
    ref = vit_ref.build("dino_vitb16", classifier=10)
    vit_ref.seeded_init_(ref, 1)
    m = VisionModelZoo.get_model("dino_vitb16", pretrained=False, classifier=10,
                                 compute_dtype="fp32").cuda()
    m.load_state_dict(ref.state_dict(), strict=True)
    x, y = data(2, 3, 224, 10)
    out_ref, loss_ref, out, loss, _ = run_step(ref, m, x, y, 10)
    e = assert_close("vitb16 logits", out, out_ref, 1e-3)
    assert abs(loss.item() - loss_ref.item()) < 1e-3
    worst = 0.0
    for (n, pr), (_, pm) in zip(ref.named_parameters(), m.named_parameters()):
        gn_ref, gn = pr.grad.norm().item(), pm.grad.norm().item()
        worst = max(worst, abs(gn - gn_ref) / max(gn_ref, 1e-12))
    assert worst < 1e-3
    # every gradient ENTRY, rel-to-max per parameter, and the direction
    werr = 0.0
    for (n, pr), (_, pm) in zip(ref.named_parameters(), m.named_parameters()):
        if pr.grad.abs().max().item() > 1e-9:
            werr = max(werr, assert_close(f"grad[{n}]", pm.grad, pr.grad, 1e-3))
    _, _, cmin, cname = grad_agreement(ref, m)
    assert cmin > 1 - 1e-6, (cname, cmin)
    print(f"\nvitb16 fp32: logits rel err {e:.2e}, loss diff {abs(loss.item()-loss_ref.item()):.2e}, worst grad-norm rel {worst:.2e}, "
          f"worst grad entry rel-to-max {werr:.2e}, worst cosine {cmin:.8f} ({cname})")


@pytest.mark.parametrize("residual", ["fp32", "bf16"])
def test_vitb16_full_size_bf16_deviation_is_bounded(residual):
    """Perf mode on the real architecture (dino_vitb16 @224, batch 4): the deviation from the
    fp32 CPU oracle is MEASURED and bounded at ~2x the measurement (BF16_TOL above);
    bf16 operands cannot reach the 1e-3 parity bar, which the fp32 mode test above meets."""
    from oracle import vit_ref
    from vit_torch_amd import VisionModelZoo
    ref = vit_ref.build("dino_vitb16", classifier=10)
    vit_ref.seeded_init_(ref, 1)
    m = VisionModelZoo.get_model("dino_vitb16", pretrained=False, classifier=10,
                                 compute_dtype="bf16", residual_dtype=residual).cuda()
    m.load_state_dict(ref.state_dict(), strict=True)
    x, y = data(4, 3, 224, 10)
    out_ref, loss_ref, out, loss, _ = run_step(ref, m, x, y, 10)
    e = assert_close("vitb16 bf16 logits", out, out_ref, BF16_TOL[residual][0])
    assert abs(loss.item() - loss_ref.item()) < BF16_TOL[residual][1]
    worst, worst_name = 0.0, ""
    for (n, pr), (_, pm) in zip(ref.named_parameters(), m.named_parameters()):
        gn_ref, gn = pr.grad.norm().item(), pm.grad.norm().item()
        rel = abs(gn - gn_ref) / max(gn_ref, 1e-12)
        if rel > worst:
            worst, worst_name = rel, n
    assert worst < BF16_TOL[residual][2], f"{worst_name}: {worst}"
    _, _, cmin, cname = grad_agreement(ref, m)
    assert cmin > BF16_COS_FULL, (cname, cmin)
    print(f"vitb16 bf16 (residual {residual}): worst per-parameter gradient cosine {cmin:.6f} ({cname})")
    print(f"\nvitb16 bf16 (residual {residual}): logits rel err {e:.2e}, loss diff "
          f"{abs(loss.item()-loss_ref.item()):.2e}, worst grad-norm rel {worst:.2e} ({worst_name})")


def test_network_fit_matches_reference_loop_fp32():
    """Harness parity (utils_network.py:406-453 + LambdaLR per epoch :311-313): two epochs of
    three batches with the step LR schedule against the same loop on the CPU oracle."""
    from vit_torch_amd.network import LRSchedule, Network
    cfg, K = TINY, 10
    ref, m = make_pair(cfg, 10, "fp32")
    batches = [data(4, 3, cfg["img_size"], K, seed=30 + i) for i in range(3)]
    opt_ref = torch.optim.SGD(ref.parameters(), lr=0.05, momentum=0.9)
    sch_ref = torch.optim.lr_scheduler.LambdaLR(opt_ref, LRSchedule.get_step_fn(step=1, gamma=0.5))
    ref_losses, ref_correct = [], []
    for _ in range(2):
        for x, y in batches:
            out = ref(x)
            loss = F.cross_entropy(out, y)
            opt_ref.zero_grad(); loss.backward(); opt_ref.step()
            ref_losses.append(loss.item()); ref_correct.extend((out.argmax(-1) == y).tolist())
        sch_ref.step()
    net = Network(m, opt="sgd", lr=0.05, lr_type="step", lr_step=1, lr_gamma=0.5, device="cuda")
    hist = net.fit(batches, epochs=2)
    got_losses = hist[0]["train"]["loss"] + hist[1]["train"]["loss"]
    got_correct = list(hist[0]["train"]["correct"]) + list(hist[1]["train"]["correct"])
    assert max(abs(a - b) for a, b in zip(got_losses, ref_losses)) < 1e-4
    assert got_correct == ref_correct
    assert hist[1]["lr"] == pytest.approx(0.025)
    for (n, pr), (_, pm) in zip(ref.named_parameters(), m.named_parameters()):
        assert_close(f"param[{n}] after 2 epochs", pm.data, pr.data, 2e-4)


@pytest.mark.parametrize("decoupled,wd", [(True, 1e-2), (False, 0.0), (False, 5e-2)])
def test_fused_adamw_matches_torch_over_steps_and_under_graph_replay(decoupled, wd):
    """optim.AdamW / optim.Adam (utils_network.py:121,124) as one kernel over the flat buffers;
    the step count lives on the device, so a replayed HIP graph applies the right bias
    corrections."""
    from vit_torch_amd import CrossEntropyLoss, FusedAdamW, VisionTransformer
    from vit_torch_amd.graph import GraphedStep
    torch.manual_seed(2)
    cfg = dict(img_size=32, patch_size=8, embed_dim=64, depth=2, num_heads=2, num_classes=10)
    m = VisionTransformer(**cfg, compute_dtype="fp32").cuda()
    m.head = torch.nn.Linear(64, 10, bias=False).cuda()
    g = torch.Generator("cpu").manual_seed(0)
    data = [(torch.randn(16, 3, 32, 32, generator=g).cuda(), torch.randint(0, 10, (16,), generator=g).cuda())
            for _ in range(4)]
    m.engine()
    p0 = m.engine().pack.flat.clone()
    opt = FusedAdamW(m.parameters(), lr=3e-3, weight_decay=wd, decoupled=decoupled)
    crit = CrossEntropyLoss()
    grads = []
    for x, y in data:
        opt.zero_grad()
        crit(m(x), y).backward()
        grads.append(m.engine().pack.grad.clone())
        opt.step()
    got = m.engine().pack.flat.clone()
    # torch on the same gradient sequence
    p = torch.nn.Parameter(p0.clone())
    topt = (torch.optim.AdamW([p], lr=3e-3, weight_decay=wd) if decoupled
            else torch.optim.Adam([p], lr=3e-3, weight_decay=wd))
    # the update rule under test: torch is fed the gradient sequence the HIP path produced
    for gr in grads:
        p.grad = gr.clone()
        topt.step()
    assert_close("adam trajectory", got, p.detach(), 2e-6)
    # graph replay: two more steps through a captured graph == two more eager steps
    m2 = VisionTransformer(**cfg, compute_dtype="fp32").cuda()
    m2.head = torch.nn.Linear(64, 10, bias=False).cuda()
    m2.load_state_dict(m.state_dict())
    opt2 = FusedAdamW(m2.parameters(), lr=3e-3, weight_decay=wd, decoupled=decoupled)
    for x, y in data[:2]:                      # eager reference on the copy
        opt2.zero_grad(); crit(m2(x), y).backward(); opt2.step()
    want = m2.engine().pack.flat.clone()
    m3 = VisionTransformer(**cfg, compute_dtype="fp32").cuda()
    m3.head = torch.nn.Linear(64, 10, bias=False).cuda()
    m3.load_state_dict(m.state_dict())
    opt3 = FusedAdamW(m3.parameters(), lr=3e-3, weight_decay=wd, decoupled=decoupled)
    step = GraphedStep(m3, crit, opt3, *data[0], warmup=1)       # the warm-up step is data[0]'s update
    step(*data[1])
    assert_close("adam under graph replay", m3.engine().pack.flat, want, 1e-6)


def test_deferred_folds_leave_every_gradient_bit_identical(monkeypatch):
    """The engine queues the small folds of a backward pass (ops.FoldQueue) and runs them in one launch: all parameter
    gradients equal those of the fold-at-once form bit for bit."""
    from oracle import vit_ref
    from vit_torch_amd import CrossEntropyLoss, VisionTransformer
    cfg = dict(img_size=32, patch_size=8, in_chans=3, embed_dim=64, depth=3, num_heads=2)
    g = torch.Generator("cpu").manual_seed(3)
    x = torch.randn(6, 3, 32, 32, generator=g).cuda()
    y = torch.randint(0, 10, (6,), generator=g).cuda()
    ref = vit_ref.VisionTransformer(**cfg, apply_head=True)
    ref.head = vit_ref.get_classifier_head(64, 10)
    vit_ref.seeded_init_(ref, 1)
    grads = []
    for flag in ("0", "1"):
        monkeypatch.setenv("VITMI_DEFER_FOLDS", flag)
        m = VisionTransformer(**cfg, apply_head=True, compute_dtype="bf16", residual_dtype="auto")
        m.head = vit_ref.get_classifier_head(64, 10)
        m.load_state_dict(ref.state_dict())
        m = m.cuda()
        CrossEntropyLoss()(m(x), y).backward()
        assert (m.engine().folds is None) == (flag == "0")
        grads.append({n: p.grad.clone() for n, p in m.named_parameters()})
    for n in grads[0]:
        assert torch.equal(grads[0][n], grads[1][n]), n


# ------------------------------------------------------------- torch's accumulation contract ---
@pytest.mark.parametrize("compute", ["fp32", "bf16"])
def test_second_backward_accumulates(compute):
    """backward() twice without zero_grad(): .grad must hold the SUM (torch.autograd's contract).  The engine overwrites
    its flat gradient buffer, and after the first backward every .grad IS a view of that buffer: round 3 silently returned
    the second gradient alone.  Now the aliased spans are saved and added back (vitmi_axpy)."""
    from vit_torch_amd import CrossEntropyLoss
    _, m = make_pair(TINY, 10, compute)
    crit = CrossEntropyLoss()
    (xa, ya), (xb, yb) = data(6, 3, 32, 10, seed=3), data(6, 3, 32, 10, seed=4)
    singles = []
    for x, y in ((xa, ya), (xb, yb)):
        m.zero_grad()                                   # set_to_none: a fresh gradient
        crit(m(x.cuda()), y.cuda()).backward()
        singles.append([p.grad.clone() for p in m.parameters()])
    m.zero_grad()
    crit(m(xa.cuda()), ya.cuda()).backward()
    first_ptrs = [p.grad.data_ptr() for p in m.parameters()]
    crit(m(xb.cuda()), yb.cuda()).backward()            # no zero_grad in between
    for p, ptr, ga, gb in zip(m.parameters(), first_ptrs, *singles):
        assert p.grad.data_ptr() == ptr                 # still the engine's buffer, accumulated in place
        torch.testing.assert_close(p.grad, ga + gb, rtol=1e-6, atol=1e-7)
    # zero_grad(set_to_none=False) keeps the aliasing .grad tensors, zeroed: the next backward is a plain gradient again
    m.zero_grad(set_to_none=False)
    crit(m(xb.cuda()), yb.cuda()).backward()
    for p, gb in zip(m.parameters(), singles[1]):
        torch.testing.assert_close(p.grad, gb, rtol=1e-6, atol=1e-7)


# ------------------------------------------------------------- CLS-only last block (opt-in) ---
@pytest.mark.parametrize("compute,residual", [("fp32", "fp32"), ("bf16", "bf16"), ("bf16", "fp32")])
def test_cls_only_last_block_gives_the_same_step(compute, residual):
    """`cls_only_last_block=True`: the model returns norm(x)[:, 0], so the last block's attention output, proj and MLP are
    needed on the CLS row only (k / v still from all tokens) and only that row sends a gradient back.  Same weights, same
    batch: logits, loss and EVERY gradient (the last block's included: its weight gradients are sums over rows of which
    all but the CLS ones are exactly zero in the full computation) must equal the full computation's, and the oracle's."""
    from vit_torch_amd import CrossEntropyLoss, VisionTransformer, VisionModelZoo
    cfg = dict(img_size=64, patch_size=16, in_chans=3, embed_dim=128, depth=3, num_heads=2)      # N = 17: ragged M = 17 B
    ref, full = make_pair(cfg, 10, compute, residual)
    x, y = data(24, 3, 64, 10, seed=7)
    m = VisionTransformer(**cfg, apply_head=True, compute_dtype=compute, residual_dtype=residual, cls_only_last_block=True)
    m.head = VisionModelZoo.get_classifier_head(cfg["embed_dim"], 10)
    m.load_state_dict(ref.state_dict(), strict=True)
    m = m.cuda()
    crit = CrossEntropyLoss()
    res = []
    for mod in (full, m):
        out = mod(x.cuda())
        loss = crit(out, y.cuda())
        mod.zero_grad()
        loss.backward()
        res.append((out.detach().float().cpu(), loss.item(), {n: p.grad.detach().float().cpu() for n, p in mod.named_parameters()}))
    assert m.engine().cls_last and not full.engine().cls_last
    tol = 2e-5 if compute == "fp32" else 8e-3
    assert_close("logits", res[1][0], res[0][0], tol)
    assert abs(res[1][1] - res[0][1]) < (1e-5 if compute == "fp32" else 2e-3)
    gmax = max(g.norm().item() for g in res[0][2].values())
    for n, g0 in res[0][2].items():
        g1 = res[1][2][n]
        if g0.norm().item() < 1e-6 * gmax:
            assert g1.norm().item() < 1e-4 * gmax, n
            continue
        assert_close(f"grad[{n}]", g1, g0, 1e-4 if compute == "fp32" else 4e-2)
    # and against the oracle (fp32 mode: the parity claim holds with the option on)
    if compute == "fp32":
        lo = ref(x)
        lr = F.cross_entropy(lo, y)
        ref.zero_grad(); lr.backward()
        assert_close("logits vs oracle", res[1][0], lo.detach(), 1e-4)
        for (n, pr) in ref.named_parameters():
            if pr.grad is not None and pr.grad.abs().max().item() > 1e-9:
                assert_close(f"grad[{n}] vs oracle", res[1][2][n], pr.grad, 3e-4)


def test_cls_only_last_block_at_tile_kernel_width():
    """the CLS-only last block at embed_dim 768 and a batch whose M = 197 B is ragged (B = 16: 3 152 rows) — the
    d ln1 = dkv Wkv product is an fp32-output launch on the 256x256 tile kernel with padded rows, the shape that used to
    split K over a workspace strided by M.  Same step as the full computation."""
    from vit_torch_amd import CrossEntropyLoss, VisionTransformer, VisionModelZoo
    cfg = dict(img_size=224, patch_size=16, in_chans=3, embed_dim=768, depth=2, num_heads=12)
    ref, full = make_pair(cfg, 10, "bf16", "bf16")
    x, y = data(16, 3, 224, 10, seed=11)
    m = VisionTransformer(**cfg, apply_head=True, compute_dtype="bf16", residual_dtype="bf16", cls_only_last_block=True)
    m.head = VisionModelZoo.get_classifier_head(cfg["embed_dim"], 10)
    m.load_state_dict(ref.state_dict(), strict=True)
    m = m.cuda()
    crit = CrossEntropyLoss()
    res = []
    for mod in (full, m):
        out = mod(x.cuda())
        loss = crit(out, y.cuda())
        mod.zero_grad()
        loss.backward()
        res.append((out.detach().float().cpu(), {n: p.grad.detach().float().cpu() for n, p in mod.named_parameters()}))
    assert m.engine().cls_last
    assert_close("logits", res[1][0], res[0][0], 8e-3)
    gmax = max(g.norm().item() for g in res[0][1].values())
    for n, g0 in res[0][1].items():
        g1 = res[1][1][n]
        if g0.norm().item() < 1e-6 * gmax:
            assert g1.norm().item() < 1e-4 * gmax, n
            continue
        assert_close(f"grad[{n}]", g1, g0, 4e-2)


def test_cls_only_last_block_refuses_shapes_its_kernel_cannot_run():
    """The class-attention kernels take hd <= 64 and N <= 256: a head dimension beyond that is refused when the engine is
    built, a longer sequence (dino_vitb8 at 224 x 224: N = 785) before the first kernel of the forward is launched."""
    from vit_torch_amd import VisionTransformer
    from vit_torch_amd._lib import VitmiError
    m = VisionTransformer(img_size=32, patch_size=8, embed_dim=256, depth=1, num_heads=2, cls_only_last_block=True).cuda()
    with pytest.raises(VitmiError, match="cls_only_last_block"):
        m.engine()
    m = VisionTransformer(img_size=224, patch_size=8, embed_dim=128, depth=1, num_heads=2, cls_only_last_block=True).cuda()
    with pytest.raises(VitmiError, match="cls_only_last_block"):
        m(torch.randn(1, 3, 224, 224, device="cuda"))
    m(torch.randn(2, 3, 96, 96, device="cuda"))          # N = 145: fine
