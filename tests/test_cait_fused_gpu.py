"""Fused talking-heads attention (vitmi_th_attn_fwd / _bwd, cait_fused.hip; reference
/root/reference/models/cait.py:111-128) against fp32 torch autograd of the reference's formula on the same
bf16-rounded inputs, and against the three-call form it replaces (gemm_small q k^T -> th_softmax -> gemm_small P' v).
Shapes: cait_S24_224's (N = 196, H = 8, hd = 48) and ragged ones (N = 100: a last row block of 4 rows, N = 8)."""
import pytest
import torch

from util import assert_close, cosine

pytestmark = pytest.mark.gpu
bt = torch.bfloat16


def reference(qkv, Wl, bl, Ww, bw, dO, scale):
    """models/cait.py:111-128 in fp32 on the bf16-rounded operands; returns O and every gradient."""
    B, N, _, H, hd = qkv.shape
    x = qkv.float().clone().requires_grad_(True)
    Wl, bl, Ww, bw = (t.clone().requires_grad_(True) for t in (Wl, bl, Ww, bw))
    q, k, v = x[:, :, 0].permute(0, 2, 1, 3) * scale, x[:, :, 1].permute(0, 2, 1, 3), x[:, :, 2].permute(0, 2, 1, 3)
    attn = q @ k.transpose(-2, -1)                                   # [B,H,N,N]
    attn = (attn.permute(0, 2, 3, 1) @ Wl.t() + bl).permute(0, 3, 1, 2)
    attn = attn.softmax(dim=-1)
    attn = (attn.permute(0, 2, 3, 1) @ Ww.t() + bw).permute(0, 3, 1, 2)
    out = (attn @ v).transpose(1, 2).reshape(B, N, H * hd)
    out.backward(dO.float().reshape(B, N, H * hd))
    return out.detach(), x.grad, Wl.grad, bl.grad, Ww.grad, bw.grad


def make(B, N, seed=0, H=8, hd=48):
    g = torch.Generator("cpu").manual_seed(seed)
    qkv = (torch.randn(B, N, 3, H, hd, generator=g) * 0.7).to(bt)
    dO = torch.randn(B, N, H, hd, generator=g).to(bt)
    eye = torch.eye(H)
    Wl = eye + 0.3 * torch.randn(H, H, generator=g)
    Ww = eye + 0.3 * torch.randn(H, H, generator=g)
    bl, bw = 0.2 * torch.randn(H, generator=g), 0.05 * torch.randn(H, generator=g)
    return qkv, dO, Wl, bl, Ww, bw


def fused(ops, qkv, dO, Wl, bl, Ww, bw, scale):
    B, N, _, H, hd = qkv.shape
    D, D3, NS = H * hd, 3 * H * hd, 224
    dev = "cuda"
    q, do = qkv.cuda().contiguous(), dO.cuda().contiguous()
    W = [t.cuda().contiguous() for t in (Wl, bl, Ww, bw)]
    O = torch.full((B, N, H, hd), float("nan"), device=dev, dtype=bt)
    ops.th_attn_fwd(q, *W, O, B, H, N, hd, scale)
    dqkv = torch.full((B * N, D3), float("nan"), device=dev, dtype=bt)
    dS = torch.zeros((B, H, N, NS), device=dev, dtype=bt)
    Pm = torch.zeros((B, H, N, NS), device=dev, dtype=bt)
    gr = [torch.full((H, H), float("nan"), device=dev), torch.full((H,), float("nan"), device=dev),
          torch.full((H, H), float("nan"), device=dev), torch.full((H,), float("nan"), device=dev)]
    ops.th_attn_bwd(q, do, *W, dqkv, dS, Pm, NS, *gr, B, H, N, hd, scale)
    torch.cuda.synchronize()
    return O.float().cpu().reshape(B, N, D), dqkv.float().cpu().view(B, N, 3, H, hd), [t.cpu() for t in gr]


@pytest.mark.parametrize("B,N", [(3, 196), (2, 100), (5, 8), (2, 224)])
def test_fused_talking_heads_attention_matches_the_reference_formula(B, N):
    from vit_torch_amd import ops as O_
    assert O_.th_attn_supported(bt, 8, N, 48)
    qkv, dO, Wl, bl, Ww, bw = make(B, N, seed=N)
    scale = 48 ** -0.5
    want = reference(qkv, Wl, bl, Ww, bw, dO, scale)
    out, dqkv, (dWl, dbl, dWw, dbw) = fused(O_, qkv, dO, Wl, bl, Ww, bw, scale)
    e_out = assert_close("O", out, want[0], 1.5e-2)
    e_dq = assert_close("dq", dqkv[:, :, 0], want[1][:, :, 0], 2e-2)
    e_dk = assert_close("dk", dqkv[:, :, 1], want[1][:, :, 1], 2e-2)
    e_dv = assert_close("dv", dqkv[:, :, 2], want[1][:, :, 2], 2e-2)
    e_wl = assert_close("dWl", dWl, want[2], 2e-2)
    e_ww = assert_close("dWw", dWw, want[4], 2e-2)
    e_bw = assert_close("dbw", dbw, want[5], 2e-2)
    assert dbl.abs().max().item() == 0.0                     # analytically zero (softmax shift invariance): written as zeros
    assert want[3].abs().max().item() < 1e-3 * want[2].abs().max().item()
    for name, a, b_ in (("dq", dqkv[:, :, 0], want[1][:, :, 0]), ("dk", dqkv[:, :, 1], want[1][:, :, 1]), ("dWl", dWl, want[2])):
        assert cosine(a, b_) > 0.9995, name
    print(f"\nfused talking-heads B={B} N={N}: O {e_out:.2e}, dq {e_dq:.2e}, dk {e_dk:.2e}, dv {e_dv:.2e}, dWl {e_wl:.2e}, dWw {e_ww:.2e}, dbw {e_bw:.2e}")


def test_unsupported_shapes_are_refused_and_fall_back():
    from vit_torch_amd import ops as O_
    assert not O_.th_attn_supported(torch.float32, 8, 196, 48)      # fp32 parity mode: the three-call form
    assert not O_.th_attn_supported(bt, 4, 196, 48)                 # cait_XXS24: 4 heads
    assert not O_.th_attn_supported(bt, 8, 576, 48)                 # 384x384 variants: 576 tokens
    assert not O_.th_attn_supported(bt, 8, 197, 48)


def test_cait_engine_fused_and_three_call_forms_agree(monkeypatch):
    """cait_S24-shaped blocks (H = 8, hd = 48, N = 196) through the whole engine: the fused attention against the
    three-call form, same weights and batch: logits, loss and every gradient."""
    from functools import partial
    import torch.nn as nn
    from oracle.vit_ref import seeded_init_
    from vit_torch_amd import CrossEntropyLoss, cait_models
    cfg = dict(img_size=224, patch_size=16, embed_dim=384, depth=2, num_heads=8, mlp_ratio=4, qkv_bias=True,
               norm_layer=partial(nn.LayerNorm, eps=1e-6), init_scale=1e-1, depth_token_only=1, num_classes=10)
    g = torch.Generator("cpu").manual_seed(3)
    x, y = torch.randn(3, 3, 224, 224, generator=g).cuda(), torch.randint(0, 10, (3,), generator=g).cuda()
    res = {}
    for form in ("1", "0"):
        monkeypatch.setenv("VITMI_TH_FUSED", form)
        m = cait_models(**cfg, compute_dtype="bf16", residual_dtype="bf16")
        seeded_init_(m, 4)
        m = m.cuda()
        assert m.engine().fused_th == (form == "1")
        out = m(x)
        loss = CrossEntropyLoss()(out, y)
        loss.backward()
        res[form] = (out.detach().float().cpu(), loss.item(), {n: p.grad.detach().float().cpu() for n, p in m.named_parameters()})
    assert_close("logits", res["1"][0], res["0"][0], 2e-2)
    assert abs(res["1"][1] - res["0"][1]) < 5e-3
    worst = (1.0, "")
    for n, ga in res["1"][2].items():
        gb = res["0"][2][n]
        if n.endswith(("proj_l.bias", "attn.k.bias")) or gb.norm().item() == 0:      # analytically zero gradients
            continue
        c = cosine(ga, gb)
        if c < worst[0]:
            worst = (c, n)
    assert worst[0] > 0.999, worst
    print(f"\nfused vs three-call engine: logits {(res['1'][0] - res['0'][0]).abs().max().item():.2e}, worst gradient cosine {worst}")
