"""GPU parity of every C-ABI op against plain fp32 PyTorch math on the CPU.

fp32 kernels: tolerance 2e-5 relative to max|ref| (fp32 accumulation order).
bf16 kernels: inputs are rounded to bf16 BEFORE the reference sees them; the
tolerance 1.5e-2 covers the bf16 rounding of the stored output (2^-8 = 3.9e-3
of each value) plus bf16 rounding of in-kernel operands (P in attention).
"""
import math

import pytest
import torch
import torch.nn.functional as F

from util import assert_close, bf16_round, rel_err

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 2e-5, torch.bfloat16: 1.5e-2}


@pytest.fixture(scope="module")
def ops(lib):
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from vit_torch_amd import ops as _ops
    return _ops


def gen(shape, seed, scale=1.0):
    g = torch.Generator("cpu").manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def dev(t, dt=torch.float32):
    return t.to(device="cuda", dtype=dt)


def gelu_grad(x):
    x = x.clone().requires_grad_(True)
    F.gelu(x).sum().backward()
    return x.grad


# ------------------------------------------------------------------ gemm ---
SHAPES = [(64, 64, 32), (197, 10, 768), (130, 75, 40), (5, 384, 96), (256, 3, 8), (333, 129, 65)]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("akm,bkm", [(True, True), (True, False), (False, False), (False, True)])
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_gemm_generic_layouts(ops, dt, akm, bkm, M, N, K):
    from vit_torch_amd._lib import GEMM_GENERIC
    a = gen((M, K), 1)
    b = gen((N, K), 2)
    if dt == torch.bfloat16:
        a, b = bf16_round(a), bf16_round(b)
    want = a @ b.t()
    A = dev(a if akm else a.t().contiguous(), dt)
    B = dev(b if bkm else b.t().contiguous(), dt)
    C = torch.empty((M, N), device="cuda", dtype=torch.float32)
    ops.gemm(A, B, C, a_kmajor=akm, b_kmajor=bkm, impl=GEMM_GENERIC)
    assert_close("gemm", C, want, 2e-5 if dt == torch.float32 else 1e-5 * math.sqrt(K) + 2e-5)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gemm_epilogues(ops, dt):
    from vit_torch_amd._lib import (EPI_BIAS_GELU, EPI_DGELU, EPI_PATCH_POS, EPI_RESIDUAL, GEMM_GENERIC)
    M, N, K = 150, 72, 48
    rd = (lambda t: bf16_round(t)) if dt == torch.bfloat16 else (lambda t: t)
    a, b = rd(gen((M, K), 3)), rd(gen((N, K), 4, 0.2))
    bias = gen((N,), 5)
    A, B, bias_d = dev(a, dt), dev(b, dt), dev(bias)
    acc = a @ b.t()
    tol = TOL[dt]
    # bias + alpha, bf16/fp32 output
    C = torch.empty((M, N), device="cuda", dtype=dt)
    ops.gemm(A, B, C, bias=bias_d, alpha=0.5, impl=GEMM_GENERIC)
    assert_close("store", C, 0.5 * acc + bias, tol)
    # accumulate into fp32
    C32 = dev(gen((M, N), 6))
    ops.gemm(A, B, C32, accumulate=True, impl=GEMM_GENERIC)
    assert_close("accumulate", C32, gen((M, N), 6) + acc, tol)
    # bias + GELU with the pre-activation as second output
    H = torch.empty((M, N), device="cuda", dtype=dt)
    P = torch.empty((M, N), device="cuda", dtype=dt)
    ops.gemm(A, B, H, epilogue=EPI_BIAS_GELU, bias=bias_d, C2=P, impl=GEMM_GENERIC)
    assert_close("pre", P, acc + bias, tol)
    assert_close("gelu", H, F.gelu(rd(acc + bias)), tol)
    # residual with LayerScale, both residual dtypes
    for rdt in ([torch.float32] if dt == torch.float32 else [torch.float32, torch.bfloat16]):
        r = gen((M, N), 7)
        if rdt == torch.bfloat16:
            r = bf16_round(r)
        gam = gen((N,), 8)
        X = torch.empty((M, N), device="cuda", dtype=rdt)
        ops.gemm(A, B, X, epilogue=EPI_RESIDUAL, bias=bias_d, R=dev(r, rdt), gamma=dev(gam), impl=GEMM_GENERIC)
        assert_close(f"residual[{rdt}]", X, r + gam * (acc + bias), TOL[rdt])
        ops.gemm(A, B, X, epilogue=EPI_RESIDUAL, R=dev(r, rdt), impl=GEMM_GENERIC)
        assert_close(f"residual-plain[{rdt}]", X, r + acc, TOL[rdt])
        # DropPath: per-sample factor on the branch (6 samples x 25 rows), one sample dropped
        rsc = torch.tensor([1.25, 0.0, 1.25, 1.25, 0.0, 1.25])
        ops.gemm(A, B, X, epilogue=EPI_RESIDUAL, bias=bias_d, R=dev(r, rdt), rowscale=dev(rsc), rows_per_group=25,
                 impl=GEMM_GENERIC)
        want = r + rsc.repeat_interleave(25)[:, None] * (acc + bias)
        assert_close(f"residual-droppath[{rdt}]", X, want, TOL[rdt])
        assert torch.equal(X[25:50].float().cpu(), r[25:50]), "a dropped sample must pass the residual through"
    # dgelu
    aux = rd(gen((M, N), 9))
    Dg = torch.empty((M, N), device="cuda", dtype=dt)
    ops.gemm(A, B, Dg, epilogue=EPI_DGELU, aux=dev(aux, dt), impl=GEMM_GENERIC)
    assert_close("dgelu", Dg, acc * gelu_grad(aux), tol)
    # the pair with the derivative kept instead of the pre-activation (aux_is_derivative)
    ops.gemm(A, B, H, epilogue=EPI_BIAS_GELU, bias=bias_d, C2=P, impl=GEMM_GENERIC, aux_deriv=True)
    assert_close("gelu[deriv]", H, F.gelu(acc + bias), tol)
    assert_close("gelu'[deriv]", P, gelu_grad(acc + bias), tol)
    ops.gemm(A, B, Dg, epilogue=EPI_DGELU, aux=dev(aux, dt), impl=GEMM_GENERIC, aux_deriv=True)
    assert_close("dgelu[deriv]", Dg, acc * aux, tol)
    # patch + pos (+cls): M = 6 images x 25 tokens
    n_tok = 25
    pos, cls = gen((n_tok, N), 10), gen((N,), 11)
    t = torch.arange(M) % n_tok
    want = acc + bias + pos[t]
    want[t == 0] = cls + pos[0]
    for rdt in ([torch.float32] if dt == torch.float32 else [torch.float32, torch.bfloat16]):
        X = torch.empty((M, N), device="cuda", dtype=rdt)
        ops.gemm(A, B, X, epilogue=EPI_PATCH_POS, bias=bias_d, pos=dev(pos), n_tok=n_tok, cls=dev(cls), impl=GEMM_GENERIC)
        assert_close(f"patch_pos[{rdt}]", X, want, TOL[rdt])
        ops.gemm(A, B, X, epilogue=EPI_PATCH_POS, bias=bias_d, pos=dev(pos), n_tok=n_tok, impl=GEMM_GENERIC)
        assert_close(f"patch_pos_nocls[{rdt}]", X, acc + bias + pos[t], TOL[rdt])


# ------------------------------------------------------------- layernorm ---
@pytest.mark.parametrize("M,D", [(7, 64), (197, 768), (33, 384), (10, 96), (5, 1536), (4, 2048)])
@pytest.mark.parametrize("xdt,ydt", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16),
                                     (torch.bfloat16, torch.bfloat16)])
def test_layernorm_fwd(ops, M, D, xdt, ydt):
    x = gen((M, D), 1) * 2 + 0.5
    if xdt == torch.bfloat16:
        x = bf16_round(x)
    g, b = 1 + 0.1 * gen((D,), 2), 0.1 * gen((D,), 3)
    want = F.layer_norm(x, (D,), g, b, eps=1e-6)
    y = torch.empty((M, D), device="cuda", dtype=ydt)
    mean = torch.empty(M, device="cuda")
    rstd = torch.empty(M, device="cuda")
    ops.layernorm_fwd(dev(x, xdt), dev(g), dev(b), y, mean, rstd, 1e-6)
    assert_close("ln.y", y, want, TOL[ydt])
    assert_close("ln.mean", mean, x.mean(-1), 1e-5)
    assert_close("ln.rstd", rstd, (x.var(-1, unbiased=False) + 1e-6).rsqrt(), 1e-5)


def test_layernorm_fwd_strided_cls_rows(ops):
    B, N, D = 6, 5, 64
    x = gen((B, N, D), 4)
    g, b = 1 + 0.1 * gen((D,), 2), 0.1 * gen((D,), 3)
    y = torch.empty((B, D), device="cuda")
    ops.layernorm_fwd(dev(x), dev(g), dev(b), y, None, None, 1e-5, M=B, D=D, x_stride=N * D, y_stride=D)
    assert_close("ln.cls", y, F.layer_norm(x[:, 0], (D,), g, b, eps=1e-5), 2e-5)


@pytest.mark.parametrize("M,D", [(9, 64), (394, 768), (1030, 96), (2100, 384)])
@pytest.mark.parametrize("T,R", [(torch.float32, torch.float32), (torch.bfloat16, torch.float32),
                                 (torch.bfloat16, torch.bfloat16)])
def test_layernorm_bwd(ops, M, D, T, R):
    rT = (lambda t: bf16_round(t)) if T == torch.bfloat16 else (lambda t: t)
    rR = (lambda t: bf16_round(t)) if R == torch.bfloat16 else (lambda t: t)
    x = rR(gen((M, D), 1) * 1.5 + 0.3)
    dy = rT(gen((M, D), 2))
    gin = rR(gen((M, D), 3))
    g = 1 + 0.1 * gen((D,), 4)
    b = 0.1 * gen((D,), 5)
    xr = x.clone().requires_grad_(True)
    gr = g.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    F.layer_norm(xr, (D,), gr, br, eps=1e-6).backward(dy)
    mean = x.mean(-1)
    rstd = (x.var(-1, unbiased=False) + 1e-6).rsqrt()
    G = dev(gin, R)
    Gb = torch.empty((M, D), device="cuda", dtype=T) if T != R else None
    dg = torch.empty(D, device="cuda")
    db = torch.empty(D, device="cuda")
    gsum = torch.empty(D, device="cuda")
    ops.layernorm_bwd(dev(dy, T), dev(x, R), dev(mean), dev(rstd), dev(g), G, G, Gb, dg, db, gsum=gsum)
    assert_close("ln.g_out", G, gin + xr.grad, TOL[R])
    assert_close("ln.gsum", gsum, (gin + xr.grad).sum(0), 2e-3 if R == torch.bfloat16 else 1e-4)
    if Gb is not None:
        assert_close("ln.gb_out", Gb, gin + xr.grad, TOL[T])
    assert_close("ln.dgamma", dg, gr.grad, 1e-4)
    assert_close("ln.dbeta", db, br.grad, 1e-4)


def test_layernorm_bwd_gb_row_and_column_scale(ops):
    """Gb (and its column sum) carry LayerScale x DropPath of the consuming branch; g_out does not."""
    M, D, rpg = 96, 128, 16
    x, dy, gin = gen((M, D), 1), gen((M, D), 2), gen((M, D), 3)
    g = 1 + 0.1 * gen((D,), 4)
    col = gen((D,), 5)
    row = torch.tensor([2.0, 0.0, 2.0, 2.0, 0.0, 0.0])
    xr = x.clone().requires_grad_(True)
    F.layer_norm(xr, (D,), g, torch.zeros(D), eps=1e-6).backward(dy)
    mean, rstd = x.mean(-1), (x.var(-1, unbiased=False) + 1e-6).rsqrt()
    G = torch.empty((M, D), device="cuda")
    Gb = torch.empty((M, D), device="cuda", dtype=torch.bfloat16)
    dg, db, gsum = (torch.empty(D, device="cuda") for _ in range(3))
    ops.layernorm_bwd(dev(dy), dev(x), dev(mean), dev(rstd), dev(g), dev(gin), G, Gb, dg, db, gsum=gsum,
                      gb_scale=dev(col), gb_rowscale=dev(row), rows_per_group=rpg)
    want = gin + xr.grad
    wb = want * col * row.repeat_interleave(rpg)[:, None]
    assert_close("g_out", G, want, 2e-5)
    assert_close("gb_out", Gb, wb, TOL[torch.bfloat16])
    assert_close("gsum", gsum, wb.sum(0), 1e-4)


def test_scale_cast_row_and_column(ops):
    M, N, rpg = 60, 96, 12
    x, col = gen((M, N), 1), gen((N,), 2)
    row = torch.tensor([0.0, 1.5, 1.5, 0.0, 1.5])
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    ops.scale_cast(dev(x), out, dev(col), M=M, N=N, rowscale=dev(row), rows_per_group=rpg)
    assert_close("scale_cast", out, x * col * row.repeat_interleave(rpg)[:, None], TOL[torch.bfloat16])
    o32 = torch.empty((M, N), device="cuda")
    ops.scale_cast(dev(x), o32, None, M=M, N=N, rowscale=dev(row), rows_per_group=rpg)
    assert torch.equal(o32.cpu(), x * row.repeat_interleave(rpg)[:, None])


def test_layernorm_bwd_cls_rows_no_gin(ops):
    B, N, D = 6, 5, 64
    x = gen((B, N, D), 1)
    dy = gen((B, D), 2)
    g = 1 + 0.1 * gen((D,), 4)
    xr = x[:, 0].clone().requires_grad_(True)
    F.layer_norm(xr, (D,), g, torch.zeros(D), eps=1e-6).backward(dy)
    mean = x[:, 0].mean(-1)
    rstd = (x[:, 0].var(-1, unbiased=False) + 1e-6).rsqrt()
    G = torch.zeros((B * N, D), device="cuda")
    dg = torch.empty(D, device="cuda")
    db = torch.empty(D, device="cuda")
    ops.layernorm_bwd(dev(dy), dev(x), dev(mean), dev(rstd), dev(g), None, G, None, dg, db,
                      M=B, D=D, dy_stride=D, x_stride=N * D, g_stride=N * D)
    want = torch.zeros(B, N, D)
    want[:, 0] = xr.grad
    assert_close("ln.cls.g", G.view(B, N, D), want, 2e-5)
    assert_close("ln.cls.db", db, dy.sum(0), 1e-5)


# -------------------------------------------------------------- attention ---
def attn_ref(qkv, B, N, H, hd, scale):
    """softmax(scale * q k^T) v on [B,N,3,H,hd] (models/swin.py:120-142 without bias/mask)."""
    q, k, v = qkv.view(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-2, -1)) * scale
    lse = torch.logsumexp(s, dim=-1)
    o = (s.softmax(-1) @ v).transpose(1, 2).reshape(B, N, H * hd)
    return o, lse


ATTN_CASES = [(2, 197, 3, 64), (3, 5, 2, 64), (1, 145, 2, 64), (2, 49, 3, 32), (1, 300, 1, 64),
              (2, 64, 2, 32), (1, 785, 1, 64), (1, 1, 1, 64)]


@pytest.fixture(params=["fused", "split"])
def attn_bwd_mode(request, lib):
    """bf16 attention backward: one workgroup per (image, head) with dQ in LDS (N <= 256),
    or the dkdv + dq kernel pair; the fp32 kernels ignore the switch."""
    import ctypes
    from vit_torch_amd import _lib
    raw = ctypes.CDLL(str(_lib.LIB_PATH))
    raw.vitmi_debug_attn_bwd(1 if request.param == "fused" else 0)
    yield request.param
    raw.vitmi_debug_attn_bwd(-1)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,N,H,hd", ATTN_CASES + [(2, 224, 2, 64), (1, 256, 2, 64), (2, 256, 1, 32), (1, 33, 1, 32)])
def test_attention_fwd_bwd(ops, attn_bwd_mode, dt, B, N, H, hd):
    if dt == torch.float32 and attn_bwd_mode == "split":
        pytest.skip("fp32 kernels have one backward path")
    scale = hd ** -0.5
    qkv = gen((B, N, 3 * H * hd), 1)
    do = gen((B, N, H * hd), 2)
    if dt == torch.bfloat16:
        qkv, do = bf16_round(qkv), bf16_round(do)
    qr = qkv.clone().requires_grad_(True)
    o_ref, lse_ref = attn_ref(qr, B, N, H, hd, scale)
    o_ref.backward(do)
    QKV = dev(qkv, dt)
    O = torch.empty((B, N, H * hd), device="cuda", dtype=dt)
    lse = torch.empty(B * H * N, device="cuda")
    ops.attn_fwd(QKV, O, lse, B, N, H, hd, scale)
    tol = TOL[dt]
    assert_close("attn.out", O, o_ref.detach(), tol)
    assert_close("attn.lse", lse.view(B, H, N), lse_ref.detach(), 1e-5 if dt == torch.float32 else 2e-3)
    dqkv = torch.full((B, N, 3 * H * hd), float("nan"), device="cuda").to(dt)
    # feed backward the kernel's own (rounded) output, as the engine does
    ops.attn_bwd(QKV, O, dev(do, dt), lse, dqkv, B, N, H, hd, scale)
    g = qr.grad.view(B, N, 3, H, hd)
    d = dqkv.float().cpu().view(B, N, 3, H, hd)
    for i, nm in enumerate("qkv"):
        assert_close(f"attn.d{nm}", d[:, :, i], g[:, :, i], tol if dt == torch.float32 else 2.5e-2)
    if dt == torch.bfloat16:
        # fused qkv-bias gradient: per-workgroup column sums of dqkv, same dqkv bits as without
        rows = ops.attn_bwd_dbias_rows(B, N)
        part = torch.full((rows, 3 * H * hd), float("nan"), device="cuda")
        dqkv2 = torch.empty_like(dqkv)
        ops.attn_bwd(QKV, O, dev(do, dt), lse, dqkv2, B, N, H, hd, scale, dbias_part=part)
        assert torch.equal(dqkv2, dqkv)
        assert rows % B == 0
        per_img = part.view(B, rows // B, -1).sum(1)
        want = dqkv.float().view(B, N, -1).sum(1)
        # the kernel sums the fp32 values before they are rounded to bf16 for dqkv
        assert_close("attn.dbias_part", per_img, want, 1.5e-2)
        ref_sum = qr.grad.view(B * N, -1).sum(0)
        assert_close("attn.dbias", part.sum(0), ref_sum, 2.5e-2)
        if attn_bwd_mode == "fused" and N <= 224:
            # the whole-sequence kernel uses two identities of softmax attention instead of summing its
            # accumulators: sum_k dV[k] = sum_q dO[q] (rows of P sum to one) and sum_k dK[k] = 0 (rows of
            # dS sum to zero).  The k slice is written as exact zeros; the reference's is rounding noise.
            D = H * hd
            ksl = part.view(rows, 3, D)[:, 1]
            assert torch.count_nonzero(ksl).item() == 0
            assert ref_sum[D:2 * D].abs().max() <= 1e-4 * ref_sum.abs().max()
            vsum = do.view(B, N, D).sum(1)
            assert_close("attn.dbias_v", per_img.view(B, 3, D)[:, 2], vsum, 2e-3)


def test_attention_online_softmax_rescale_branch(ops):
    """Force the running max to jump in the LAST key tile (guide §5.4 rule 26)."""
    B, N, H, hd = 1, 200, 1, 64
    qkv = gen((B, N, 3 * H * hd), 3) * 0.5
    v = qkv.view(B, N, 3, H, hd)
    v[0, 197, 1, 0] = v[0, 7, 0, 0] * 6.0        # key 197 aligned with query 7 -> huge score
    qkv = bf16_round(qkv)
    o_ref, _ = attn_ref(qkv, B, N, H, hd, hd ** -0.5)
    O = torch.empty((B, N, H * hd), device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B * H * N, device="cuda")
    ops.attn_fwd(dev(qkv, torch.bfloat16), O, lse, B, N, H, hd, hd ** -0.5)
    assert_close("attn.rescale", O, o_ref, 1.5e-2)


# ------------------------------------------------------------ elementwise ---
@pytest.mark.parametrize("n", [4, 1000, 4099, 1 << 20])
def test_cast(ops, n):
    x = gen((n,), 1)
    y = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    ops.cast(dev(x), y)
    assert torch.equal(y.cpu(), x.to(torch.bfloat16))     # round-to-nearest-even, bit exact
    z = torch.empty(n, device="cuda")
    ops.cast(y, z)
    assert torch.equal(z.cpu(), x.to(torch.bfloat16).float())


@pytest.mark.parametrize("B,C,H,p,cls", [(2, 3, 32, 16, 1), (3, 3, 96, 8, 1), (2, 3, 56, 4, 0), (1, 5, 32, 16, 1)])
@pytest.mark.parametrize("channels_last", [False, True])
def test_patchify(ops, B, C, H, p, cls, channels_last):
    x = gen((B, C, H, H), 1)
    xd = dev(x)
    if channels_last:
        xd = xd.contiguous(memory_format=torch.channels_last)
    want = F.unfold(x, kernel_size=p, stride=p).transpose(1, 2)            # [B, L, C*p*p]
    if cls:
        want = torch.cat([torch.zeros(B, 1, want.shape[-1]), want], dim=1)
    out = torch.empty((want.shape[0] * want.shape[1], want.shape[2]), device="cuda")
    ops.patchify(xd, out, p, cls)
    assert torch.equal(out.cpu(), want.reshape(out.shape))
    outb = torch.empty(out.shape, device="cuda", dtype=torch.bfloat16)
    ops.patchify(xd, outb, p, cls)
    assert torch.equal(outb.cpu(), want.reshape(out.shape).to(torch.bfloat16))
    # rows at a padded stride: the columns past C*p*p are written as zeros (Swin's 48 -> 64)
    ld = (out.shape[1] + 31) // 32 * 32 + 32
    outp = torch.full((out.shape[0], ld), float("nan"), device="cuda", dtype=torch.bfloat16)
    ops.patchify(xd, outp, p, cls)
    assert torch.equal(outp[:, :out.shape[1]].cpu(), want.reshape(out.shape).to(torch.bfloat16))
    assert torch.count_nonzero(outp[:, out.shape[1]:]).item() == 0


def test_colsum_narrow_strided_rows(ops):
    """N <= 512 takes the rows-per-workgroup kernel; rows may sit at a larger stride."""
    M, N, ld = 1234, 96, 288
    x = bf16_round(gen((M, ld), 3))
    xd = dev(x, torch.bfloat16)
    out = torch.empty(N, device="cuda")
    ops.colsum(xd[:, 96:192], out, M=M, N=N, ld=ld)
    assert_close("colsum strided", out, x[:, 96:192].double().sum(0).float(), 1e-5)


@pytest.mark.parametrize("M,N", [(1, 8), (50, 10), (777, 768), (4000, 2304), (256, 197 * 64), (33, 7),
                                 (5000, 96), (50176, 96), (1234, 384), (300, 512), (3, 4)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_colsum(ops, M, N, dt):
    x = gen((M, N), 1)
    if dt == torch.bfloat16:
        x = bf16_round(x)
    out = torch.empty(N, device="cuda")
    ops.colsum(dev(x, dt), out)
    assert_close("colsum", out, x.double().sum(0).float(), 1e-5)


@pytest.mark.parametrize("B,K", [(1, 10), (256, 10), (130, 768), (7, 1000)])
def test_softmax_xent(ops, B, K):
    logits = gen((B, K), 1) * 3
    labels = torch.randint(0, K, (B,), generator=torch.Generator("cpu").manual_seed(2))
    lr = logits.clone().requires_grad_(True)
    loss_ref = F.cross_entropy(lr, labels)
    loss_ref.backward()
    loss = torch.empty(1 + B, device="cuda")
    correct = torch.empty(1 + B, device="cuda", dtype=torch.int32)
    dl = torch.empty((B, K), device="cuda")
    ops.softmax_xent(dev(logits), labels.cuda(), loss, dl, correct)
    assert abs(loss[0].item() - loss_ref.item()) <= 2e-6 * max(1.0, abs(loss_ref.item()))
    assert_close("xent.dlogits", dl, lr.grad, 2e-5)
    assert correct[0].item() == (logits.argmax(-1) == labels).sum().item()


def test_sgd_momentum_two_steps_match_torch(ops):
    n = 10007
    p0, g1, g2 = gen((n,), 1), gen((n,), 2), gen((n,), 3)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.SGD([ref], lr=0.01, momentum=0.9)
    p = dev(p0)
    buf = torch.zeros(n, device="cuda")
    shadow = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    for g in (g1, g2):
        ref.grad = g.clone()
        opt.step()
        ops.sgd_momentum(p, dev(g), buf, shadow, 0.01, 0.9)
    assert_close("sgd.p", p, ref.data, 1e-6)
    assert torch.equal(shadow.cpu(), p.cpu().to(torch.bfloat16))


# ------------------------------------------------------------- fast gemm ---
FAST_SHAPES = [(256, 256, 64), (512, 768, 768), (768, 512, 1024), (1024, 256, 128), (256, 512, 192),
               (512, 256, 6464), (256, 128, 64), (512, 384, 320)]


@pytest.fixture(params=["t1p0", "t1p1", "t1p2", "t1p3", "t2", "t1p1-oneshot", "t1p2-oneshot"])
def pipe(request, lib):
    """Run a test once per variant of the fast GEMM — 256x256 tiles with the simple /
    4-slab-ring / 64-deep-stage / 3-slab-ring (residual epilogue only) main loops, persistent (one
    workgroup per CU walking tiles, default) or one tile per workgroup, and 256x128 tiles with two workgroups per CU —
    through the diagnostic hooks the library exports."""
    import ctypes
    from vit_torch_amd import _lib as L
    raw = ctypes.CDLL(str(L.LIB_PATH))
    if request.param == "t2":
        raw.vitmi_debug_gemm_tile(2)
    else:
        raw.vitmi_debug_gemm_tile(1)
        raw.vitmi_debug_gemm_pipe(int(request.param[3]))
        raw.vitmi_debug_gemm_persist(0 if request.param.endswith("oneshot") else 1)   # grid = tiles (round-1 form)
    yield request.param
    raw.vitmi_debug_gemm_pipe(-1)
    raw.vitmi_debug_gemm_tile(-1)
    raw.vitmi_debug_gemm_persist(1)


@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
@pytest.mark.parametrize("M,N,K", FAST_SHAPES)
@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16])
def test_gemm_fast_layouts(ops, pipe, layout, M, N, K, cdt):
    from vit_torch_amd._lib import GEMM_FAST
    akm, bkm = {"nt": (True, True), "nn": (True, False), "tn": (False, False)}[layout]
    a, b = bf16_round(gen((M, K), 1)), bf16_round(gen((N, K), 2))
    want = a @ b.t()
    A = dev(a if akm else a.t().contiguous(), torch.bfloat16)
    B = dev(b if bkm else b.t().contiguous(), torch.bfloat16)
    C = torch.full((M, N), float("nan"), device="cuda").to(cdt)
    ops.gemm(A, B, C, a_kmajor=akm, b_kmajor=bkm, impl=GEMM_FAST)
    assert_close(f"gemm_fast[{layout}]", C, want, 1e-4 if cdt == torch.float32 else TOL[cdt])


RAGGED = [(392, 96, 96), (6272, 288, 96), (1000, 200, 160), (264, 8, 64), (3136, 384, 96), (520, 1152, 384)]


@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
@pytest.mark.parametrize("M,N,K", RAGGED)
def test_gemm_ragged_shapes_take_the_256x128_kernel(ops, layout, M, N, K):
    """Swin's C = 96/192 stages and odd batch sizes: M, N not multiples of the tile, K % 32 == 0.
    Surplus tile rows / columns are clamped duplicates whose stores are masked: the canary
    border around C must stay untouched."""
    from vit_torch_amd._lib import GEMM_FAST
    if layout == "tn":
        M, K = (M // 8 * 8), max(64, K // 32 * 32)
    akm, bkm = {"nt": (True, True), "nn": (True, False), "tn": (False, False)}[layout]
    assert ops.gemm_uses_fast(M, N, K, a_kmajor=akm, b_kmajor=bkm)
    a, b = bf16_round(gen((M, K), 1)), bf16_round(gen((N, K), 2))
    want = a @ b.t()
    A = dev(a if akm else a.t().contiguous(), torch.bfloat16)
    B = dev(b if bkm else b.t().contiguous(), torch.bfloat16)
    for cdt in (torch.float32, torch.bfloat16):
        buf = torch.full((M + 2, N + 16), 777.0, device="cuda").to(cdt)
        C = buf[1:M + 1, 8:8 + N]
        ops.gemm(A, B, C, a_kmajor=akm, b_kmajor=bkm, impl=GEMM_FAST)
        assert_close(f"ragged[{layout}]", C, want, 1e-4 if cdt == torch.float32 else TOL[cdt])
        border = buf.clone()
        border[1:M + 1, 8:8 + N] = 777.0
        assert (border == 777.0).all(), "store outside the M x N result"


def test_gemm_ragged_epilogues(ops):
    from vit_torch_amd._lib import EPI_BIAS_GELU, EPI_DGELU, EPI_RESIDUAL, GEMM_FAST
    M, N, K = 1000, 96, 160
    bt = torch.bfloat16
    a, b = bf16_round(gen((M, K), 3)), bf16_round(gen((N, K), 4, 0.2))
    bias = gen((N,), 5)
    A, B, Bt, bias_d = dev(a, bt), dev(b, bt), dev(b.t().contiguous(), bt), dev(bias)
    acc = a @ b.t()
    H = torch.empty((M, N), device="cuda", dtype=bt)
    P = torch.empty((M, N), device="cuda", dtype=bt)
    ops.gemm(A, B, H, epilogue=EPI_BIAS_GELU, bias=bias_d, C2=P, impl=GEMM_FAST)
    assert_close("pre", P, acc + bias, TOL[bt])
    assert_close("gelu", H, F.gelu(bf16_round(acc + bias)), TOL[bt])
    r, gam = gen((M, N), 7), gen((N,), 8)
    rsc = torch.tensor([1.25, 0.0, 1.25, 0.0, 1.25])
    X = torch.empty((M, N), device="cuda")
    ops.gemm(A, B, X, epilogue=EPI_RESIDUAL, bias=bias_d, R=dev(r), gamma=dev(gam), rowscale=dev(rsc),
             rows_per_group=200, impl=GEMM_FAST)
    assert_close("residual", X, r + rsc.repeat_interleave(200)[:, None] * gam * (acc + bias), 1e-4)
    aux = bf16_round(gen((M, N), 9))
    Dg = torch.empty((M, N), device="cuda", dtype=bt)
    ops.gemm(A, Bt, Dg, b_kmajor=False, epilogue=EPI_DGELU, aux=dev(aux, bt), impl=GEMM_FAST)
    assert_close("dgelu", Dg, acc * gelu_grad(aux), TOL[bt])
    ops.gemm(A, B, H, epilogue=EPI_BIAS_GELU, bias=bias_d, C2=P, impl=GEMM_FAST, aux_deriv=True)
    assert_close("gelu[deriv]", H, F.gelu(acc + bias), TOL[bt])
    assert_close("gelu'[deriv]", P, gelu_grad(acc + bias), TOL[bt])
    Dd = torch.empty_like(Dg)
    ops.gemm(A, Bt, Dd, b_kmajor=False, epilogue=EPI_DGELU, aux=dev(aux, bt), impl=GEMM_FAST, aux_deriv=True)
    assert_close("dgelu[deriv]", Dd, acc * aux, TOL[bt])
    # fused bias gradient on a ragged shape: partial rows of 128, the last one short
    part = torch.full(((M + 127) // 128, N), float("nan"), device="cuda")
    Dg2 = torch.empty_like(Dg)
    ops.gemm(A, Bt, Dg2, b_kmajor=False, epilogue=EPI_DGELU, aux=dev(aux, bt), impl=GEMM_FAST, colsum_part=part)
    assert torch.equal(Dg2, Dg)
    want = F.pad(acc * gelu_grad(aux), (0, 0, 0, part.shape[0] * 128 - M)).view(-1, 128, N).sum(1)
    assert_close("dgelu.colsum_part(ragged)", part, want, 2e-3)
    # weight-gradient form with split-K: [N x K_out] = dy^T x over M tokens
    W = torch.empty((N, K), device="cuda")
    Mk = M // 32 * 32 - 32                           # 960 tokens: an odd number of 32-deep slabs per split
    dy = bf16_round(gen((Mk, N), 10))
    ops.gemm(dev(dy, bt), A[:Mk], W, a_kmajor=False, b_kmajor=False, impl=GEMM_FAST)
    assert_close("wgrad", W, dy.t() @ a[:Mk], 1e-4)


def test_gemm_split_tail_matches_whole_tile_launch(ops, lib):
    """591 tiles on 256 CUs = 2 full rounds + 79 tiles: the remainder is contracted in three
    k-slices and finished by the row-wise epilogue kernel.  Same results (up to the fp32
    summation order) as the one-launch path, for the plain store and the residual epilogue."""
    import ctypes
    from vit_torch_amd import _lib
    from vit_torch_amd._lib import EPI_RESIDUAL, GEMM_FAST
    raw = ctypes.CDLL(str(_lib.LIB_PATH))
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("tile counts below are chosen for 256 CUs")
    M, N, K = 197 * 256, 768, 768
    bt = torch.bfloat16
    A = (torch.randn(M, K, device="cuda") * 0.5).to(bt)
    Bw = (torch.randn(N, K, device="cuda") * 0.05).to(bt)
    Bt = Bw.t().contiguous()
    bias = torch.randn(N, device="cuda")
    R = torch.randn(M, N, device="cuda")
    gam = torch.randn(N, device="cuda")
    rsc = (torch.rand(256, device="cuda") > 0.3).float() * 1.25
    outs = {}
    for mode in (0, 1):
        raw.vitmi_debug_gemm_tail(mode)
        C1 = torch.empty((M, N), device="cuda", dtype=bt)
        ops.gemm(A, Bw, C1, bias=bias, impl=GEMM_FAST)                                   # NT store
        C2 = torch.empty((M, N), device="cuda", dtype=bt)
        ops.gemm(A, Bt, C2, b_kmajor=False, impl=GEMM_FAST)                              # NN store
        X = torch.empty((M, N), device="cuda")
        F1 = torch.empty((M, N), device="cuda", dtype=bt)
        ops.gemm(A, Bw, X, epilogue=EPI_RESIDUAL, bias=bias, R=R, gamma=gam, C2=F1, rowscale=rsc,
                 rows_per_group=197, impl=GEMM_FAST)
        outs[mode] = (C1, C2, X, F1)
    raw.vitmi_debug_gemm_tail(-1)
    for name, a, b in zip(("nt store", "nn store", "residual", "residual C2"), outs[0], outs[1]):
        assert_close(name, b, a.float().cpu(), 8e-3 if b.dtype == bt else 1e-5)   # bf16: one ulp at the maximum
    # and against fp32 math on a slice that lies in the tail tiles (last rows)
    ref = (A[-512:].float() @ Bw.float().t() + bias).cpu()
    assert_close("tail rows vs fp32", outs[1][0][-512:], ref, TOL[bt])


def test_gemm_fast_epilogues(ops, pipe):
    from vit_torch_amd._lib import (EPI_BIAS_GELU, EPI_DGELU, EPI_PATCH_POS, EPI_RESIDUAL, GEMM_FAST)
    M, N, K = 512, 256, 192
    bt = torch.bfloat16
    a, b = bf16_round(gen((M, K), 3)), bf16_round(gen((N, K), 4, 0.2))
    bias = gen((N,), 5)
    A, B, Bt, bias_d = dev(a, bt), dev(b, bt), dev(b.t().contiguous(), bt), dev(bias)
    acc = a @ b.t()
    tol = TOL[bt]
    C = torch.empty((M, N), device="cuda", dtype=bt)
    ops.gemm(A, B, C, bias=bias_d, alpha=0.5, impl=GEMM_FAST)
    assert_close("store", C, 0.5 * acc + bias, tol)
    C32 = dev(gen((M, N), 6))
    ops.gemm(A, B, C32, accumulate=True, impl=GEMM_FAST)
    assert_close("accumulate", C32, gen((M, N), 6) + acc, 1e-4)
    H = torch.empty((M, N), device="cuda", dtype=bt)
    P = torch.empty((M, N), device="cuda", dtype=bt)
    ops.gemm(A, B, H, epilogue=EPI_BIAS_GELU, bias=bias_d, C2=P, impl=GEMM_FAST)
    assert_close("pre", P, acc + bias, tol)
    assert_close("gelu", H, F.gelu(bf16_round(acc + bias)), tol)
    for rdt in (torch.float32, bt):
        r = gen((M, N), 7)
        if rdt == bt:
            r = bf16_round(r)
        gam = gen((N,), 8)
        X = torch.empty((M, N), device="cuda", dtype=rdt)
        ops.gemm(A, B, X, epilogue=EPI_RESIDUAL, bias=bias_d, R=dev(r, rdt), gamma=dev(gam), impl=GEMM_FAST)
        assert_close(f"residual[{rdt}]", X, r + gam * (acc + bias), TOL[rdt] if rdt == bt else 1e-4)
        ops.gemm(A, B, X, epilogue=EPI_RESIDUAL, R=dev(r, rdt), impl=GEMM_FAST)
        assert_close(f"residual-plain[{rdt}]", X, r + acc, TOL[rdt] if rdt == bt else 1e-4)
        rsc = torch.tensor([1.25, 0.0, 1.25, 1.25, 0.0, 1.25, 1.25, 1.25])      # 8 samples x 64 rows
        ops.gemm(A, B, X, epilogue=EPI_RESIDUAL, bias=bias_d, R=dev(r, rdt), gamma=dev(gam), rowscale=dev(rsc),
                 rows_per_group=64, impl=GEMM_FAST)
        want = r + rsc.repeat_interleave(64)[:, None] * gam * (acc + bias)
        assert_close(f"residual-droppath[{rdt}]", X, want, TOL[rdt] if rdt == bt else 1e-4)
        assert torch.equal(X[64:128].float().cpu(), r[64:128])
    aux = bf16_round(gen((M, N), 9))
    Dg = torch.empty((M, N), device="cuda", dtype=bt)
    ops.gemm(A, Bt, Dg, b_kmajor=False, epilogue=EPI_DGELU, aux=dev(aux, bt), impl=GEMM_FAST)
    assert_close("dgelu", Dg, acc * gelu_grad(aux), tol)
    # the pair with the derivative kept instead of the pre-activation (aux_is_derivative)
    ops.gemm(A, B, H, epilogue=EPI_BIAS_GELU, bias=bias_d, C2=P, impl=GEMM_FAST, aux_deriv=True)
    assert_close("gelu[deriv]", H, F.gelu(acc + bias), tol)
    assert_close("gelu'[deriv]", P, gelu_grad(acc + bias), tol)
    Dd = torch.empty_like(Dg)
    ops.gemm(A, Bt, Dd, b_kmajor=False, epilogue=EPI_DGELU, aux=dev(aux, bt), impl=GEMM_FAST, aux_deriv=True)
    assert_close("dgelu[deriv]", Dd, acc * aux, tol)
    # fused bias gradient: column sums per 128-row group (256x256-tile path only)
    from vit_torch_amd import _lib
    if ops.gemm_uses_fast(M, N, K, b_kmajor=False, epilogue=EPI_DGELU, colsum_part=True):
        part = torch.full((M // 128, N), float("nan"), device="cuda")
        Dg2 = torch.empty_like(Dg)
        ops.gemm(A, Bt, Dg2, b_kmajor=False, epilogue=EPI_DGELU, aux=dev(aux, bt), impl=GEMM_FAST, colsum_part=part)
        assert torch.equal(Dg2, Dg)
        assert_close("dgelu.colsum_part", part, (acc * gelu_grad(aux)).view(M // 128, 128, N).sum(1), 2e-3)
        out = torch.empty(N, device="cuda")
        ops.colsum(part, out)
        assert_close("dgelu.colsum", out, (acc * gelu_grad(aux)).sum(0), 2e-3)
    else:
        with pytest.raises(_lib.VitmiError):
            ops.gemm(A, Bt, Dg, b_kmajor=False, epilogue=EPI_DGELU, aux=dev(aux, bt), impl=GEMM_FAST,
                     colsum_part=torch.empty((M // 128, N), device="cuda"))
    n_tok = 64
    pos, cls = gen((n_tok, N), 10), gen((N,), 11)
    t = torch.arange(M) % n_tok
    want = acc + bias + pos[t]
    want[t == 0] = cls + pos[0]
    for rdt in (torch.float32, bt):
        X = torch.empty((M, N), device="cuda", dtype=rdt)
        ops.gemm(A, B, X, epilogue=EPI_PATCH_POS, bias=bias_d, pos=dev(pos), n_tok=n_tok, cls=dev(cls), impl=GEMM_FAST)
        assert_close(f"patch_pos[{rdt}]", X, want, TOL[rdt] if rdt == bt else 1e-4)


def test_gemm_fast_matches_generic_bitwise_on_integers(ops, pipe):
    """Exact check of the fragment/tile index maps: small-integer operands make every
    product and partial sum exact in fp32, so fast and generic must agree bit for bit
    (asymmetric data, guide §3 'A=I-check with ASYMMETRIC B')."""
    from vit_torch_amd._lib import GEMM_FAST, GEMM_GENERIC
    M, N, K = 512, 512, 256
    g = torch.Generator("cpu").manual_seed(5)
    a = torch.randint(-4, 5, (M, K), generator=g).float()
    b = torch.randint(-4, 5, (N, K), generator=g).float()
    want = a @ b.t()
    for akm, bkm in [(True, True), (True, False), (False, False)]:
        A = dev(a if akm else a.t().contiguous(), torch.bfloat16)
        B = dev(b if bkm else b.t().contiguous(), torch.bfloat16)
        Cf = torch.empty((M, N), device="cuda")
        Cg = torch.empty((M, N), device="cuda")
        ops.gemm(A, B, Cf, a_kmajor=akm, b_kmajor=bkm, impl=GEMM_FAST)
        ops.gemm(A, B, Cg, a_kmajor=akm, b_kmajor=bkm, impl=GEMM_GENERIC)
        assert torch.equal(Cf.cpu(), want), f"fast kernel wrong for layout a_km={akm} b_km={bkm}"
        assert torch.equal(Cg.cpu(), want)


@pytest.mark.parametrize("M,N,K", [(512, 256, 640), (768, 768, 768), (512, 512, 3072), (256, 256, 1024)])
def test_gemm_residual_fold_through_lds(ops, lib, M, N, K):
    """EPI_RESIDUAL, fp32 stream, no LayerScale / DropPath (every ViT block): the 256x256 kernel
    streams R through LDS during the main loop and adds it into the accumulators
    (gemm_fast.hip ResFold) instead of reading it in the epilogue.  Must equal the epilogue form
    (debug switch) to fp32 rounding and the torch reference; LayerScale / DropPath / a second
    output keep the epilogue form."""
    import ctypes
    from vit_torch_amd import _lib as L
    from vit_torch_amd._lib import EPI_RESIDUAL, GEMM_FAST
    raw = ctypes.CDLL(str(L.LIB_PATH))
    bt = torch.bfloat16
    a, b = bf16_round(gen((M, K), 31)), bf16_round(gen((N, K), 32, 0.2))
    r, bias, gam = gen((M, N), 33, 3.0), gen((N,), 34), gen((N,), 35)
    A, B, R, bias_d = dev(a, bt), dev(b, bt), dev(r), dev(bias)
    acc = a.double() @ b.double().t()
    try:
        outs = {}
        for mode in (1, 0):
            raw.vitmi_debug_gemm_rfold(mode)
            X = torch.full((M, N), float("nan"), device="cuda")
            ops.gemm(A, B, X, epilogue=EPI_RESIDUAL, bias=bias_d, R=R, impl=GEMM_FAST)
            outs[mode] = X.cpu()
            assert_close(f"residual[rfold={mode}]", X, (r.double() + acc + bias.double()).float(), 2e-6)
            X2 = torch.full((M, N), float("nan"), device="cuda")
            ops.gemm(A, B, X2, epilogue=EPI_RESIDUAL, R=R, impl=GEMM_FAST)                 # no bias
            assert_close(f"residual-nobias[rfold={mode}]", X2, (r.double() + acc).float(), 2e-6)
        assert_close("fold vs epilogue form", outs[1], outs[0], 1e-6)
        raw.vitmi_debug_gemm_rfold(1)
        X = torch.empty((M, N), device="cuda")
        ops.gemm(A, B, X, epilogue=EPI_RESIDUAL, bias=bias_d, R=R, gamma=dev(gam), impl=GEMM_FAST)
        assert_close("residual+gamma (epilogue form)", X, (r.double() + gam.double() * (acc + bias.double())).float(), 2e-6)
        # in place (C aliases R): every strip is fetched long before its tile is stored
        Xi = R.clone()
        ops.gemm(A, B, Xi, epilogue=EPI_RESIDUAL, bias=bias_d, R=Xi, impl=GEMM_FAST)
        assert torch.equal(Xi.cpu(), outs[1])
    finally:
        raw.vitmi_debug_gemm_rfold(-1)


@pytest.mark.parametrize("layout,epi", [("nt", "store"), ("nt", "gelu"), ("nt", "res"), ("nn", "dgelu"), ("nn", "store")])
def test_gemm_persistent_walk_covers_many_tiles_per_workgroup(ops, lib, layout, epi):
    """More tiles than CUs (3 x 256 + a ragged remainder): every persistent workgroup walks several
    tiles, prefetching the next tile's stages before its epilogue.  Must equal the one-tile-per-
    workgroup launch bit for bit (same arithmetic, same order) and the torch reference."""
    import ctypes
    from vit_torch_amd import _lib as L
    from vit_torch_amd._lib import EPI_BIAS_GELU, EPI_DGELU, EPI_RESIDUAL, EPI_STORE, GEMM_FAST
    raw = ctypes.CDLL(str(L.LIB_PATH))
    M, N, K = 256 * 67, 256 * 13, 768                    # 871 tiles = 3.4 rounds on 256 CUs
    bt = torch.bfloat16
    akm, bkm = (True, True) if layout == "nt" else (True, False)
    g = torch.Generator("cpu").manual_seed(77)
    a = bf16_round(torch.randn(M, K, generator=g))
    b = bf16_round(torch.randn(N, K, generator=g) * 0.05)
    A = dev(a, bt)
    B = dev(b if bkm else b.t().contiguous(), bt)
    bias = dev(torch.randn(N, generator=g))
    Rres = dev(torch.randn(M, N, generator=g)) if epi == "res" else None
    acc = a @ b.t()
    outs = []
    for persist in (1, 0):
        raw.vitmi_debug_gemm_persist(persist)
        try:
            if epi == "store":
                C = torch.full((M, N), float("nan"), device="cuda").to(bt)
                ops.gemm(A, B, C, a_kmajor=akm, b_kmajor=bkm, bias=bias if layout == "nt" else None, impl=GEMM_FAST)
                want = acc + (bias.cpu() if layout == "nt" else 0)
                res = (C,)
            elif epi == "gelu":
                C, P = (torch.full((M, N), float("nan"), device="cuda").to(bt) for _ in range(2))
                ops.gemm(A, B, C, epilogue=EPI_BIAS_GELU, bias=bias, C2=P, impl=GEMM_FAST)
                want = F.gelu(bf16_round(acc + bias.cpu()))
                res = (C, P)
            elif epi == "res":
                C = torch.full((M, N), float("nan"), device="cuda")
                ops.gemm(A, B, C, epilogue=EPI_RESIDUAL, bias=bias, R=Rres, impl=GEMM_FAST)
                want = Rres.cpu() + acc + bias.cpu()
                res = (C,)
            else:
                aux = dev(bf16_round(torch.randn(M, N, generator=torch.Generator("cpu").manual_seed(79))), bt)
                C = torch.full((M, N), float("nan"), device="cuda").to(bt)
                ops.gemm(A, B, C, a_kmajor=akm, b_kmajor=bkm, epilogue=EPI_DGELU, aux=aux, impl=GEMM_FAST)
                want = acc * gelu_grad(aux.float().cpu())
                res = (C,)
        finally:
            raw.vitmi_debug_gemm_persist(1)
        assert_close(f"{layout}/{epi} persist={persist}", res[0], want, 1e-4 if res[0].dtype == torch.float32 else TOL[bt])
        outs.append([t.float().cpu() for t in res])
    for x, y in zip(outs[0], outs[1]):
        assert torch.equal(x, y), "persistent and one-tile-per-workgroup launches must agree bit for bit"


@pytest.mark.parametrize("layout", ["nt", "nn"])
def test_gemm_tile_order_and_store_policy_do_not_change_results(ops, lib, layout):
    """Column-band tile orders (vitmi_debug_gemm_band: row-major, bands of 2, ragged bands of 5 over 13
    column tiles, with and without the split tail) and the cache policies of the output stores
    (plain / sc1 / nt) only change WHERE and HOW a tile is written: every combination must give the
    same bits, and cover every tile (the output starts as NaN)."""
    import ctypes
    from vit_torch_amd import _lib as L
    from vit_torch_amd._lib import EPI_BIAS_GELU, GEMM_FAST
    raw = ctypes.CDLL(str(L.LIB_PATH))
    M, N, K = 256 * 70, 256 * 13, 256
    bt = torch.bfloat16
    g = torch.Generator("cpu").manual_seed(91)
    a = bf16_round(torch.randn(M, K, generator=g))
    b = bf16_round(torch.randn(N, K, generator=g) * 0.05)
    A = dev(a, bt)
    bkm = layout == "nt"
    B = dev(b if bkm else b.t().contiguous(), bt)
    bias = dev(torch.randn(N, generator=g))
    ref = None
    try:
        for band, pol in [(0, 0), (2, 0), (5, 0), (5, 1), (5, 2), (-1, -1)]:
            raw.vitmi_debug_gemm_band(band)
            raw.vitmi_debug_gemm_store_policy(pol)
            C = torch.full((M, N), float("nan"), device="cuda").to(bt)
            if layout == "nt":
                P = torch.full((M, N), float("nan"), device="cuda").to(bt)
                ops.gemm(A, B, C, epilogue=EPI_BIAS_GELU, bias=bias, C2=P, impl=GEMM_FAST)
                out = (C.float().cpu(), P.float().cpu())
            else:
                ops.gemm(A, B, C, b_kmajor=False, impl=GEMM_FAST)
                out = (C.float().cpu(),)
            assert all(torch.isfinite(t).all() for t in out), (band, pol)
            if ref is None:
                ref = out
                want = a @ b.t()
                assert_close("band0", out[0], F.gelu(bf16_round(want + bias.cpu())) if layout == "nt" else want, TOL[bt])
            else:
                for x, y in zip(out, ref):
                    assert torch.equal(x, y), (band, pol)
    finally:
        raw.vitmi_debug_gemm_band(-1)
        raw.vitmi_debug_gemm_store_policy(-1)


def test_attention_bwd_persistent_pair_walk(ops, lib):
    """More (image, head) pairs than CUs: the fused backward's workgroups walk several pairs each,
    loading the next pair while the current one is written out (276 pairs on 256 CUs: 20 workgroups
    take two).  Must equal the one-pair-per-workgroup launch bit for bit, bias partials included,
    and autograd within the bf16 tolerance."""
    import ctypes
    from vit_torch_amd import _lib as L
    raw = ctypes.CDLL(str(L.LIB_PATH))
    B, N, H, hd = 23, 197, 12, 64
    bt = torch.bfloat16
    scale = hd ** -0.5
    qkv = bf16_round(gen((B, N, 3 * H * hd), 21))
    do = bf16_round(gen((B, N, H * hd), 22))
    QKV, DO = dev(qkv, bt), dev(do, bt)
    O = torch.empty((B, N, H * hd), device="cuda", dtype=bt)
    lse = torch.empty(B * H * N, device="cuda")
    ops.attn_fwd(QKV, O, lse, B, N, H, hd, scale)
    outs = []
    for flags in (0, L.LAUNCH_SHARED_DEVICE):          # per-call launch form (ABI 105)
        dqkv = torch.full((B, N, 3 * H * hd), float("nan"), device="cuda").to(bt)
        part = torch.full((ops.attn_bwd_dbias_rows(B, N), 3 * H * hd), float("nan"), device="cuda")
        ops.attn_bwd(QKV, O, DO, lse, dqkv, B, N, H, hd, scale, dbias_part=part, launch_flags=flags)
        outs.append((dqkv.float().cpu(), part.cpu()))
        assert torch.isfinite(outs[-1][0]).all() and torch.isfinite(outs[-1][1]).all()
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])
    qr = qkv.clone().requires_grad_(True)
    o_ref, _ = attn_ref(qr, B, N, H, hd, scale)
    o_ref.backward(do)
    assert_close("dqkv", outs[0][0], qr.grad, 2.5e-2)
    assert_close("dbias", outs[0][1].sum(0), qr.grad.view(B * N, -1).sum(0), 2.5e-2)


def test_gemm_pair_shares_one_split_k_launch(ops, lib):
    """vitmi_gemm_pair: the proj (768 x 768) and qkv (2304 x 768) weight gradients of an attention block in one grid
    (36 tiles x 7 k-slices instead of 9 x 28 + 27 x 9).  Must agree with the fp64 reference and with the two
    separate launches (summation order over k-slices differs: fp32 rounding only); unpairable inputs fall back."""
    K, D = 197 * 64, 768                       # 12 608 tokens: a multiple of 64, ragged against the slice length
    g = torch.Generator("cpu").manual_seed(91)
    bt = torch.bfloat16
    dy0, x0 = bf16_round(torch.randn(K, D, generator=g)), bf16_round(torch.randn(K, D, generator=g))
    dy1, x1 = bf16_round(torch.randn(K, 3 * D, generator=g)), bf16_round(torch.randn(K, D, generator=g))
    DY0, X0, DY1, X1 = dev(dy0, bt), dev(x0, bt), dev(dy1, bt), dev(x1, bt)
    assert ops.gemm_pair_shares_a_launch(D, D, 3 * D, D, K)
    W0 = torch.full((D, D), float("nan"), device="cuda")
    W1 = torch.full((3 * D, D), float("nan"), device="cuda")
    ops.gemm_pair(DY0, X0, W0, DY1, X1, W1)
    S0, S1 = torch.empty_like(W0), torch.empty_like(W1)
    ops.gemm(DY0, X0, S0, a_kmajor=False, b_kmajor=False)
    ops.gemm(DY1, X1, S1, a_kmajor=False, b_kmajor=False)
    want0, want1 = (dy0.double().t() @ x0.double()).float(), (dy1.double().t() @ x1.double()).float()
    assert_close("pair dW0", W0, want0, 2e-5)
    assert_close("pair dW1", W1, want1, 2e-5)
    assert_close("pair vs separate dW0", W0, S0.cpu(), 2e-5)
    assert_close("pair vs separate dW1", W1, S1.cpu(), 2e-5)
    # not pairable (ragged N): falls back to two launches with the same results
    dy2 = bf16_round(torch.randn(K, 200, generator=g))
    assert not ops.gemm_pair_shares_a_launch(D, D, 200, D, K)
    W2 = torch.full((200, D), float("nan"), device="cuda")
    W0b = torch.full((D, D), float("nan"), device="cuda")
    ops.gemm_pair(DY0, X0, W0b, dev(dy2, bt), X1, W2)
    assert torch.equal(W0b, S0)
    assert_close("fallback dW2", W2, (dy2.double().t() @ x1.double()).float(), 2e-5)


@pytest.mark.parametrize("layout,epi,cdt", [("nt", "store", torch.bfloat16), ("nt", "gelu", torch.bfloat16), ("nt", "res", torch.bfloat16),
                                            ("nt", "res", torch.float32), ("nn", "store", torch.bfloat16), ("nn", "dgelu", torch.bfloat16)])
def test_gemm_counted_prefetch_wait_equals_strict_wait(ops, lib, layout, epi, cdt):
    """the first wait of a prefetched tile counts E_MIN epilogue stores as younger operations (gemm_fast.hip,
    the INVARIANT next to E_MIN).  With vitmi_debug_gemm_strict_wait(1) that wait is vmcnt(0): both forms must give the
    same bits on every epilogue, on a launch where workgroups walk two tiles each (320 tiles on 256 CUs)."""
    import ctypes
    from vit_torch_amd import _lib as L
    from vit_torch_amd._lib import EPI_BIAS_GELU, EPI_DGELU, EPI_RESIDUAL, GEMM_FAST
    raw = ctypes.CDLL(str(L.LIB_PATH))
    M, N, K = 256 * 40, 256 * 8, 768
    akm, bkm = True, layout == "nt"
    g = torch.Generator("cpu").manual_seed(93)
    bt = torch.bfloat16
    A = dev(bf16_round(torch.randn(M, K, generator=g)), bt)
    B = dev(bf16_round(torch.randn((N, K) if bkm else (K, N), generator=g) * 0.05), bt)
    bias = dev(torch.randn(N, generator=g))
    kw = dict(a_kmajor=akm, b_kmajor=bkm)
    if epi == "gelu":
        kw.update(epilogue=EPI_BIAS_GELU, bias=bias, aux_deriv=True)
    elif epi == "res":
        kw.update(epilogue=EPI_RESIDUAL, bias=bias, R=dev(torch.randn(M, N, generator=g), cdt))
    elif epi == "dgelu":
        kw.update(epilogue=EPI_DGELU, aux=dev(bf16_round(torch.randn(M, N, generator=g)), bt), aux_deriv=True)
    outs = []
    try:
        for strict in (0, 1, 0):
            raw.vitmi_debug_gemm_strict_wait(strict)
            C = torch.full((M, N), float("nan"), device="cuda").to(cdt)
            extra = dict(C2=torch.full((M, N), float("nan"), device="cuda").to(cdt)) if epi == "gelu" else {}
            ops.gemm(A, B, C, impl=GEMM_FAST, **kw, **extra)
            outs.append([C.float().cpu()] + [t.float().cpu() for t in extra.values()])
            assert torch.isfinite(outs[-1][0]).all()
    finally:
        raw.vitmi_debug_gemm_strict_wait(0)
    for a, b, c in zip(*outs):
        assert torch.equal(a, b) and torch.equal(a, c), "counted and strict waits of a prefetched tile disagree"


@pytest.mark.parametrize("layout,epi", [("nt", "store"), ("nn", "store"), ("tn", "store"), ("nt", "gelu"), ("nt", "res"), ("nn", "dgelu")])
@pytest.mark.parametrize("nt_mb", [1, 1 << 20])
def test_gemm_straight_line_epilogues_equal_the_general_row_function(ops, lib, layout, epi, nt_mb):
    """Round 3: the bf16 epilogues of the step exist as straight-line code (plain store / gelu + gelu' / gelu' multiply with
    `nt` stores, the plain residual) with THREE strips of the side input in flight; vitmi_debug_gemm_side_depth(1) selects
    the general row function with one strip ahead.  Same arithmetic: the outputs must agree bit for bit (gelu' to one bf16
    ulp on a vanishing fraction: its formula is a chain of explicit fmas, but hipcc is free to keep an intermediate in a
    different form), with the `nt` policy forced on (1 MB threshold) and off, on a launch where workgroups walk several tiles."""
    import ctypes
    from vit_torch_amd import _lib as L
    from vit_torch_amd._lib import EPI_BIAS_GELU, EPI_DGELU, EPI_RESIDUAL, GEMM_FAST
    raw = ctypes.CDLL(str(L.LIB_PATH))
    M, N, K = 256 * 40, 256 * 8, 768
    if layout == "tn":
        M, N, K = 768, 2048, 256 * 40
    akm, bkm = layout != "tn", layout == "nt"
    g = torch.Generator("cpu").manual_seed(94)
    bt = torch.bfloat16
    A = dev(bf16_round(torch.randn((M, K) if akm else (K, M), generator=g)), bt)
    B = dev(bf16_round(torch.randn((N, K) if bkm else (K, N), generator=g) * 0.05), bt)
    bias = dev(torch.randn(N, generator=g))
    kw = dict(a_kmajor=akm, b_kmajor=bkm)
    if epi == "store" and layout == "nt":
        kw.update(bias=bias)
    elif epi == "gelu":
        kw.update(epilogue=EPI_BIAS_GELU, bias=bias, aux_deriv=True)
    elif epi == "res":
        kw.update(epilogue=EPI_RESIDUAL, bias=bias, R=dev(bf16_round(torch.randn(M, N, generator=g)), bt))
    elif epi == "dgelu":
        kw.update(epilogue=EPI_DGELU, aux=dev(bf16_round(torch.randn(M, N, generator=g)), bt), aux_deriv=True,
                  colsum_part=torch.full((M // 128, N), float("nan"), device="cuda"))
    outs = []
    try:
        raw.vitmi_debug_gemm_nt_min_mb(nt_mb)
        for depth in (1, 3, 1):
            raw.vitmi_debug_gemm_side_depth(depth)
            C = torch.full((M, N), float("nan"), device="cuda").to(bt)
            extra = dict(C2=torch.full((M, N), float("nan"), device="cuda").to(bt)) if epi == "gelu" else {}
            ops.gemm(A, B, C, impl=GEMM_FAST, **kw, **extra)
            outs.append([C.float().cpu()] + [t.float().cpu() for t in extra.values()] +
                        ([kw["colsum_part"].clone().cpu()] if epi == "dgelu" else []))
            assert all(torch.isfinite(t).all() for t in outs[-1])
    finally:
        raw.vitmi_debug_gemm_side_depth(3)
        raw.vitmi_debug_gemm_nt_min_mb(64)
    for i, (a, b, c) in enumerate(zip(*outs)):
        assert torch.equal(a, c)
        if epi == "gelu" and i == 1:
            bad = a != b
            assert bad.float().mean().item() < 1e-4 and (a - b).abs().max().item() <= 2.0 ** -6, "gelu' differs by more than a rounding"
        else:
            assert torch.equal(a, b), f"output {i} of the two epilogue forms differs"


@pytest.mark.parametrize("layout,epi,cdt", [("nt", "res", torch.bfloat16), ("nt", "res", torch.float32), ("nn", "store", torch.bfloat16),
                                            ("nt", "store", torch.bfloat16)])
def test_gemm_tail_fixup_by_the_last_arriving_slice(ops, lib, layout, epi, cdt):
    """Split tail (vitmi_debug_gemm_tail(1)): the remainder tiles' k-slices store raw partial tiles and the slice that
    arrives LAST applies the epilogue (gemm_fast.hip FIX) instead of a finisher kernel.  Same
    summation order as the finisher: the two forms must agree bit for bit and with the reference, on a launch whose
    remainder is spread over all XCDs, repeated so that the arrival order varies."""
    import ctypes
    from vit_torch_amd import _lib as L
    from vit_torch_amd._lib import EPI_RESIDUAL, GEMM_FAST
    raw = ctypes.CDLL(str(L.LIB_PATH))
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    tiles_m = cus // 3 + 21                   # 3 column tiles: (cus + 63) tiles = one full round + a remainder of 63 <= cus / 3
    M, N, K = 256 * tiles_m, 768, 2304
    akm, bkm = True, layout == "nt"
    g = torch.Generator("cpu").manual_seed(97)
    bt = torch.bfloat16
    a = bf16_round(torch.randn(M, K, generator=g))
    b = bf16_round(torch.randn(N, K, generator=g) * 0.05)
    A, B = dev(a, bt), dev(b if bkm else b.t().contiguous(), bt)
    bias = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    r = bf16_round(r) if cdt == bt else r
    kw = dict(a_kmajor=akm, b_kmajor=bkm)
    want = a @ b.t()
    if epi == "res":
        kw.update(epilogue=EPI_RESIDUAL, bias=dev(bias), R=dev(r, cdt))
        want = r + want + bias
    outs = []
    try:
        raw.vitmi_debug_gemm_tail(1)
        for fix in (1, 0, 1, 1):
            raw.vitmi_debug_gemm_tail_fixup(fix)
            C = torch.full((M, N), float("nan"), device="cuda").to(cdt)
            ops.gemm(A, B, C, impl=GEMM_FAST, **kw)
            outs.append(C.float().cpu())
            assert torch.isfinite(outs[-1]).all(), f"fixup={fix}: some tile never received its epilogue"
    finally:
        raw.vitmi_debug_gemm_tail(-1)
        raw.vitmi_debug_gemm_tail_fixup(1)
    assert_close("tail fix-up", outs[0], want, 1e-4 if cdt == torch.float32 else TOL[bt])
    for o in outs[1:]:
        assert torch.equal(outs[0], o), "fix-up by the last slice and the finisher kernel must agree bit for bit"


# ------------------------------------------------------- deferred folds ---
def test_fold_many_equals_the_single_folds_bitwise(ops):
    """vitmi_fold_many: 40 queued folds of mixed shapes (more than one 32-entry launch), one to three segments each,
    against the one-launch-per-fold calls: every output bit-identical."""
    q = ops.FoldQueue()
    want, got = [], []
    for i in range(40):
        S, N, nseg = [7, 64, 394, 1500][i % 4], [10, 96, 768, 2304, 33][i % 5], 1 + i % 3
        part = dev(gen((S, nseg * N + (8 if i % 2 else 0)), 100 + i))
        ld = part.shape[1]
        ref = [torch.empty(N, device="cuda") for _ in range(nseg)]
        for k in range(nseg):
            ops.colsum(part[:, k * N:], ref[k], M=S, N=N, ld=ld)
        outs = [torch.full((N,), float("nan"), device="cuda") for _ in range(nseg)]
        q.add(part, S, N, ld, outs)
        want.append(ref)
        got.append(outs)
    assert len(q) == 40
    q.flush()
    assert len(q) == 0
    torch.cuda.synchronize()
    for ref, outs in zip(want, got):
        for r, o in zip(ref, outs):
            assert torch.equal(r, o)
        assert_close("fold", outs[0], ref[0], 0.0)


@pytest.mark.parametrize("M,D,T,R", [(394, 768, torch.bfloat16, torch.bfloat16), (1030, 96, torch.bfloat16, torch.float32),
                                     (77, 384, torch.float32, torch.float32)])
def test_layernorm_bwd_deferred_fold_equals_the_immediate_one(ops, M, D, T, R):
    x, dy, gin = dev(gen((M, D), 1) * 1.5 + 0.3, R), dev(gen((M, D), 2), T), dev(gen((M, D), 3), R)
    g = dev(1 + 0.1 * gen((D,), 4))
    mean, rstd = x.float().mean(-1), (x.float().var(-1, unbiased=False) + 1e-6).rsqrt()
    res = []
    for q in (None, ops.FoldQueue()):
        G = gin.clone()
        Gb = torch.empty((M, D), device="cuda", dtype=T) if T != R else None
        dg, db, gs = (torch.full((D,), float("nan"), device="cuda") for _ in range(3))
        ops.layernorm_bwd(dy, x, mean, rstd, g, G, G, Gb, dg, db, gsum=gs, fold=q)
        if q is not None:
            ops.workspace(1 << 20, x.device).fill_(0x7f)      # the deferred partials do not live in the shared workspace
            q.flush()
        res.append((G, dg, db, gs))
    torch.cuda.synchronize()
    for a, b in zip(*res):
        assert torch.equal(a, b)


# ----------------------------------------------------- skinny fp32 GEMMs ---
@pytest.mark.parametrize("akm,bkm,M,N,K", [(True, True, 256, 10, 768), (True, True, 37, 16, 100), (True, True, 5, 1, 64),
                                           (True, False, 256, 768, 10), (True, False, 33, 70, 32),
                                           (False, False, 10, 768, 256), (False, False, 32, 45, 1000)])
def test_gemm_skinny_fp32_forms(ops, akm, bkm, M, N, K):
    """The classifier head's fp32 products take the plain-FMA kernels (gemm_skinny_kernel) by default: against fp32 torch
    math and against the 64x64-tile kernel (impl = GENERIC), with bias / alpha / accumulate through the same epilogue."""
    from vit_torch_amd._lib import GEMM_GENERIC
    a, b, bias = gen((M, K), 1), gen((N, K), 2), gen((N,), 3)
    A = dev(a if akm else a.t().contiguous())
    B = dev(b if bkm else b.t().contiguous())
    C, Cg = torch.empty((M, N), device="cuda"), torch.empty((M, N), device="cuda")
    ops.gemm(A, B, C, a_kmajor=akm, b_kmajor=bkm, bias=dev(bias), alpha=0.5)
    ops.gemm(A, B, Cg, a_kmajor=akm, b_kmajor=bkm, bias=dev(bias), alpha=0.5, impl=GEMM_GENERIC)
    assert_close("skinny", C, 0.5 * (a @ b.t()) + bias, 2e-5)
    assert_close("skinny vs generic", C, Cg.cpu(), 2e-5)
    ops.gemm(A, B, C, a_kmajor=akm, b_kmajor=bkm, accumulate=True)
    assert_close("skinny accumulate", C, 1.5 * (a @ b.t()) + bias, 2e-5)


def test_gemm_skinny_gelu_pair(ops):
    """Head with a hidden GELU layer: EPI_BIAS_GELU (second output = pre-activation) and EPI_DGELU through the skinny forms."""
    from vit_torch_amd._lib import EPI_BIAS_GELU, EPI_DGELU
    M, N, K = 64, 12, 256
    a, w, bias = gen((M, K), 1), gen((N, K), 2) * 0.1, gen((N,), 3)
    pre = a @ w.t() + bias
    C, P = torch.empty((M, N), device="cuda"), torch.empty((M, N), device="cuda")
    ops.gemm(dev(a), dev(w), C, epilogue=EPI_BIAS_GELU, bias=dev(bias), C2=P)
    assert_close("gelu", C, F.gelu(pre), 2e-5)
    assert_close("pre", P, pre, 2e-5)
    d, w2 = gen((M, 7), 4), gen((7, N), 5)          # dx = (d @ w2) * gelu'(pre): K = 7 -> FORM 1
    dx = torch.empty((M, N), device="cuda")
    ops.gemm(dev(d), dev(w2), dx, b_kmajor=False, epilogue=EPI_DGELU, aux=P)
    assert_close("dgelu", dx, (d @ w2) * gelu_grad(pre), 2e-5)


# ------------------------------------------------------------- ragged M on the 256x256 tile kernel ---
@pytest.mark.parametrize("layout", ["nt", "nn"])
@pytest.mark.parametrize("epi", ["store", "gelu", "res", "dgelu"])
@pytest.mark.parametrize("M,N,K", [(300, 256, 64), (640, 768, 192), (1000, 512, 768), (18560, 768, 768)])
def test_gemm_ragged_m_runs_on_the_tile_kernel_with_padded_rows(ops, layout, epi, M, N, K):
    """VITMI_LAUNCH_ROWS_PADDED (round 4): C / C2 / R / AUX allocated to the next multiple of 256 rows; a k-major A with
    M % 256 != 0 is staged with its last row repeated and the last row tile stores into the padding.  Rows < M must equal
    the fp32 reference (and the 256x128 ragged kernel's result to bf16 rounding), the column sums must leave the padding
    out, and nothing beyond the padding may be touched."""
    from vit_torch_amd._lib import EPI_BIAS_GELU, EPI_DGELU, EPI_RESIDUAL, EPI_STORE, LAUNCH_ROWS_PADDED
    if (epi in ("gelu", "res") and layout != "nt") or (epi == "dgelu" and layout != "nn"):
        pytest.skip("epilogue / layout combination the library does not build")
    akm, bkm = {"nt": (True, True), "nn": (True, False)}[layout]
    bt = torch.bfloat16
    Mp = (M + 255) // 256 * 256
    a, b = bf16_round(gen((M, K), 1)), bf16_round(gen((N, K), 2) * 0.2)
    A = dev(a).to(bt)
    Bm = dev(b if bkm else b.t().contiguous()).to(bt)
    guard = 64

    def padded(fill):
        buf = torch.full((Mp + guard, N), fill, device="cuda", dtype=bt)
        return buf, buf[:M]

    cbuf, C = padded(float("nan"))
    want = a @ b.t()
    kw = dict(a_kmajor=akm, b_kmajor=bkm, launch_flags=LAUNCH_ROWS_PADDED)
    extra = None
    bias = gen((N,), 3)
    side = bf16_round(gen((M, N), 4))
    if epi == "store":
        kw.update(bias=dev(bias))
        want = want + bias
    elif epi == "gelu":
        c2buf, C2 = padded(float("nan"))
        kw.update(epilogue=EPI_BIAS_GELU, bias=dev(bias), C2=C2, aux_deriv=True)
        pre = want + bias
        want = torch.nn.functional.gelu(pre)
        extra = (C2, 0.5 * (1 + torch.erf(pre / 2 ** 0.5)) + pre * torch.exp(-0.5 * pre * pre) / (2 * torch.pi) ** 0.5)
    elif epi == "res":
        rbuf, R = padded(0.5)
        R.copy_(dev(side).to(bt))
        kw.update(epilogue=EPI_RESIDUAL, bias=dev(bias), R=R)
        want = side + want + bias
    else:
        abuf, AUX = padded(0.25)
        AUX.copy_(dev(side).to(bt))
        pbuf = torch.full(((M + 127) // 128 + 2, N), float("nan"), device="cuda")       # two guard rows behind the partial rows
        part = pbuf[:(M + 127) // 128]
        kw.update(epilogue=EPI_DGELU, aux=AUX, aux_deriv=True, colsum_part=part)
        want = want * side
    assert ops.gemm_uses_fast(M, N, K, b_kmajor=bkm) or True
    ops.gemm(A, Bm, C, **kw)
    torch.cuda.synchronize()
    assert_close(f"ragged {layout} {epi}", C, want, 1.2e-2)
    assert torch.isnan(cbuf[Mp:].float()).all(), "rows beyond the padding were written"
    # proof that the 256x256 tile kernel ran: it stores whole tiles, so the padding rows hold numbers; the 256x128 ragged
    # kernel masks its stores and would have left the NaN fill
    assert not torch.isnan(cbuf[M:Mp].float()).any(), "the padding rows were not written: the call did not take the tile kernel"
    if extra is not None:
        assert_close("gelu' (C2)", extra[0], extra[1], 1.2e-2)
    if epi == "dgelu":
        assert_close("column sums without the padding rows", part.sum(0).cpu(), C.float().cpu().sum(0), 2e-2)
        assert torch.isnan(pbuf[(M + 127) // 128:]).all(), "a partial row beyond ceil(M / 128) was written (the last tile's second half lies in the padding)"
    # the same product without the flag (256x128 ragged kernel, unpadded buffers): equal to bf16 rounding
    C0 = torch.empty((M, N), device="cuda", dtype=bt)
    kw0 = {k: v for k, v in kw.items() if k not in ("launch_flags", "colsum_part")}
    for key in ("C2", "R", "aux"):
        if key in kw0:
            kw0[key] = kw0[key].contiguous().clone() if key != "C2" else torch.empty((M, N), device="cuda", dtype=bt)
    ops.gemm(A, Bm, C0, **kw0)
    assert_close("tile kernel vs ragged kernel", C, C0.float().cpu(), 8e-3)


def test_gemm_ragged_m_fp32_output_never_splits_k(ops):
    """an fp32-C plain-store launch with VITMI_LAUNCH_ROWS_PADDED whose tile count invites split-K
    (nn, M = 6304 = 24.6 row tiles x 3 column tiles = 75 tiles <= 128, K = 1536 = 24 k-steps) used to stride the k-slices by
    M rows while the ragged last row tile stored whole 256-row tiles: the padding rows of slice s landed on the first rows of
    slice s + 1 and the last slice wrote beyond the workspace.  Ragged M now never splits K: the first rows must equal the
    fp32 reference and nothing behind the padded C may be touched."""
    from vit_torch_amd._lib import LAUNCH_ROWS_PADDED
    M, N, K = 6304, 768, 1536
    bt = torch.bfloat16
    Mp = (M + 255) // 256 * 256
    a, b = bf16_round(gen((M, K), 11)), bf16_round(gen((N, K), 12) * 0.2)
    A = dev(a).to(bt)
    Bt = dev(b.t().contiguous()).to(bt)
    guard = 64
    cbuf = torch.full((Mp + guard, N), float("nan"), device="cuda")
    C = cbuf[:M]
    ops.gemm(A, Bt, C, b_kmajor=False, launch_flags=LAUNCH_ROWS_PADDED)
    torch.cuda.synchronize()
    assert_close("ragged nn fp32", C, a @ b.t(), 1e-4)
    assert torch.isnan(cbuf[Mp:]).all(), "rows beyond the padding were written"


# ------------------------------------------------------------- "bf16x3": fp32 products as three bf16 products ---
@pytest.mark.parametrize("Cn", [64, 36])          # 16-byte stores (cols % 8 == 0) and the 8-byte form
@pytest.mark.parametrize("stacked", [False, True])
@pytest.mark.parametrize("b_pattern", [False, True])
def test_split3_images(ops, stacked, b_pattern, Cn):
    """vitmi_split3: hi = bf16(x), lo = bf16(x - hi); A pattern hi|lo|hi, B pattern hi|hi|lo, side by side (k-major) or
    stacked (k-minor).  Bit-exact against the same two roundings in torch; hi + lo recovers x to 2^-16."""
    R = 37
    x = gen((R, Cn), 21) * 3.0
    out = ops.split3(dev(x), b_pattern=b_pattern, stacked=stacked).float().cpu()
    hi = x.to(torch.bfloat16).float()
    lo = (x - hi).to(torch.bfloat16).float()
    parts = [hi, hi, lo] if b_pattern else [hi, lo, hi]
    want = torch.cat(parts, dim=0 if stacked else 1)
    assert torch.equal(out, want)
    assert ((hi + lo) - x).abs().max().item() <= 2.0 ** -16 * x.abs().max().item()


@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
@pytest.mark.parametrize("M,N,K", [(512, 256, 128), (300, 96, 160), (1024, 768, 768)])
def test_gemm_split3_is_fp32_grade(ops, layout, M, N, K):
    """Three bf16 products of the hi / lo halves against the fp64 product of the fp32 operands: 1e-5 of the result's
    scale (a plain bf16 product of the same operands is ~4e-3)."""
    akm, bkm = {"nt": (True, True), "nn": (True, False), "tn": (False, False)}[layout]
    if layout == "tn":
        M, K = M // 8 * 8, K // 32 * 32
    a, b = gen((M, K), 31), gen((N, K), 32) * 0.3
    want = (a.double() @ b.double().t()).float()
    A = dev(a if akm else a.t().contiguous())
    B = dev(b if bkm else b.t().contiguous())
    C = torch.empty((M, N), device="cuda")
    ops.gemm_split3(A, B, C, a_kmajor=akm, b_kmajor=bkm)
    assert_close(f"split3[{layout}]", C, want, 2e-5)
    Cb = torch.empty((M, N), device="cuda")
    ops.gemm(A.to(torch.bfloat16), B.to(torch.bfloat16), Cb, a_kmajor=akm, b_kmajor=bkm)
    assert rel_err(Cb, want) > 20 * rel_err(C, want)


def test_gemm_split3_epilogues(ops):
    from vit_torch_amd._lib import EPI_BIAS_GELU, EPI_DGELU, EPI_PATCH_POS, EPI_RESIDUAL
    M, N, K = 788, 256, 192          # 4 images x 197 tokens
    a, w, bias = gen((M, K), 41), gen((N, K), 42) * 0.2, gen((N,), 43)
    acc = (a.double() @ w.double().t()).float()
    A, W, Wt, bd = dev(a), dev(w), dev(w.t().contiguous()), dev(bias)
    H, P = torch.empty((M, N), device="cuda"), torch.empty((M, N), device="cuda")
    ops.gemm_split3(A, W, H, epilogue=EPI_BIAS_GELU, bias=bd, C2=P)
    assert_close("pre", P, acc + bias, 2e-5)
    assert_close("gelu", H, F.gelu(acc + bias), 2e-5)
    r = gen((M, N), 44)
    X = torch.empty((M, N), device="cuda")
    ops.gemm_split3(A, W, X, epilogue=EPI_RESIDUAL, bias=bd, R=dev(r))
    assert_close("residual", X, r + acc + bias, 2e-5)
    d, pre = gen((M, N), 45), gen((M, N), 46)
    dx = torch.empty((M, K), device="cuda")
    ops.gemm_split3(dev(d), W, dx, b_kmajor=False, epilogue=EPI_DGELU, aux=dev(gen((M, K), 47)))
    want = (d.double() @ w.double()).float() * gelu_grad(gen((M, K), 47))
    assert_close("dgelu", dx, want, 3e-5)
    pos, cls = gen((197, N), 48), gen((N,), 49)
    Xp = torch.empty((M, N), device="cuda")
    ops.gemm_split3(A, W, Xp, epilogue=EPI_PATCH_POS, bias=bd, pos=dev(pos), n_tok=197, cls=dev(cls))
    wantp = (acc + bias).view(4, 197, N) + pos
    wantp[:, 0] = cls + pos[0]
    assert_close("patch_pos", Xp, wantp.reshape(M, N), 2e-5)


@pytest.mark.parametrize("B,N,H,hd", [(3, 197, 12, 64), (2, 145, 3, 64), (4, 5, 2, 64), (2, 256, 2, 32), (1, 33, 3, 32)])
def test_attention_fp32_on_the_matrix_pipe_equals_the_vector_form(ops, lib, B, N, H, hd):
    """Round 5: fp32 attention on v_mfma_f32_32x32x2_f32 (hd in {32, 64}, N <= 256) against (i) an fp64 evaluation of the
    reference's formula (models/swin.py:124-142 minus bias / mask) and (ii) the VALU kernels it replaces
    (vitmi_debug_attn_f32_valu): forward, lse and all three gradients; padded keys / queries must not leak (N = 197, 145,
    5, 33 are not multiples of 32) and nothing outside the [B, N] rows may be written."""
    import ctypes
    from vit_torch_amd import _lib as L
    raw = ctypes.CDLL(str(L.LIB_PATH))
    scale = hd ** -0.5
    qkv = gen((B, N, 3 * H * hd), 51)
    do = gen((B, N, H * hd), 52)
    q64 = qkv.double().clone().requires_grad_(True)
    o_ref, lse_ref = attn_ref(q64, B, N, H, hd, scale)
    o_ref.backward(do.double())
    QKV, DO = dev(qkv), dev(do)
    res = {}
    for form in ("mfma", "valu"):
        raw.vitmi_debug_attn_f32_valu(1 if form == "valu" else 0)
        O = torch.full((B, N, H * hd), float("nan"), device="cuda")
        lse = torch.full((B * H * N,), float("nan"), device="cuda")
        dqkv = torch.full((B, N, 3 * H * hd), float("nan"), device="cuda")
        ops.attn_fwd(QKV, O, lse, B, N, H, hd, scale)
        ops.attn_bwd(QKV, O, DO, lse, dqkv, B, N, H, hd, scale)
        torch.cuda.synchronize()
        res[form] = (O.cpu(), lse.cpu(), dqkv.cpu())
    raw.vitmi_debug_attn_f32_valu(0)
    O, lse, dqkv = res["mfma"]
    assert_close("out", O, o_ref.detach().float(), 3e-6)
    assert_close("lse", lse.view(B, H, N), lse_ref.detach().float(), 2e-6)
    g = q64.grad.float().view(B, N, 3, H, hd)
    d = dqkv.view(B, N, 3, H, hd)
    for i, nm in enumerate("qkv"):
        assert_close(f"d{nm}", d[:, :, i], g[:, :, i], 1e-5)
    assert_close("out vs VALU form", O, res["valu"][0], 3e-6)
    assert_close("dqkv vs VALU form", dqkv, res["valu"][2], 1e-5)
