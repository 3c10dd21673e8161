"""CaiT (LayerScale + talking-heads trunk + class-attention layers) on libvitmi kernels.

Drop-in for /root/reference/models/cait.py `cait_models` (:155-253) and its registered
variants (:255-480): same constructor arguments, parameter names and shapes (state-dict
compatible: cls_token, pos_embed[1,Np,D], blocks.{i}.{gamma_1,gamma_2,norm1,attn.{qkv,proj,
proj_l,proj_w},norm2,mlp.{fc1,fc2}}, blocks_token_only.{i}.{...,attn.{q,k,v,proj},...}, norm,
head) and call contract.  `forward` runs `CaitEngine`: an explicit forward/backward kernel
sequence (the reference relies on autograd).

First version of this path: the talking-heads score tensors S / P / P' live
in HBM between batched MFMA products and the talking-heads softmax kernel; fusing them is
the next step for this row.
"""
from __future__ import annotations

from typing import Optional

import os

import torch
import torch.nn as nn

from . import ops
from ._lib import EPI_BIAS_GELU, EPI_DGELU, EPI_PATCH_POS, EPI_RESIDUAL, GEMM_AUTO, VitmiError
from .packing import ParamPack
from .vit import Mlp, PatchEmbed, _DT, _EngineFn, _head_layers, _trunc_normal_, dgelu_gemm_with_bias_grad, engine_gemm


class ClassAttention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.k = nn.Linear(dim, dim, bias=qkv_bias)
        self.v = nn.Linear(dim, dim, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class TalkingHeadAttention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.proj_l = nn.Linear(num_heads, num_heads)
        self.proj_w = nn.Linear(num_heads, num_heads)


class LayerScaleBlock(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, qkv_bias, eps, init_values):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = TalkingHeadAttention(dim, num_heads, qkv_bias)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.gamma_1 = nn.Parameter(init_values * torch.ones(dim))
        self.gamma_2 = nn.Parameter(init_values * torch.ones(dim))


class LayerScaleBlockCA(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, qkv_bias, eps, init_values):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = ClassAttention(dim, num_heads, qkv_bias)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.gamma_1 = nn.Parameter(init_values * torch.ones(dim))
        self.gamma_2 = nn.Parameter(init_values * torch.ones(dim))


def _ln_eps(norm_layer, default=1e-5):
    """eps of a `partial(nn.LayerNorm, eps=...)` (models/cait.py:258) or of nn.LayerNorm."""
    if norm_layer is None or norm_layer is nn.LayerNorm:
        return default
    kw = getattr(norm_layer, "keywords", None) or {}
    return kw.get("eps", default)


class cait_models(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0,
                 drop_path_rate=0.0, norm_layer=nn.LayerNorm, init_scale=1e-4, depth_token_only=2,
                 mlp_ratio_clstk=4.0, compute_dtype="bf16", residual_dtype="fp32", **_ignored):
        super().__init__()
        if drop_rate or attn_drop_rate or drop_path_rate:
            # the reference's factory always passes 0 (models/vision_all.py:189-191); cait.py's
            # DropPath branch would raise NameError there
            raise VitmiError("CaiT with dropout / drop-path is not supported (the reference never uses it)")
        if qk_scale is not None:
            raise VitmiError("qk_scale override is not supported")
        eps = _ln_eps(norm_layer)
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.compute_dtype = _DT[compute_dtype]
        self.split3 = compute_dtype == "bf16x3"        # vit.py: fp32 GEMMs as three bf16 products (ops.gemm_split3)
        # the residual stream between the blocks: "fp32" (default: the reference trains in fp32, and the fp32 stream ends a
        # 50-step run within 0.06-0.08 % of the oracle loss, tests/test_training_curve_gpu.py), "bf16" (what bench.py times:
        # 1.1-2.2 % on the same test), or "auto" = follow the compute dtype
        self.residual_dtype = self.compute_dtype if residual_dtype == "auto" else _DT[residual_dtype]
        self.apply_head = True            # cait_models.forward applies self.head (models/cait.py:248-253)
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        n = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n, embed_dim))
        self.blocks = nn.ModuleList([LayerScaleBlock(embed_dim, num_heads, mlp_ratio, qkv_bias, eps, init_scale)
                                     for _ in range(depth)])
        self.blocks_token_only = nn.ModuleList([
            LayerScaleBlockCA(embed_dim, num_heads, mlp_ratio_clstk, qkv_bias, eps, init_scale)
            for _ in range(depth_token_only)])
        self.norm = nn.LayerNorm(embed_dim, eps=eps)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        _trunc_normal_(self.pos_embed)
        _trunc_normal_(self.cls_token)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                _trunc_normal_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.LayerNorm):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
        self._engine: Optional[CaitEngine] = None

    def engine(self):
        if self._engine is None or not self._engine.is_current():
            self._engine = CaitEngine(self)
        return self._engine

    def forward(self, x):
        if not x.is_cuda:
            raise VitmiError("vit_torch_amd models run on an MI355X (HIP) device; got a CPU tensor "
                             "and there is no CPU fallback")
        eng = self.engine()
        if torch.is_grad_enabled() and any(p.requires_grad for p in eng.pack.params):
            return _EngineFn.apply(eng, x, *eng.pack.params)
        return eng.forward(x, save=False)


VARIANTS = {   # name: (img, embed_dim, depth, heads, init_scale)  models/cait.py:255-480
    "cait_XXS24_224": (224, 192, 24, 4, 1e-5), "cait_XXS24": (384, 192, 24, 4, 1e-5),
    "cait_XXS36_224": (224, 192, 36, 4, 1e-5), "cait_XXS36": (384, 192, 36, 4, 1e-5),
    "cait_XS24": (384, 288, 24, 6, 1e-5), "cait_S24_224": (224, 384, 24, 8, 1e-5),
    "cait_S24": (384, 384, 24, 8, 1e-5), "cait_S36": (384, 384, 36, 8, 1e-6),
    "cait_M36": (384, 768, 36, 16, 1e-6), "cait_M48": (448, 768, 48, 16, 1e-6),
}


def create_cait(arch, pretrained=False, **kwargs):
    if pretrained:
        raise RuntimeError("pretrained weights cannot be downloaded here; load_state_dict() a local checkpoint "
                           "(keys as in models/cait.py:381-385 without the 'module.' prefix)")
    img, d, depth, heads, scale = VARIANTS[arch]
    from functools import partial
    return cait_models(img_size=img, patch_size=16, embed_dim=d, depth=depth, num_heads=heads, mlp_ratio=4,
                       qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), init_scale=scale,
                       depth_token_only=2, **kwargs)


def _ru8(n):
    return (n + 7) // 8 * 8


class CaitEngine:
    """Forward / backward kernel sequence of cait_models."""

    def __init__(self, model: cait_models):
        self.model = model
        dev = model.pos_embed.device
        if dev.type != "cuda":
            raise VitmiError("move the model to the GPU before the first forward")
        self.T, self.R = model.compute_dtype, model.residual_dtype
        self.split3 = bool(getattr(model, "split3", False))
        # talking-heads attention as one fused op where the shape allows (VITMI_TH_FUSED=0: the three-call form, for A/B)
        self.fused_th = os.environ.get("VITMI_TH_FUSED", "1") != "0"
        if self.T == torch.float32 and self.R != torch.float32:
            raise VitmiError("fp32 compute needs an fp32 residual stream")
        self.head = _head_layers(model.head)
        if self.head is None:
            raise VitmiError("head must be Identity, Linear or Sequential(Linear[, GELU], ...)")
        self.pack = ParamPack(list(model.named_parameters()), dev, shadow=self.T == torch.bfloat16)
        self.saved = None
        self.reducer = None
        # small folds of a backward pass in one launch per flush (ops.FoldQueue; VITMI_DEFER_FOLDS=0: at once)
        self.folds = ops.FoldQueue() if os.environ.get("VITMI_DEFER_FOLDS", "1") != "0" else None
        self.profile = None
        self.gemm_impl = GEMM_AUTO

    def is_current(self):
        m = self.model
        return (self.pack.is_current() and m.compute_dtype == self.T and m.residual_dtype == self.R
                and bool(getattr(m, "split3", False)) == self.split3
                and len(self.pack.params) == sum(1 for _ in m.parameters()))

    def _w(self, p):
        return self.pack.w(p)

    def _gemm(self, A, B, C, **k):
        return engine_gemm(self, A, B, C, **k)

    def _ready(self, *objs):
        if self.reducer is None:
            return
        if self.folds is not None:
            self.folds.flush()
        ps = []
        for o in objs:
            ps.extend(o.parameters() if isinstance(o, nn.Module) else [o])
        self.reducer.section_ready(ps)

    # batched per-(image, head) products on the qkv tensor [B, Np, 3, H, hd] and on the score
    # tensors [B, H, Np, NS]
    def _scores(self, qkv, S, B, Np, H, hd, NS, scale):
        D3 = 3 * H * hd
        ops.gemm_batched(qkv, qkv, S, M=Np, N=Np, K=hd, lda=D3, ldb=D3, ldc=NS, a_kmajor=True, b_kmajor=True,
                         batch=B * H, batch_inner=H, a_bs=(Np * D3, hd), b_bs=(Np * D3, hd),
                         c_bs=(H * Np * NS, Np * NS), b_off=H * hd, alpha=scale)

    # ---------------------------------------------------------------- forward ---
    def forward(self, x, save: bool):
        m, T, R, pk = self.model, self.T, self.R, self.pack
        dev = x.device
        x = x.float() if x.dtype != torch.float32 else x
        B, Cin, Hi, Wi = x.shape
        p = m.patch_embed.patch_size
        conv = m.patch_embed.proj
        Np = (Hi // p) * (Wi // p)
        if Np != m.pos_embed.shape[1]:
            raise VitmiError(f"CaiT is fixed-size: input gives {Np} patches, pos_embed has {m.pos_embed.shape[1]} "
                             "(models/cait.py:183)")
        M, D = B * Np, m.embed_dim
        H = m.blocks[0].attn.num_heads if len(m.blocks) else m.blocks_token_only[0].attn.num_heads
        hd = D // H
        Kp = Cin * p * p
        NS = _ru8(Np)
        pk.refresh_shadow()
        f32 = torch.float32

        def new(r, c, dt):
            return torch.empty((r, c), dtype=dt, device=dev)

        def vec(n):
            return torch.empty(n, dtype=f32, device=dev)

        patches = new(M, Kp, T)
        ops.patchify(x, patches, p, cls_rows=0)
        X = new(M, D, R)
        self._gemm(patches, self._w(conv.weight).view(D, Kp), X, epilogue=EPI_PATCH_POS,
                   bias=pk.f32(conv.bias) if conv.bias is not None else None,
                   pos=pk.f32(m.pos_embed).view(Np, D), n_tok=Np)
        trunk = []
        fused_th = self.fused_th and T == torch.bfloat16 and ops.th_attn_supported(T, H, Np, hd)
        for blk in m.blocks:
            a, mlp = blk.attn, blk.mlp
            ln1, mean1, rstd1 = new(M, D, T), vec(M), vec(M)
            ops.layernorm_fwd(X, pk.f32(blk.norm1.weight), pk.f32(blk.norm1.bias), ln1, mean1, rstd1,
                              blk.norm1.eps, M=M, D=D)
            qkv = new(M, 3 * D, T)
            self._gemm(ln1, self._w(a.qkv.weight), qkv, bias=pk.f32(a.qkv.bias) if a.qkv.bias is not None else None)
            O = new(M, D, T)
            D3 = 3 * D
            if fused_th:
                # ONE op: the score rows stay in LDS, both head mixes run on the matrix pipe, nothing but O is kept
                # (cait_fused.hip; the backward recomputes the scores from q, k)
                ops.th_attn_fwd(qkv, pk.f32(a.proj_l.weight), pk.f32(a.proj_l.bias), pk.f32(a.proj_w.weight),
                                pk.f32(a.proj_w.bias), O, B, H, Np, hd, a.scale)
                S = P = Pm = None
            else:
                S = torch.empty((B, H, Np, NS), dtype=T, device=dev)
                self._scores(qkv, S, B, Np, H, hd, NS, a.scale)
                P, Pm = torch.empty_like(S), torch.empty_like(S)
                ops.th_softmax_fwd(S, pk.f32(a.proj_l.weight), pk.f32(a.proj_l.bias), pk.f32(a.proj_w.weight),
                                   pk.f32(a.proj_w.bias), P, Pm, B, H, Np, Np, NS)
                ops.gemm_batched(Pm, qkv, O, M=Np, N=hd, K=Np, lda=NS, ldb=D3, ldc=D, a_kmajor=True, b_kmajor=False,
                                 batch=B * H, batch_inner=H, a_bs=(H * Np * NS, Np * NS), b_bs=(Np * D3, hd),
                                 c_bs=(Np * D, hd), b_off=2 * D)
            X1, f1 = new(M, D, R), new(M, D, T)
            self._gemm(O, self._w(a.proj.weight), X1, epilogue=EPI_RESIDUAL, bias=pk.f32(a.proj.bias), R=X,
                       gamma=pk.f32(blk.gamma_1), C2=f1)
            ln2, mean2, rstd2 = new(M, D, T), vec(M), vec(M)
            ops.layernorm_fwd(X1, pk.f32(blk.norm2.weight), pk.f32(blk.norm2.bias), ln2, mean2, rstd2,
                              blk.norm2.eps, M=M, D=D)
            Dh = mlp.fc1.out_features
            pre, hid = new(M, Dh, T), new(M, Dh, T)
            self._gemm(ln2, self._w(mlp.fc1.weight), hid, epilogue=EPI_BIAS_GELU, bias=pk.f32(mlp.fc1.bias), C2=pre)
            X2, f2 = new(M, D, R), new(M, D, T)
            self._gemm(hid, self._w(mlp.fc2.weight), X2, epilogue=EPI_RESIDUAL, bias=pk.f32(mlp.fc2.bias), R=X1,
                       gamma=pk.f32(blk.gamma_2), C2=f2)
            if save:
                trunk.append((X, ln1, mean1, rstd1, qkv, S, P, Pm, O, f1, X1, ln2, mean2, rstd2, pre, hid, f2))
            X = X2

        # ---- class-attention stage: u = cat(cls, x) per layer, only the CLS row changes
        N1 = Np + 1
        Mu = B * N1
        Cx = new(B, D, R)
        ops.scale_cast(pk.f32(m.cls_token).view(1, D).expand(B, D).contiguous(), Cx, M=B, N=D)
        ca = []
        for blk in m.blocks_token_only:
            a, mlp = blk.attn, blk.mlp
            u = new(Mu, D, R)
            uv = u.view(B, N1 * D)
            ops.scale_cast(Cx, uv, M=B, N=D, ldo=N1 * D)                                  # row 0 of every image
            ops.scale_cast(X.view(B, Np * D), uv[:, D:], M=B, N=Np * D, ldo=N1 * D)        # rows 1..Np
            lnu, meanu, rstdu = new(Mu, D, T), vec(Mu), vec(Mu)
            ops.layernorm_fwd(u, pk.f32(blk.norm1.weight), pk.f32(blk.norm1.bias), lnu, meanu, rstdu,
                              blk.norm1.eps, M=Mu, D=D)
            lnu_cls = lnu.view(B, N1 * D)[:, :D]                                          # strided CLS rows
            q = new(B, D, T)
            self._gemm(lnu_cls, self._w(a.q.weight), q, bias=pk.f32(a.q.bias) if a.q.bias is not None else None)
            k, v = new(Mu, D, T), new(Mu, D, T)
            self._gemm(lnu, self._w(a.k.weight), k, bias=pk.f32(a.k.bias) if a.k.bias is not None else None)
            self._gemm(lnu, self._w(a.v.weight), v, bias=pk.f32(a.v.bias) if a.v.bias is not None else None)
            oc = new(B, D, T)
            psave = torch.empty(B * H * N1, dtype=f32, device=dev)
            ops.class_attn_fwd(q, k, v, D, oc, psave, B, H, N1, hd, a.scale)
            C1, g1 = new(B, D, R), new(B, D, T)
            self._gemm(oc, self._w(a.proj.weight), C1, epilogue=EPI_RESIDUAL, bias=pk.f32(a.proj.bias), R=Cx,
                       gamma=pk.f32(blk.gamma_1), C2=g1)
            ln2c, mean2c, rstd2c = new(B, D, T), vec(B), vec(B)
            ops.layernorm_fwd(C1, pk.f32(blk.norm2.weight), pk.f32(blk.norm2.bias), ln2c, mean2c, rstd2c,
                              blk.norm2.eps, M=B, D=D)
            Dh = mlp.fc1.out_features
            prec, hidc = new(B, Dh, T), new(B, Dh, T)
            self._gemm(ln2c, self._w(mlp.fc1.weight), hidc, epilogue=EPI_BIAS_GELU, bias=pk.f32(mlp.fc1.bias), C2=prec)
            C2, g2 = new(B, D, R), new(B, D, T)
            self._gemm(hidc, self._w(mlp.fc2.weight), C2, epilogue=EPI_RESIDUAL, bias=pk.f32(mlp.fc2.bias), R=C1,
                       gamma=pk.f32(blk.gamma_2), C2=g2)
            if save:
                ca.append((u, lnu, meanu, rstdu, q, k, v, oc, psave, g1, C1, ln2c, mean2c, rstd2c, prec, hidc, g2))
            Cx = C2

        feat, meanf, rstdf = torch.empty((B, D), dtype=f32, device=dev), vec(B), vec(B)
        ops.layernorm_fwd(Cx, pk.f32(m.norm.weight), pk.f32(m.norm.bias), feat, meanf, rstdf, m.norm.eps, M=B, D=D)
        acts, pres, cur = [feat], [], feat
        for lin, gelu in self.head:
            out = torch.empty((B, lin.out_features), dtype=f32, device=dev)
            bias = pk.f32(lin.bias) if lin.bias is not None else None
            if gelu:
                ph = torch.empty_like(out)
                ops.gemm(cur, pk.f32(lin.weight), out, epilogue=EPI_BIAS_GELU, bias=bias, C2=ph)
                pres.append(ph)
            else:
                ops.gemm(cur, pk.f32(lin.weight), out, bias=bias)
                pres.append(None)
            acts.append(out)
            cur = out
        if save:
            self.saved = dict(B=B, Np=Np, D=D, H=H, hd=hd, Kp=Kp, NS=NS, patches=patches, trunk=trunk, ca=ca,
                              Cf=Cx, meanf=meanf, rstdf=rstdf, acts=acts, pres=pres)
        return cur

    # --------------------------------------------------------------- backward ---
    def backward(self, dout):
        try:
            self._backward(dout)
        except BaseException:
            if self.folds is not None:
                self.folds.clear()
            if self.reducer is not None:
                self.reducer.abort()
            raise

    def _backward(self, dout):
        s = self.saved
        if s is None:
            raise VitmiError("backward called without a saved forward (or called twice)")
        self.saved = None
        m, T, R, pk = self.model, self.T, self.R, self.pack
        B, Np, D, H, hd, NS = s["B"], s["Np"], s["D"], s["H"], s["hd"], s["NS"]
        M, N1 = B * Np, Np + 1
        Mu = B * N1
        dev = dout.device
        f32 = torch.float32
        d = dout.contiguous().float()

        def new(r, c, dt):
            return torch.empty((r, c), dtype=dt, device=dev)

        # ---- head
        acts, pres = s["acts"], s["pres"]
        if self.head and self.head[-1][1]:
            raise VitmiError("a head ending in GELU is not supported")
        for li in range(len(self.head) - 1, -1, -1):
            lin, _ = self.head[li]
            ops.gemm(d, acts[li], pk.g(lin.weight), a_kmajor=False, b_kmajor=False)
            if lin.bias is not None:
                ops.colsum(d, pk.g(lin.bias))
            dx = torch.empty((B, lin.in_features), dtype=f32, device=dev)
            if li > 0 and self.head[li - 1][1]:
                ops.gemm(d, pk.f32(lin.weight), dx, b_kmajor=False, epilogue=EPI_DGELU, aux=pres[li - 1])
            else:
                ops.gemm(d, pk.f32(lin.weight), dx, b_kmajor=False)
            d = dx

        # ---- class-attention stage.  Gu [B, Np+1, D]: row 0 = gradient of the CLS stream, rows
        # 1.. = gradient of the trunk output (both CA layers read the same x)
        Gu = torch.zeros((Mu, D), dtype=R, device=dev)
        Gc = Gu.view(B, N1 * D)[:, :D]                       # strided view of the CLS rows
        ldu = N1 * D
        cab = list(m.blocks_token_only)
        GCb = new(B, D, T)
        last = cab[-1] if cab else None
        ops.layernorm_bwd(d, s["Cf"], s["meanf"], s["rstdf"], pk.f32(m.norm.weight), None, Gc, GCb,
                          pk.g(m.norm.weight), pk.g(m.norm.bias),
                          gsum=pk.g(last.mlp.fc2.bias) if last is not None else None,
                          gb_scale=pk.f32(last.gamma_2) if last is not None else None,
                          M=B, D=D, dy_stride=D, x_stride=D, g_stride=ldu, gb_stride=D, fold=self.folds)
        self._ready(m.norm, *([m.head] if self.head else []))
        for bi in range(len(cab) - 1, -1, -1):
            blk = cab[bi]
            a, mlp = blk.attn, blk.mlp
            u, lnu, meanu, rstdu, q, k, v, oc, psave, g1, C1, ln2c, mean2c, rstd2c, prec, hidc, g2 = s["ca"].pop()
            Dh = mlp.fc1.out_features
            ops.colsum_mul(Gc, g2, pk.g(blk.gamma_2), M=B, N=D, ldx=ldu, ldy=D)
            dHc = new(B, Dh, T)
            self._gemm(GCb, self._w(mlp.fc2.weight), dHc, b_kmajor=False, epilogue=EPI_DGELU, aux=prec)
            self._gemm(GCb, hidc, pk.g(mlp.fc2.weight), a_kmajor=False, b_kmajor=False)
            dln2c = new(B, D, T)
            self._gemm(dHc, self._w(mlp.fc1.weight), dln2c, b_kmajor=False)
            self._gemm(dHc, ln2c, pk.g(mlp.fc1.weight), a_kmajor=False, b_kmajor=False)
            ops.colsum(dHc, pk.g(mlp.fc1.bias))
            ops.layernorm_bwd(dln2c, C1, mean2c, rstd2c, pk.f32(blk.norm2.weight), Gc, Gc, GCb,
                              pk.g(blk.norm2.weight), pk.g(blk.norm2.bias), gsum=pk.g(a.proj.bias),
                              gb_scale=pk.f32(blk.gamma_1), M=B, D=D, g_stride=ldu, gb_stride=D, fold=self.folds)
            ops.colsum_mul(Gc, g1, pk.g(blk.gamma_1), M=B, N=D, ldx=ldu, ldy=D)
            doc = new(B, D, T)
            self._gemm(GCb, self._w(a.proj.weight), doc, b_kmajor=False)
            self._gemm(GCb, oc, pk.g(a.proj.weight), a_kmajor=False, b_kmajor=False)
            dq, dk, dv = new(B, D, T), new(Mu, D, T), new(Mu, D, T)
            ops.class_attn_bwd(q, k, v, D, doc, psave, dq, dk, dv, D, B, H, N1, hd, a.scale)
            # d LN(u) = dk Wk + dv Wv (+ dq Wq on the CLS rows), accumulated in fp32
            dlnu = torch.empty((Mu, D), dtype=f32, device=dev)
            self._gemm(dk, self._w(a.k.weight), dlnu, b_kmajor=False)
            self._gemm(dv, self._w(a.v.weight), dlnu, b_kmajor=False, accumulate=True)
            self._gemm(dq, self._w(a.q.weight), dlnu.view(B, ldu)[:, :D], b_kmajor=False, accumulate=True)
            lnu_cls = lnu.view(B, ldu)[:, :D]
            self._gemm(dk, lnu, pk.g(a.k.weight), a_kmajor=False, b_kmajor=False)
            self._gemm(dv, lnu, pk.g(a.v.weight), a_kmajor=False, b_kmajor=False)
            self._gemm(dq, lnu_cls, pk.g(a.q.weight), a_kmajor=False, b_kmajor=False)
            if a.k.bias is not None:
                ops.colsum(dk, pk.g(a.k.bias))
                ops.colsum(dv, pk.g(a.v.bias))
                ops.colsum(dq, pk.g(a.q.bias))
            # through LN(u): adds into Gu (row 0 also carries the residual path of the CLS stream)
            ops.layernorm_bwd(dlnu, u, meanu, rstdu, pk.f32(blk.norm1.weight), Gu, Gu, None,
                              pk.g(blk.norm1.weight), pk.g(blk.norm1.bias), M=Mu, D=D, fold=self.folds)
            if bi > 0:      # operand copy of the CLS gradient for the previous CA layer's MLP branch
                prev = cab[bi - 1]
                ops.scale_cast(Gc, GCb, pk.f32(prev.gamma_2), M=B, N=D, ldx=ldu)
                ops.colsum_mul(Gc, pk.f32(prev.gamma_2).view(1, D).expand(B, D).contiguous(), pk.g(prev.mlp.fc2.bias),
                               M=B, N=D, ldx=ldu, ldy=D)
            self._ready(blk)
        ops.colsum(Gc, pk.g(m.cls_token).view(-1), M=B, N=D, ld=ldu)      # d cls_token = sum over images

        # ---- trunk.  G [B*Np, D] = rows 1.. of Gu, Gb = cast(G * gamma_2 of the block entered)
        G = new(M, D, R)
        ops.scale_cast(Gu.view(B, ldu)[:, D:], G.view(B, Np * D), M=B, N=Np * D, ldx=ldu)
        tb = list(m.blocks)
        Gb = new(M, D, T)
        if tb:
            ops.scale_cast(G, Gb, pk.f32(tb[-1].gamma_2), M=M, N=D)
            ops.colsum(Gb, pk.g(tb[-1].mlp.fc2.bias))
        else:
            ops.scale_cast(G, Gb, None, M=M, N=D)
        D3 = 3 * D
        for bi in range(len(tb) - 1, -1, -1):
            blk = tb[bi]
            a, mlp = blk.attn, blk.mlp
            X, ln1, mean1, rstd1, qkv, S, P, Pm, O, f1, X1, ln2, mean2, rstd2, pre, hid, f2 = s["trunk"].pop()
            Dh = mlp.fc1.out_features
            ops.colsum_mul(G, f2, pk.g(blk.gamma_2), M=M, N=D)
            dH = new(M, Dh, T)
            fold = dgelu_gemm_with_bias_grad(self, Gb, self._w(mlp.fc2.weight), dH, pre, pk.g(mlp.fc1.bias))
            self._gemm(Gb, hid, pk.g(mlp.fc2.weight), a_kmajor=False, b_kmajor=False)
            dln2 = new(M, D, T)
            self._gemm(dH, self._w(mlp.fc1.weight), dln2, b_kmajor=False)
            self._gemm(dH, ln2, pk.g(mlp.fc1.weight), a_kmajor=False, b_kmajor=False)
            fold()
            ops.layernorm_bwd(dln2, X1, mean2, rstd2, pk.f32(blk.norm2.weight), G, G, Gb,
                              pk.g(blk.norm2.weight), pk.g(blk.norm2.bias), gsum=pk.g(a.proj.bias),
                              gb_scale=pk.f32(blk.gamma_1), M=M, D=D, fold=self.folds)
            ops.colsum_mul(G, f1, pk.g(blk.gamma_1), M=M, N=D)
            dO = new(M, D, T)
            self._gemm(Gb, self._w(a.proj.weight), dO, b_kmajor=False)
            self._gemm(Gb, O, pk.g(a.proj.weight), a_kmajor=False, b_kmajor=False)
            # talking-heads attention backward
            dqkv = new(M, D3, T)
            if S is None:
                # fused form: the scores are recomputed inside the row kernel, which also forms dP' = dO v^T, runs the softmax
                # backward through both mixes and writes the four mixing-parameter gradients; dS and P' (scratch here) go
                # through HBM once to the products kernel, which writes dQ, dK and dV
                NSb = 224                # the kernel writes all 224 key slots of a score row (no per-tile store branches)
                dS = torch.empty((B, H, Np, NSb), dtype=T, device=dev)
                Pm = torch.empty((B, H, Np, NSb), dtype=T, device=dev)
                ops.th_attn_bwd(qkv, dO, pk.f32(a.proj_l.weight), pk.f32(a.proj_l.bias), pk.f32(a.proj_w.weight),
                                pk.f32(a.proj_w.bias), dqkv, dS, Pm, NSb, pk.g(a.proj_l.weight), pk.g(a.proj_l.bias),
                                pk.g(a.proj_w.weight), pk.g(a.proj_w.bias), B, H, Np, hd, a.scale)
            else:
                dPm = torch.empty_like(S)
                ops.gemm_batched(dO, qkv, dPm, M=Np, N=Np, K=hd, lda=D, ldb=D3, ldc=NS, a_kmajor=True, b_kmajor=True,
                                 batch=B * H, batch_inner=H, a_bs=(Np * D, hd), b_bs=(Np * D3, hd),
                                 c_bs=(H * Np * NS, Np * NS), b_off=2 * D)
                dS = torch.empty_like(S)
                ops.th_softmax_bwd(S, P, dPm, pk.f32(a.proj_l.weight), pk.f32(a.proj_w.weight), dS,
                                   pk.g(a.proj_l.weight), pk.g(a.proj_l.bias), pk.g(a.proj_w.weight),
                                   pk.g(a.proj_w.bias), B, H, Np, Np, NS)
                ops.gemm_batched(dS, qkv, dqkv, M=Np, N=hd, K=Np, lda=NS, ldb=D3, ldc=D3, a_kmajor=True, b_kmajor=False,
                                 batch=B * H, batch_inner=H, a_bs=(H * Np * NS, Np * NS), b_bs=(Np * D3, hd),
                                 c_bs=(Np * D3, hd), b_off=D, alpha=a.scale)                          # dQ = scale dS K
            if S is not None:            # three-call form: the two products that contract over the queries
                ops.gemm_batched(Pm, dO, dqkv, M=Np, N=hd, K=Np, lda=NS, ldb=D, ldc=D3, a_kmajor=False, b_kmajor=False,
                                 batch=B * H, batch_inner=H, a_bs=(H * Np * NS, Np * NS), b_bs=(Np * D, hd),
                                 c_bs=(Np * D3, hd), c_off=2 * D)                                   # dV = P'^T dO
                ops.gemm_batched(dS, qkv, dqkv, M=Np, N=hd, K=Np, lda=NS, ldb=D3, ldc=D3, a_kmajor=False, b_kmajor=False,
                                 batch=B * H, batch_inner=H, a_bs=(H * Np * NS, Np * NS), b_bs=(Np * D3, hd),
                                 c_bs=(Np * D3, hd), c_off=D, alpha=a.scale)                          # dK = scale dS^T Q
            dln1 = new(M, D, T)
            self._gemm(dqkv, self._w(a.qkv.weight), dln1, b_kmajor=False)
            self._gemm(dqkv, ln1, pk.g(a.qkv.weight), a_kmajor=False, b_kmajor=False)
            if a.qkv.bias is not None:
                ops.colsum(dqkv, pk.g(a.qkv.bias))
            prev = tb[bi - 1] if bi > 0 else None
            ops.layernorm_bwd(dln1, X, mean1, rstd1, pk.f32(blk.norm1.weight), G, G, Gb,
                              pk.g(blk.norm1.weight), pk.g(blk.norm1.bias),
                              gsum=pk.g(prev.mlp.fc2.bias) if prev is not None else None,
                              gb_scale=pk.f32(prev.gamma_2) if prev is not None else None, M=M, D=D, fold=self.folds)
            self._ready(blk)

        # ---- embeddings (no CLS row in the trunk: pos_embed is [Np, D])
        conv = m.patch_embed.proj
        Kp = s["Kp"]
        dpos = torch.empty(Np * D, dtype=f32, device=dev)
        ops.colsum(G, dpos, M=B, N=Np * D, ld=Np * D)
        ops.cast(dpos, pk.g(m.pos_embed).view(-1))
        self._gemm(Gb, s["patches"], pk.g(conv.weight).view(D, Kp), a_kmajor=False, b_kmajor=False)
        if conv.bias is not None:
            ops.colsum(dpos.view(Np, D), pk.g(conv.bias))
        self._ready(m.cls_token, m.pos_embed, m.patch_embed)
        if self.folds is not None:
            self.folds.flush()
        if self.reducer is not None:
            self.reducer.finish()
