"""Swin Transformer (classification) on libvitmi kernels.

Drop-in for /root/reference/models/swin.py `SwinTransformer` (:458-600), `configs`
(:768-820) and `get_swin_model` (:823-844): same constructor arguments (unknown kwargs such
as the stray 'crop' are ignored, as the reference's **kwargs does), parameter and buffer
names/shapes (state-dict compatible incl. relative_position_index and attn_mask) and call
contract.  `forward` runs `SwinEngine`, an explicit forward/backward kernel sequence in
which tokens never leave token order: roll / window_partition / window_reverse are folded
into the window-attention kernels' addressing.

DropPath (models/swin.py:203,267-268; per-block rates from linspace(0, drop_path_rate,
sum(depths)), :520): in training mode each block draws two per-sample Bernoulli(keep)
masks (attention branch, MLP branch); keep/keep_prob rides as the per-row-group scale of
the branch GEMM's residual epilogue and of the matching backward `Gb` emission, so a
dropped sample costs no extra pass.  The reference never calls .eval(), so its DropPath is
active in every forward; this mirror follows nn.Module semantics (active iff
`model.training`).  `model.drop_path_keep_masks` (nested [stage][block] -> (m_attn[B],
m_mlp[B]) of 0/1) pins the draws for parity tests; None -> fresh draws from torch's
device generator each forward.
"""
from __future__ import annotations

from typing import Optional

import os

import torch
import torch.nn as nn

from . import ops
from ._lib import EPI_BIAS_GELU, EPI_DGELU, EPI_RESIDUAL, GEMM_AUTO, VitmiError
from .packing import ParamPack
from .vit import Mlp, _DT, _EngineFn, _head_layers, _trunc_normal_, dgelu_gemm_with_bias_grad, engine_gemm


def _relative_position_index(ws):
    coords = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij"))
    cf = torch.flatten(coords, 1)
    rel = (cf[:, :, None] - cf[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def _shift_mask(H, W, ws, shift):
    """-100 / 0 mask of the shifted windows, [nW, ws*ws, ws*ws] (models/swin.py:208-229)."""
    img = torch.zeros((H, W))
    cnt = 0
    for h in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for w in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[h, w] = cnt
            cnt += 1
    mw = img.view(H // ws, ws, W // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return am.masked_fill(am != 0, -100.0).masked_fill(am == 0, 0.0)


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=True):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        self.scale = (dim // num_heads) ** -0.5
        ws = window_size[0]
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) * (2 * ws - 1), num_heads))
        self.register_buffer("relative_position_index", _relative_position_index(ws))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        _trunc_normal_(self.relative_position_bias_table)


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, input_resolution, num_heads, window_size, shift_size, mlp_ratio, qkv_bias,
                 drop_path=0.0):
        super().__init__()
        self.dim, self.input_resolution, self.num_heads = dim, input_resolution, num_heads
        self.drop_path_rate = float(drop_path)
        self.window_size, self.shift_size = window_size, shift_size
        if min(input_resolution) <= window_size:      # models/swin.py:192-195
            self.shift_size = 0
            self.window_size = min(input_resolution)
        assert 0 <= self.shift_size < self.window_size, "shift_size must in 0-window_size"
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, (self.window_size, self.window_size), num_heads, qkv_bias)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        mask = _shift_mask(*input_resolution, self.window_size, self.shift_size) if self.shift_size > 0 else None
        self.register_buffer("attn_mask", mask)


class PatchMerging(nn.Module):
    def __init__(self, input_resolution, dim):
        super().__init__()
        self.input_resolution, self.dim = input_resolution, dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)


class BasicLayer(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio, qkv_bias, downsample,
                 drop_path=0.0):
        super().__init__()
        self.dim, self.input_resolution, self.depth = dim, input_resolution, depth
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, input_resolution, num_heads, window_size,
                                 0 if i % 2 == 0 else window_size // 2, mlp_ratio, qkv_bias,
                                 drop_path[i] if isinstance(drop_path, (list, tuple)) else drop_path)
            for i in range(depth)])
        self.downsample = PatchMerging(input_resolution, dim) if downsample else None


class PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim, patch_norm):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.patches_resolution = [img_size // patch_size, img_size // patch_size]
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        self.in_chans, self.embed_dim = in_chans, embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.LayerNorm(embed_dim) if patch_norm else None


class SwinTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=4, in_chans=3, num_classes=1000, embed_dim=96,
                 depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), window_size=7, mlp_ratio=4.0, qkv_bias=True,
                 qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.1, norm_layer=nn.LayerNorm,
                 ape=False, patch_norm=True, use_checkpoint=False, compute_dtype="bf16", residual_dtype="fp32",
                 **_ignored):
        super().__init__()
        if drop_rate or attn_drop_rate or qk_scale is not None or ape or use_checkpoint:
            raise VitmiError("dropout / qk_scale / ape / checkpointing are not supported (the reference's configs "
                             "never set them, models/swin.py:768-820)")
        if not 0.0 <= drop_path_rate < 1.0:
            raise VitmiError("drop_path_rate must be in [0, 1)")
        if norm_layer is not nn.LayerNorm:
            raise VitmiError("norm_layer must be nn.LayerNorm (eps 1e-5), as everywhere in models/swin.py")
        self.num_classes = num_classes
        self.num_layers = len(depths)
        self.embed_dim = embed_dim
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))
        self.mlp_ratio = mlp_ratio
        self.apply_head = True
        self.compute_dtype = _DT[compute_dtype]
        self.split3 = compute_dtype == "bf16x3"        # vit.py: fp32 GEMMs as three bf16 products (ops.gemm_split3)
        # the residual stream between the blocks: "fp32" (default: the reference trains in fp32, and the fp32 stream ends a
        # 50-step run within 0.06-0.08 % of the oracle loss, tests/test_training_curve_gpu.py), "bf16" (what bench.py times:
        # 1.1-2.2 % on the same test), or "auto" = follow the compute dtype
        self.residual_dtype = self.compute_dtype if residual_dtype == "auto" else _DT[residual_dtype]
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim, patch_norm)
        pr = self.patch_embed.patches_resolution
        self.patches_resolution = pr
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]   # models/swin.py:520
        self.drop_path_keep_masks = None
        self.layers = nn.ModuleList([
            BasicLayer(int(embed_dim * 2 ** i), (pr[0] // 2 ** i, pr[1] // 2 ** i), depths[i], num_heads[i],
                       window_size, mlp_ratio, qkv_bias, downsample=i < self.num_layers - 1,
                       drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])])
            for i in range(self.num_layers)])
        self.norm = nn.LayerNorm(self.num_features)
        self.avgpool = nn.AdaptiveAvgPool1d(1)
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
        for m in self.modules():
            if isinstance(m, nn.Linear):
                _trunc_normal_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.LayerNorm):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
        self._engine: Optional[SwinEngine] = None

    def engine(self):
        if self._engine is None or not self._engine.is_current():
            self._engine = SwinEngine(self)
        return self._engine

    def forward(self, x):
        if not x.is_cuda:
            raise VitmiError("vit_torch_amd models run on an MI355X (HIP) device; got a CPU tensor "
                             "and there is no CPU fallback")
        eng = self.engine()
        if torch.is_grad_enabled() and any(p.requires_grad for p in eng.pack.params):
            return _EngineFn.apply(eng, x, *eng.pack.params)
        return eng.forward(x, save=False)


configs = {   # models/swin.py:768-820 (the 224 / window-7 classification variants)
    "swin_tiny_patch4_window7_224": dict(drop_path_rate=0.2, embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=7),
    "swin_small_patch4_window7_224": dict(drop_path_rate=0.3, embed_dim=96, depths=[2, 2, 18, 2], num_heads=[3, 6, 12, 24], window_size=7),
    "swin_base_patch4_window7_224": dict(drop_path_rate=0.5, embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=7),
    "swin_large_patch4_window7_224": dict(embed_dim=192, depths=[2, 2, 18, 2], num_heads=[6, 12, 24, 48], window_size=7),
}


def get_swin_model(arch="swin_tiny_patch4_window7_224", pretrained=False, **kwargs):
    """models/swin.py:823-844: the architecture is resolved by prefix."""
    name = None
    for k in configs:
        if arch.startswith(k):
            name = k
            break
    if name is None:
        raise ValueError("arch [{}] not found!".format(arch))
    if pretrained:
        raise RuntimeError("pretrained weights cannot be downloaded here; load_state_dict(strict=False) a local "
                           "checkpoint['model'] as models/swin.py:838 does")
    cfg = dict(configs[name])
    cfg.update(kwargs)
    return SwinTransformer(**cfg)


class SwinEngine:
    def __init__(self, model: SwinTransformer):
        self.model = model
        dev = model.norm.weight.device
        if dev.type != "cuda":
            raise VitmiError("move the model to the GPU before the first forward")
        self.T, self.R = model.compute_dtype, model.residual_dtype
        self.split3 = bool(getattr(model, "split3", False))
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
        # bf16 patch-embed weight at the padded contraction width (see forward); allocated here so
        # that it never lands in a graph's private pool
        Kp = model.patch_embed.proj.weight[0].numel()
        self._pe_kld = max(64, (Kp + 31) // 32 * 32) if self.T == torch.bfloat16 else Kp
        self._pe_wpad = (torch.zeros((model.embed_dim, self._pe_kld), dtype=self.T, device=dev)
                         if self._pe_kld != Kp else None)
        self.gemm_impl = GEMM_AUTO
        # DropPath keep probabilities of the blocks with a non-zero rate, built ONCE on the device (a torch.tensor(list,
        # device=...) per forward is a pageable-host copy + stream synchronise: not allowed while a HIP graph is capturing
        # — ADVICE r03; the rates are fixed at construction, models/swin.py:531)
        self._dp_slot, keeps = {}, []
        for si_, layer_ in enumerate(model.layers):
            for bi_, blk_ in enumerate(layer_.blocks):
                if blk_.drop_path_rate > 0.0:
                    self._dp_slot[(si_, bi_)] = len(keeps)
                    keeps.append(1.0 - blk_.drop_path_rate)
        self._dp_rates = tuple(keeps)
        self._dp_keep = (torch.tensor(keeps, dtype=torch.float32, device=dev).view(-1, 1, 1) if keeps else None)

    def _dp_current(self):
        """The cached keep table still describes the model's blocks (a rate edited after construction rebuilds it)."""
        keeps = tuple(1.0 - b.drop_path_rate for l in self.model.layers for b in l.blocks if b.drop_path_rate > 0.0)
        return keeps == self._dp_rates

    def is_current(self):
        m = self.model
        return (self.pack.is_current() and m.compute_dtype == self.T and m.residual_dtype == self.R
                and bool(getattr(m, "split3", False)) == self.split3
                and len(self.pack.params) == sum(1 for _ in m.parameters()) and self._dp_current())

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

    # ---------------------------------------------------------------- forward ---
    def forward(self, x, save: bool):
        m, T, R, pk = self.model, self.T, self.R, self.pack
        dev = x.device
        f32 = torch.float32
        x = x.float() if x.dtype != f32 else x
        B, Cin, Hi, Wi = x.shape
        pe = m.patch_embed
        assert Hi == pe.img_size[0] and Wi == pe.img_size[1], \
            f"Input image size ({Hi}*{Wi}) doesn't match model ({pe.img_size[0]}*{pe.img_size[1]})."
        p = pe.patch_size[0]
        pk.refresh_shadow()

        def new(r, c, dt):
            return torch.empty((r, c), dtype=dt, device=dev)

        def vec(n):
            return torch.empty(n, dtype=f32, device=dev)

        Hh, Ww = pe.patches_resolution
        C = m.embed_dim
        M = B * Hh * Ww
        Kp = Cin * p * p
        # the tile GEMM contracts in steps of 32 and needs K >= 64: Swin's 4x4x3 = 48-wide patches are
        # written at a padded row stride (the extra columns zero, weight likewise) instead of
        # falling to the any-shape kernel
        Kld = self._pe_kld
        patches = new(M, Kld, T)
        ops.patchify(x, patches, p, cls_rows=0)
        Y = new(M, C, R)
        if Kld != Kp:
            ops.scale_cast(pk.f32(pe.proj.weight).view(C, Kp), self._pe_wpad, M=C, N=Kp, ldx=Kp, ldo=Kld)
            Wpe = self._pe_wpad
        else:
            Wpe = self._w(pe.proj.weight).view(C, Kp)
        self._gemm(patches, Wpe, Y, bias=pk.f32(pe.proj.bias))
        pe_saved = None
        if pe.norm is not None:
            X, meanp, rstdp = new(M, C, R), vec(M), vec(M)
            ops.layernorm_fwd(Y, pk.f32(pe.norm.weight), pk.f32(pe.norm.bias), X, meanp, rstdp, pe.norm.eps, M=M, D=C)
            pe_saved = (Y, meanp, rstdp)
        else:
            X = Y
        # DropPath (timm's, as used at models/swin.py:203,267-268): per-sample Bernoulli(keep) / keep for each of the two
        # branches of every block with a non-zero rate — ONE uniform draw for the whole step instead of a bernoulli_ + div_
        # pair per block (22 us each for 512 numbers: 0.2 ms of the Swin-T step)
        dp_slot, dp_scales = self._dp_slot, None
        if m.training and m.drop_path_keep_masks is None and self._dp_keep is not None:
            kt = self._dp_keep
            dp_scales = (torch.rand((kt.shape[0], 2, B), dtype=f32, device=dev) < kt).to(f32) / kt
        stages = []
        for si, layer in enumerate(m.layers):
            Hh, Ww = layer.input_resolution
            C = layer.dim
            M = B * Hh * Ww
            L = Hh * Ww
            blocks = []
            for bi, blk in enumerate(layer.blocks):
                a, mlp = blk.attn, blk.mlp
                rs1 = rs2 = None
                if m.training and blk.drop_path_rate > 0.0:
                    # timm DropPath: per-sample Bernoulli(keep) / keep on each of the two branches
                    keep = 1.0 - blk.drop_path_rate
                    if m.drop_path_keep_masks is not None:
                        k1, k2 = m.drop_path_keep_masks[si][bi]
                        rs = torch.stack([k1, k2]).to(device=dev, dtype=f32) / keep
                    else:       # Bernoulli(keep) / keep per sample and branch, drawn for ALL blocks at once (below)
                        rs = dp_scales[dp_slot[(si, bi)]]
                    rs1, rs2 = rs[0].contiguous(), rs[1].contiguous()
                ws, sh, H = blk.window_size, blk.shift_size, a.num_heads
                hd = C // H
                N = ws * ws
                Bw = B * (Hh // ws) * (Ww // ws)
                ln1, mean1, rstd1 = new(M, C, T), vec(M), vec(M)
                ops.layernorm_fwd(X, pk.f32(blk.norm1.weight), pk.f32(blk.norm1.bias), ln1, mean1, rstd1,
                                  blk.norm1.eps, M=M, D=C)
                qkv = new(M, 3 * C, T)
                self._gemm(ln1, self._w(a.qkv.weight), qkv, bias=pk.f32(a.qkv.bias) if a.qkv.bias is not None else None)
                bias = torch.empty(H * N * N, dtype=f32, device=dev)
                ops.relpos_bias_gather(pk.f32(a.relative_position_bias_table), a.relative_position_index, bias,
                                       a.relative_position_bias_table.shape[0], H, N)
                O, lse = new(M, C, T), vec(Bw * H * N)
                ops.win_attn_fwd(qkv, O, lse, bias, blk.attn_mask, Bw, H, N, hd, Hh, Ww, ws, sh, a.scale)
                X1 = new(M, C, R)
                self._gemm(O, self._w(a.proj.weight), X1, epilogue=EPI_RESIDUAL, bias=pk.f32(a.proj.bias), R=X,
                           rowscale=rs1, rows_per_group=L)
                ln2, mean2, rstd2 = new(M, C, T), vec(M), vec(M)
                ops.layernorm_fwd(X1, pk.f32(blk.norm2.weight), pk.f32(blk.norm2.bias), ln2, mean2, rstd2,
                                  blk.norm2.eps, M=M, D=C)
                Dh = mlp.fc1.out_features
                pre, hid = new(M, Dh, T), new(M, Dh, T)
                self._gemm(ln2, self._w(mlp.fc1.weight), hid, epilogue=EPI_BIAS_GELU, bias=pk.f32(mlp.fc1.bias), C2=pre)
                X2 = new(M, C, R)
                self._gemm(hid, self._w(mlp.fc2.weight), X2, epilogue=EPI_RESIDUAL, bias=pk.f32(mlp.fc2.bias), R=X1,
                           rowscale=rs2, rows_per_group=L)
                if save:
                    blocks.append((X, ln1, mean1, rstd1, qkv, bias, O, lse, X1, ln2, mean2, rstd2, pre, hid, rs1, rs2))
                X = X2
            merge = None
            if layer.downsample is not None:
                ds = layer.downsample
                M2 = M // 4
                Xm = new(M2, 4 * C, R)
                ops.patch_merge(X, Xm, B, Hh, Ww, C)
                lnm, meanm, rstdm = new(M2, 4 * C, T), vec(M2), vec(M2)
                ops.layernorm_fwd(Xm, pk.f32(ds.norm.weight), pk.f32(ds.norm.bias), lnm, meanm, rstdm, ds.norm.eps,
                                  M=M2, D=4 * C)
                Xn = new(M2, 2 * C, R)
                self._gemm(lnm, self._w(ds.reduction.weight), Xn)
                if save:
                    merge = (Xm, lnm, meanm, rstdm)
                X = Xn
            if save:
                stages.append((blocks, merge))
        # final norm -> token mean -> head
        C = m.num_features
        M = B * L
        xn, meanf, rstdf = new(M, C, f32), vec(M), vec(M)
        ops.layernorm_fwd(X, pk.f32(m.norm.weight), pk.f32(m.norm.bias), xn, meanf, rstdf, m.norm.eps, M=M, D=C)
        feat = torch.empty((B, C), dtype=f32, device=dev)
        ops.token_mean_fwd(xn, feat, B, L, C)
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
            self.saved = dict(B=B, patches=patches, pe=pe_saved, stages=stages, Xf=X, meanf=meanf, rstdf=rstdf,
                              acts=acts, pres=pres, L=L)
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
        B = s["B"]
        dev = dout.device
        f32 = torch.float32
        d = dout.contiguous().float()

        def new(r, c, dt):
            return torch.empty((r, c), dtype=dt, device=dev)

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

        # token mean + final norm
        L, C = s["L"], m.num_features
        M = B * L
        dxn = new(M, C, f32)
        ops.token_mean_bwd(d, dxn, B, L, C)
        G, Gb = new(M, C, R), new(M, C, T)
        last_blk = m.layers[-1].blocks[-1]
        # Gb = (dropped-path-scaled) gradient of the last block's MLP branch output
        ops.layernorm_bwd(dxn, s["Xf"], s["meanf"], s["rstdf"], pk.f32(m.norm.weight), None, G, Gb,
                          pk.g(m.norm.weight), pk.g(m.norm.bias), gsum=pk.g(last_blk.mlp.fc2.bias),
                          gb_rowscale=s["stages"][-1][0][-1][-1], rows_per_group=L, M=M, D=C, fold=self.folds)
        self._ready(m.norm, *([m.head] if self.head else []))

        layers = list(m.layers)
        for li in range(len(layers) - 1, -1, -1):
            layer = layers[li]
            blocks_saved, merge = s["stages"].pop()
            Hh, Ww = layer.input_resolution
            C = layer.dim
            M = B * Hh * Ww
            L = Hh * Ww
            if layer.downsample is not None:
                # G / Gb currently belong to the NEXT stage's input [M/4, 2C]
                ds = layer.downsample
                Xm, lnm, meanm, rstdm = merge
                M2 = M // 4
                dlnm = new(M2, 4 * C, T)
                self._gemm(Gb, self._w(ds.reduction.weight), dlnm, b_kmajor=False)
                self._gemm(Gb, lnm, pk.g(ds.reduction.weight), a_kmajor=False, b_kmajor=False)
                Gm = new(M2, 4 * C, R)
                ops.layernorm_bwd(dlnm, Xm, meanm, rstdm, pk.f32(ds.norm.weight), None, Gm, None,
                                  pk.g(ds.norm.weight), pk.g(ds.norm.bias), M=M2, D=4 * C, fold=self.folds)
                G, Gb = new(M, C, R), new(M, C, T)
                ops.patch_merge(Gm, G, B, Hh, Ww, C, inverse=True)
                ops.scale_cast(G, Gb, None, M=M, N=C, rowscale=blocks_saved[-1][-1], rows_per_group=L)
                ops.colsum(Gb, pk.g(layer.blocks[-1].mlp.fc2.bias))
                self._ready(ds)
            blist = list(layer.blocks)
            for bi in range(len(blist) - 1, -1, -1):
                blk = blist[bi]
                a, mlp = blk.attn, blk.mlp
                X, ln1, mean1, rstd1, qkv, bias, O, lse, X1, ln2, mean2, rstd2, pre, hid, rs1, _ = blocks_saved.pop()
                rs_prev = blocks_saved[-1][-1] if bi > 0 else None     # MLP-branch factor of the block before
                ws, sh, H = blk.window_size, blk.shift_size, a.num_heads
                hd, N = C // H, ws * ws
                Bw = B * (Hh // ws) * (Ww // ws)
                Dh = mlp.fc1.out_features
                dH = new(M, Dh, T)
                fold = dgelu_gemm_with_bias_grad(self, Gb, self._w(mlp.fc2.weight), dH, pre, pk.g(mlp.fc1.bias))
                self._gemm(Gb, hid, pk.g(mlp.fc2.weight), a_kmajor=False, b_kmajor=False)
                dln2 = new(M, C, T)
                self._gemm(dH, self._w(mlp.fc1.weight), dln2, b_kmajor=False)
                self._gemm(dH, ln2, pk.g(mlp.fc1.weight), a_kmajor=False, b_kmajor=False)
                fold()
                ops.layernorm_bwd(dln2, X1, mean2, rstd2, pk.f32(blk.norm2.weight), G, G, Gb,
                                  pk.g(blk.norm2.weight), pk.g(blk.norm2.bias), gsum=pk.g(a.proj.bias),
                                  gb_rowscale=rs1, rows_per_group=L, M=M, D=C, fold=self.folds)
                dO = new(M, C, T)
                self._gemm(Gb, self._w(a.proj.weight), dO, b_kmajor=False)
                self._gemm(Gb, O, pk.g(a.proj.weight), a_kmajor=False, b_kmajor=False)
                dqkv = new(M, 3 * C, T)
                dbias = torch.empty(H * N * N, dtype=f32, device=dev)
                fuse_qb = a.qkv.bias is not None and ops.win_attn_bwd_fuses_qkv_bias(qkv, hd)
                ops.win_attn_bwd(qkv, dO, lse, bias, blk.attn_mask, dqkv, dbias, Bw, H, N, hd, Hh, Ww, ws, sh, a.scale,
                                 dqkv_bias=pk.g(a.qkv.bias) if fuse_qb else None)
                ops.relpos_bias_scatter(dbias, a.relative_position_index, pk.g(a.relative_position_bias_table),
                                        a.relative_position_bias_table.shape[0], H, N)
                dln1 = new(M, C, T)
                self._gemm(dqkv, self._w(a.qkv.weight), dln1, b_kmajor=False)
                self._gemm(dqkv, ln1, pk.g(a.qkv.weight), a_kmajor=False, b_kmajor=False)
                if a.qkv.bias is not None and not fuse_qb:
                    ops.colsum(dqkv, pk.g(a.qkv.bias))
                prev_bias = blist[bi - 1].mlp.fc2.bias if bi > 0 else None
                ops.layernorm_bwd(dln1, X, mean1, rstd1, pk.f32(blk.norm1.weight), G, G, Gb,
                                  pk.g(blk.norm1.weight), pk.g(blk.norm1.bias),
                                  gsum=pk.g(prev_bias) if prev_bias is not None else None,
                                  gb_rowscale=rs_prev, rows_per_group=L, M=M, D=C, fold=self.folds)
                self._ready(blk)

        # patch embedding (+ its LayerNorm)
        pe = m.patch_embed
        C = m.embed_dim
        Kld = s["patches"].shape[1]
        Kp = pe.proj.weight[0].numel()
        M = s["patches"].shape[0]
        if s["pe"] is not None:
            Y, meanp, rstdp = s["pe"]
            dY, dYb = new(M, C, R), new(M, C, T)
            gdt = G if R == f32 else G     # dy of this LN is the residual-stream gradient itself
            ops.layernorm_bwd(gdt, Y, meanp, rstdp, pk.f32(pe.norm.weight), None, dY, dYb,
                              pk.g(pe.norm.weight), pk.g(pe.norm.bias), gsum=pk.g(pe.proj.bias), M=M, D=C, fold=self.folds)
            Gb = dYb
        else:
            ops.colsum(Gb, pk.g(pe.proj.bias))
        if Kld != Kp:
            dWp = torch.empty((C, Kld), dtype=f32, device=Gb.device)
            self._gemm(Gb, s["patches"], dWp, a_kmajor=False, b_kmajor=False)
            ops.scale_cast(dWp, pk.g(pe.proj.weight).view(C, Kp), M=C, N=Kp, ldx=Kld, ldo=Kp)
        else:
            self._gemm(Gb, s["patches"], pk.g(pe.proj.weight).view(C, Kp), a_kmajor=False, b_kmajor=False)
        self._ready(m.patch_embed)
        if self.folds is not None:
            self.folds.flush()
        if self.reducer is not None:
            self.reducer.finish()
