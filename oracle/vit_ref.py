"""ORACLE (test infrastructure only — never imported by vit_torch_amd).

Plain-PyTorch fp32 restatement of the DINO ViT the reference fetches at run
time with `torch.hub.load('facebookresearch/dino:main', arch)`
(/root/reference/models/vision_all.py:156).  That repository is an UNPINNED
third-party dependency (branch `main`) and is absent from /root/reference and
from this container, so this file restates its published algorithm and
anchors on what IS in the reference:

  * the attribute surface the reference touches: `.patch_embed.proj` (Conv2d,
    models/vision_all.py:159-167), `.norm.weight` (:169), `.head` (:170);
  * the sibling implementations of the same timm-lineage block that are in the
    reference: `LayerScale_Block` (models/cait.py:130-150) with LayerScale = 1
    and identity talking-heads mixes, and `Mlp` (models/swin.py:14-30) —
    tests/test_oracle_golden.py pins `Block` below against golden vectors
    generated from those reference classes (tests/golden/gen_golden.py).

PARITY UNPINNED for the parts that exist only upstream:
(1) LayerNorm eps 1e-6, (2) scale applied AFTER q·kᵀ, (3) `forward` returns
norm(x)[:, 0] WITHOUT applying `.head` (so `apply_head=False` is the upstream
behaviour; the build's factory sets it explicitly), (4) bicubic pos-embed
interpolation with the +0.1 offset.  State-dict names follow upstream:
cls_token, pos_embed, patch_embed.proj.*, blocks.{i}.{norm1,attn.{qkv,proj},
norm2,mlp.{fc1,fc2}}.*, norm.*.
"""
from __future__ import annotations

import math
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F


class Mlp(nn.Module):
    """fc2(GELU(fc1(x))) — same as /root/reference/models/swin.py:14-30 (drop=0)."""

    def __init__(self, in_features, hidden_features=None, out_features=None):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features, out_features)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class Attention(nn.Module):
    """Vanilla MHSA.  Structure of /root/reference/models/swin.py:119-144 minus
    bias/mask; `scale_before` selects q*scale before the product (CaiT/Swin,
    models/cait.py:114, models/swin.py:123) or after it (DINO upstream)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, scale_before=False):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.scale_before = scale_before
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        if self.scale_before:
            attn = (q * self.scale) @ k.transpose(-2, -1)
        else:
            attn = (q @ k.transpose(-2, -1)) * self.scale
        attn = attn.softmax(dim=-1)
        x = (attn @ v).transpose(1, 2).reshape(B, N, C)
        return self.proj(x)


class Block(nn.Module):
    """x + attn(norm1(x)); x + mlp(norm2(x)) — /root/reference/models/cait.py:147-150
    with gamma_1 = gamma_2 = 1 and drop_path = 0."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, norm_layer=nn.LayerNorm,
                 scale_before=False):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, scale_before=scale_before)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio))

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        x = x + self.mlp(self.norm2(x))
        return x


class PatchEmbed(nn.Module):
    """Conv2d(k = s = patch) -> flatten(2).transpose(1, 2)
    (same op as /root/reference/models/swin.py:434,445 without the norm)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size = img_size
        self.patch_size = patch_size
        self.num_patches = (img_size // patch_size) * (img_size // patch_size)
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


def interpolate_pos_encoding(pos_embed, npatch, w, h, patch_size):
    """Bicubic resize of the patch grid of `pos_embed` [1, 1+N, D] to the grid of a
    w x h input (upstream DINO behaviour, [recall] — SURVEY.md §3.2)."""
    N = pos_embed.shape[1] - 1
    if npatch == N and w == h:
        return pos_embed
    class_pos = pos_embed[:, 0]
    patch_pos = pos_embed[:, 1:]
    dim = pos_embed.shape[-1]
    w0 = w // patch_size + 0.1
    h0 = h // patch_size + 0.1
    side = int(math.sqrt(N))
    patch_pos = F.interpolate(
        patch_pos.reshape(1, side, side, dim).permute(0, 3, 1, 2),
        scale_factor=(w0 / math.sqrt(N), h0 / math.sqrt(N)), mode="bicubic")
    assert int(w0) == patch_pos.shape[-2] and int(h0) == patch_pos.shape[-1]
    patch_pos = patch_pos.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat((class_pos.unsqueeze(0), patch_pos), dim=1)


class VisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=0, embed_dim=768,
                 depth=12, num_heads=12, mlp_ratio=4.0, qkv_bias=True,
                 norm_layer=partial(nn.LayerNorm, eps=1e-6), apply_head=False):
        super().__init__()
        self.num_features = self.embed_dim = embed_dim
        self.apply_head = apply_head
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
        self.blocks = nn.ModuleList([
            Block(embed_dim, num_heads, mlp_ratio, qkv_bias, norm_layer) for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def prepare_tokens(self, x):
        B, _, w, h = x.shape
        x = self.patch_embed(x)
        cls = self.cls_token.expand(B, -1, -1)
        x = torch.cat((cls, x), dim=1)
        return x + interpolate_pos_encoding(self.pos_embed, x.shape[1] - 1, w, h,
                                            self.patch_embed.patch_size)

    def forward_features(self, x):
        x = self.prepare_tokens(x)
        for blk in self.blocks:
            x = blk(x)
        return self.norm(x)[:, 0]

    def forward(self, x):
        x = self.forward_features(x)
        return self.head(x) if self.apply_head else x


ARCHS = {
    # name: (patch, embed_dim, depth, heads)  — DINO vit_small / vit_base [recall]
    "dino_vits16": (16, 384, 12, 6),
    "dino_vits8": (8, 384, 12, 6),
    "dino_vitb16": (16, 768, 12, 12),
    "dino_vitb8": (8, 768, 12, 12),
}


def get_classifier_head(in_features, classifier_units, act=None):
    """Restatement of VisionModelZoo.get_classifier_head
    (/root/reference/models/vision_all.py:299-320): hidden Linear(bias=True)+act,
    last Linear(bias=False); ONE shared activation instance."""
    act = act if act is not None else nn.GELU()
    layers = []
    if isinstance(classifier_units, int):
        classifier_units = [classifier_units]
    if isinstance(classifier_units, list):
        for i, v in enumerate(classifier_units):
            fin = in_features if i == 0 else classifier_units[i - 1]
            last = i >= len(classifier_units) - 1
            layers.append(nn.Linear(fin, v, bias=not last))
            if not last:
                layers.append(act)
    return nn.Sequential(*layers)


def build(arch, classifier=None, apply_head=True, img_size=224, **kw):
    p, d, depth, heads = ARCHS[arch]
    m = VisionTransformer(img_size=img_size, patch_size=p, embed_dim=d, depth=depth,
                          num_heads=heads, apply_head=apply_head and classifier is not None, **kw)
    if classifier is not None:
        m.head = get_classifier_head(d, classifier)
    return m


def seeded_init_(model, seed=1):
    """Deterministic weights for parity runs: trunc-normal(0.02)
    for >=2-D parameters, LN gamma 1 / beta 0, small random biases so that bias
    gradients are exercised.  Parameter order = model.named_parameters()."""
    g = torch.Generator("cpu").manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if ".norm" in name or name.startswith("norm") or "gamma_" in name:
                if name.endswith("weight") or "gamma_" in name:
                    p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif p.dim() >= 2:
                p.copy_(torch.nn.init.trunc_normal_(torch.empty(p.shape), std=0.02, generator=g))
            else:
                p.copy_(0.02 * torch.randn(p.shape, generator=g))
    return model
