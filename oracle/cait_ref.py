"""ORACLE (test infrastructure only — never imported by vit_torch_amd).

Plain-PyTorch fp32 restatement of the reference's CaiT, /root/reference/models/cait.py:
Class_Attention :21-55, LayerScale_Block_CA :57-84, Attention_talking_head :87-128,
LayerScale_Block :130-150, cait_models :155-253, variants :255-480.  Same parameter
names/shapes as the reference state_dict.  Pinned by golden vectors produced from the
reference classes themselves (tests/golden/{talking_heads,class_attention,layerscale_block,
cait_tiny}.npz; tests/test_oracle_golden.py).
"""
from functools import partial

import torch
import torch.nn as nn

from .vit_ref import Mlp, PatchEmbed


class ClassAttention(nn.Module):
    """q from token 0 only, k/v from all tokens (models/cait.py:38-55)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.k = nn.Linear(dim, dim, bias=qkv_bias)
        self.v = nn.Linear(dim, dim, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        H = self.num_heads
        q = self.q(x[:, 0]).unsqueeze(1).reshape(B, 1, H, C // H).permute(0, 2, 1, 3) * self.scale
        k = self.k(x).reshape(B, N, H, C // H).permute(0, 2, 1, 3)
        v = self.v(x).reshape(B, N, H, C // H).permute(0, 2, 1, 3)
        attn = (q @ k.transpose(-2, -1)).softmax(dim=-1)
        return self.proj((attn @ v).transpose(1, 2).reshape(B, 1, C))


class TalkingHeadAttention(nn.Module):
    """q*scale; S = q k^T; S <- proj_l over heads; softmax; P <- proj_w over heads; P v
    (models/cait.py:111-128)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.proj_l = nn.Linear(num_heads, num_heads)
        self.proj_w = nn.Linear(num_heads, num_heads)

    def forward(self, x):
        B, N, C = x.shape
        H = self.num_heads
        qkv = self.qkv(x).reshape(B, N, 3, H, C // H).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * self.scale, qkv[1], qkv[2]
        attn = q @ k.transpose(-2, -1)
        attn = self.proj_l(attn.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
        attn = attn.softmax(dim=-1)
        attn = self.proj_w(attn.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
        return self.proj((attn @ v).transpose(1, 2).reshape(B, N, C))


class LayerScaleBlock(nn.Module):
    """x + g1*attn(norm1(x)); x + g2*mlp(norm2(x))  (models/cait.py:147-150)."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, norm_layer=nn.LayerNorm,
                 init_values=1e-4):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = TalkingHeadAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio))
        self.gamma_1 = nn.Parameter(init_values * torch.ones(dim))
        self.gamma_2 = nn.Parameter(init_values * torch.ones(dim))

    def forward(self, x):
        x = x + self.gamma_1 * self.attn(self.norm1(x))
        return x + self.gamma_2 * self.mlp(self.norm2(x))


class LayerScaleBlockCA(nn.Module):
    """u = cat(cls, x); cls += g1*CA(norm1(u)); cls += g2*mlp(norm2(cls))  (models/cait.py:75-84)."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, norm_layer=nn.LayerNorm,
                 init_values=1e-4):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = ClassAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio))
        self.gamma_1 = nn.Parameter(init_values * torch.ones(dim))
        self.gamma_2 = nn.Parameter(init_values * torch.ones(dim))

    def forward(self, x, x_cls):
        u = torch.cat((x_cls, x), dim=1)
        x_cls = x_cls + self.gamma_1 * self.attn(self.norm1(u))
        return x_cls + self.gamma_2 * self.mlp(self.norm2(x_cls))


class CaiT(nn.Module):
    """cait_models (models/cait.py:155-253): no CLS token in the trunk, pos_embed [1, Np, D]."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4.0, qkv_bias=False, norm_layer=nn.LayerNorm, init_scale=1e-4,
                 depth_token_only=2, mlp_ratio_clstk=4.0):
        super().__init__()
        self.num_features = self.embed_dim = embed_dim
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        n = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n, embed_dim))
        self.blocks = nn.ModuleList([
            LayerScaleBlock(embed_dim, num_heads, mlp_ratio, qkv_bias, norm_layer, init_scale) for _ in range(depth)])
        self.blocks_token_only = nn.ModuleList([
            LayerScaleBlockCA(embed_dim, num_heads, mlp_ratio_clstk, qkv_bias, norm_layer, init_scale)
            for _ in range(depth_token_only)])
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

    def forward_features(self, x):
        B = x.shape[0]
        x = self.patch_embed(x) + self.pos_embed
        cls = self.cls_token.expand(B, -1, -1)
        for blk in self.blocks:
            x = blk(x)
        for blk in self.blocks_token_only:
            cls = blk(x, cls)
        return self.norm(torch.cat((cls, x), dim=1))[:, 0]

    def forward(self, x):
        return self.head(self.forward_features(x))


VARIANTS = {
    # name: (img, embed_dim, depth, heads, init_scale)   models/cait.py:255-480
    "cait_XXS24_224": (224, 192, 24, 4, 1e-5), "cait_XXS24": (384, 192, 24, 4, 1e-5),
    "cait_XXS36_224": (224, 192, 36, 4, 1e-5), "cait_XXS36": (384, 192, 36, 4, 1e-5),
    "cait_XS24": (384, 288, 24, 6, 1e-5), "cait_S24_224": (224, 384, 24, 8, 1e-5),
    "cait_S24": (384, 384, 24, 8, 1e-5), "cait_S36": (384, 384, 36, 8, 1e-6),
    "cait_M36": (384, 768, 36, 16, 1e-6), "cait_M48": (448, 768, 48, 16, 1e-6),
}


def build(arch, num_classes=1000, **kw):
    img, d, depth, heads, scale = VARIANTS[arch]
    return CaiT(img_size=img, patch_size=16, embed_dim=d, depth=depth, num_heads=heads, mlp_ratio=4,
                qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), init_scale=scale,
                depth_token_only=2, num_classes=num_classes, **kw)
