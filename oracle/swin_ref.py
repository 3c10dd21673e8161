"""ORACLE (test infrastructure only — never imported by vit_torch_amd).

Plain-PyTorch fp32 restatement of the reference's Swin Transformer,
/root/reference/models/swin.py: window_partition/reverse :33-62, WindowAttention :65-144,
SwinTransformerBlock :163-270, PatchMerging :291-328, BasicLayer :340-396, PatchEmbed
:410-448, SwinTransformer :458-591, configs :768-820.  Parameter / buffer names follow the
reference state_dict (layers.{i}.blocks.{j}.attn.{relative_position_bias_table,
relative_position_index,qkv,proj}, ...attn_mask, layers.{i}.downsample.{reduction,norm}, ...).
DropPath takes an explicit per-sample keep mask so parity can pin it.
Pinned by tests/golden/{window_attention,patch_merging,swin_tiny}.npz.
"""
import torch
import torch.nn as nn

from .vit_ref import Mlp


def window_partition(x, ws):
    B, H, W, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C)


def window_reverse(windows, ws, H, W):
    B = int(windows.shape[0] / (H * W / ws / ws))
    x = windows.view(B, H // ws, W // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


def relative_position_index(ws):
    coords = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij"))
    cf = torch.flatten(coords, 1)
    rel = (cf[:, :, None] - cf[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def shift_attn_mask(H, W, ws, shift):
    """0 / -100 mask of the shifted-window blocks (models/swin.py:208-229)."""
    img = torch.zeros((1, H, W, 1))
    cnt = 0
    for h in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for w in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, h, w, :] = cnt
            cnt += 1
    mw = window_partition(img, ws).view(-1, ws * ws)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return am.masked_fill(am != 0, float(-100.0)).masked_fill(am == 0, float(0.0))


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=True):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        self.scale = (dim // num_heads) ** -0.5
        ws = window_size[0]
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) * (2 * ws - 1), num_heads))
        self.register_buffer("relative_position_index", relative_position_index(ws))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)

    def forward(self, x, mask=None):
        B_, N, C = x.shape
        H = self.num_heads
        qkv = self.qkv(x).reshape(B_, N, 3, H, C // H).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * self.scale, qkv[1], qkv[2]
        attn = q @ k.transpose(-2, -1)
        bias = self.relative_position_bias_table[self.relative_position_index.view(-1)].view(N, N, -1)
        attn = attn + bias.permute(2, 0, 1).contiguous().unsqueeze(0)
        if mask is not None:
            nW = mask.shape[0]
            attn = attn.view(B_ // nW, nW, H, N, N) + mask.unsqueeze(1).unsqueeze(0)
            attn = attn.view(-1, H, N, N)
        attn = attn.softmax(dim=-1)
        return self.proj((attn @ v).transpose(1, 2).reshape(B_, N, C))


def drop_path(x, keep_mask, keep_prob):
    """timm DropPath with an explicit per-sample mask: x / keep_prob * mask[b]."""
    if keep_mask is None:
        return x
    return x / keep_prob * keep_mask.view(-1, *([1] * (x.ndim - 1)))


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, input_resolution, num_heads, window_size=7, shift_size=0, mlp_ratio=4.0,
                 qkv_bias=True, drop_path=0.0, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim, self.input_resolution, self.num_heads = dim, input_resolution, num_heads
        self.window_size, self.shift_size = window_size, shift_size
        if min(input_resolution) <= window_size:
            self.shift_size = 0
            self.window_size = min(input_resolution)
        self.drop_path_rate = drop_path
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, (self.window_size, self.window_size), num_heads, qkv_bias)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio))
        if self.shift_size > 0:
            H, W = input_resolution
            attn_mask = shift_attn_mask(H, W, self.window_size, self.shift_size)
        else:
            attn_mask = None
        self.register_buffer("attn_mask", attn_mask)

    def forward(self, x, keep_masks=None):
        """keep_masks: None or (mask_attn[B], mask_mlp[B]) Bernoulli(1-rate) draws."""
        H, W = self.input_resolution
        B, L, C = x.shape
        ws, s = self.window_size, self.shift_size
        shortcut = x
        x = self.norm1(x).view(B, H, W, C)
        if s > 0:
            x = torch.roll(x, shifts=(-s, -s), dims=(1, 2))
        xw = window_partition(x, ws).view(-1, ws * ws, C)
        aw = self.attn(xw, mask=self.attn_mask).view(-1, ws, ws, C)
        x = window_reverse(aw, ws, H, W)
        if s > 0:
            x = torch.roll(x, shifts=(s, s), dims=(1, 2))
        x = x.view(B, H * W, C)
        kp = 1.0 - self.drop_path_rate
        m1, m2 = keep_masks if keep_masks is not None else (None, None)
        x = shortcut + drop_path(x, m1, kp)
        return x + drop_path(self.mlp(self.norm2(x)), m2, kp)


class PatchMerging(nn.Module):
    def __init__(self, input_resolution, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.input_resolution, self.dim = input_resolution, dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = norm_layer(4 * dim)

    def forward(self, x):
        H, W = self.input_resolution
        B, L, C = x.shape
        x = x.view(B, H, W, C)
        x = torch.cat([x[:, 0::2, 0::2, :], x[:, 1::2, 0::2, :], x[:, 0::2, 1::2, :], x[:, 1::2, 1::2, :]], -1)
        return self.reduction(self.norm(x.view(B, -1, 4 * C)))


class BasicLayer(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio=4.0, qkv_bias=True,
                 drop_path=0.0, norm_layer=nn.LayerNorm, downsample=None):
        super().__init__()
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, input_resolution, num_heads, window_size,
                                 0 if i % 2 == 0 else window_size // 2, mlp_ratio, qkv_bias,
                                 drop_path[i] if isinstance(drop_path, list) else drop_path, norm_layer)
            for i in range(depth)])
        self.downsample = downsample(input_resolution, dim=dim, norm_layer=norm_layer) if downsample else None

    def forward(self, x, keep_masks=None):
        for i, blk in enumerate(self.blocks):
            x = blk(x, keep_masks[i] if keep_masks is not None else None)
        return self.downsample(x) if self.downsample is not None else x


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=4, in_chans=3, embed_dim=96, norm_layer=None):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.patches_resolution = [img_size // patch_size, img_size // patch_size]
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None

    def forward(self, x):
        x = self.proj(x).flatten(2).transpose(1, 2)
        return self.norm(x) if self.norm is not None else x


class SwinTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=4, in_chans=3, num_classes=1000, embed_dim=96,
                 depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), window_size=7, mlp_ratio=4.0, qkv_bias=True,
                 drop_path_rate=0.1, norm_layer=nn.LayerNorm, patch_norm=True, **_ignored):
        super().__init__()
        self.num_layers = len(depths)
        self.embed_dim = embed_dim
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim, norm_layer if patch_norm else None)
        pr = self.patch_embed.patches_resolution
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(BasicLayer(int(embed_dim * 2 ** i), (pr[0] // 2 ** i, pr[1] // 2 ** i), depths[i],
                                          num_heads[i], window_size, mlp_ratio, qkv_bias,
                                          dpr[sum(depths[:i]):sum(depths[:i + 1])], norm_layer,
                                          PatchMerging if i < self.num_layers - 1 else None))
        self.norm = norm_layer(self.num_features)
        self.avgpool = nn.AdaptiveAvgPool1d(1)
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
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

    def forward_features(self, x, keep_masks=None):
        x = self.patch_embed(x)
        for i, layer in enumerate(self.layers):
            x = layer(x, keep_masks[i] if keep_masks is not None else None)
        x = self.norm(x)
        return torch.flatten(self.avgpool(x.transpose(1, 2)), 1)

    def forward(self, x, keep_masks=None):
        return self.head(self.forward_features(x, keep_masks))


CONFIGS = {   # models/swin.py:768-820 (window 7, 224 variants; prefix match as :823-826)
    "swin_tiny_patch4_window7_224": dict(embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], drop_path_rate=0.2),
    "swin_small_patch4_window7_224": dict(embed_dim=96, depths=[2, 2, 18, 2], num_heads=[3, 6, 12, 24], drop_path_rate=0.3),
    "swin_base_patch4_window7_224": dict(embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], drop_path_rate=0.5),
}


def build(arch, num_classes=1000, **kw):
    cfg = dict(CONFIGS[arch])
    cfg.update(kw)
    return SwinTransformer(img_size=224, patch_size=4, window_size=7, num_classes=num_classes, **cfg)
