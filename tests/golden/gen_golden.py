"""Generate the golden vectors that pin the oracle to the REFERENCE implementation.

Runs only in the build container (needs /root/reference).  It imports the
reference's models/cait.py and models/swin.py unchanged under the timm stand-in
(oracle/timm_shim), runs them on seeded inputs/weights in fp32 on the CPU, and
stores inputs + outputs + gradients as small .npz fixtures next to this file.
No reference source text is stored: fixtures are data only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py
"""
import os
import sys
from functools import partial

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "oracle", "timm_shim"))
sys.path.insert(0, "/root/reference")

from models import cait, swin  # noqa: E402  (the reference's own files)


def rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator("cpu").manual_seed(seed)) * scale


def seeded_(module, seed):
    g = torch.Generator("cpu").manual_seed(seed)
    with torch.no_grad():
        for n, p in module.named_parameters():
            if p.dim() >= 2:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
            elif "norm" in n and n.endswith("weight"):
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g))
            elif "gamma_" in n:
                p.copy_(0.5 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    return module


def dump(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                out[f"{k}/{kk}"] = vv.detach().numpy() if torch.is_tensor(vv) else np.asarray(vv)
        else:
            out[k] = v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path)} bytes, {len(out)} arrays)")


def grads(module):
    return {n: p.grad.clone() for n, p in module.named_parameters() if p.grad is not None}


def run(module, x, dy):
    x = x.clone().requires_grad_(True)
    y = module(x)
    y.backward(dy)
    return y.detach(), x.grad.clone(), grads(module)


# 1. transformer block: cait.LayerScale_Block (models/cait.py:130-150) with the talking-heads
#    mixes set to the identity is the vanilla pre-norm block of the DINO ViT
def vit_block():
    blk = cait.LayerScale_Block(dim=64, num_heads=2, mlp_ratio=4.0, qkv_bias=True,
                                norm_layer=partial(nn.LayerNorm, eps=1e-6), init_values=1.0)
    seeded_(blk, 11)
    with torch.no_grad():
        blk.gamma_1.fill_(1.0); blk.gamma_2.fill_(1.0)
        for m in (blk.attn.proj_l, blk.attn.proj_w):
            m.weight.copy_(torch.eye(2)); m.bias.zero_()
    x, dy = rnd((3, 17, 64), 1), rnd((3, 17, 64), 2)
    y, dx, g = run(blk, x, dy)
    keep = {k: v for k, v in blk.state_dict().items() if "proj_l" not in k and "proj_w" not in k and "gamma" not in k}
    g = {k: v for k, v in g.items() if k in keep}
    dump("vit_block", x=x, dy=dy, y=y, dx=dx, state=keep, grad=g)


# 2. Mlp (models/swin.py:14-30)
def mlp():
    m = seeded_(swin.Mlp(48, 96), 12)
    x, dy = rnd((5, 7, 48), 3), rnd((5, 7, 48), 4)
    y, dx, g = run(m, x, dy)
    dump("mlp", x=x, dy=dy, y=y, dx=dx, state=m.state_dict(), grad=g)


# 3. WindowAttention (models/swin.py:65-144): with bias table, with and without shift mask
def window_attention():
    wa = seeded_(swin.WindowAttention(dim=32, window_size=(7, 7), num_heads=2), 13)
    with torch.no_grad():
        wa.relative_position_bias_table.copy_(rnd(wa.relative_position_bias_table.shape, 14, 0.5))
    x, dy = rnd((8, 49, 32), 5), rnd((8, 49, 32), 6)
    y, dx, g = run(wa, x, dy)
    mask = torch.where(rnd((4, 49, 49), 7) > 0.8, torch.tensor(-100.0), torch.tensor(0.0))
    wa.zero_grad()
    xr = x.clone().requires_grad_(True)
    ym = wa(xr, mask)
    ym.backward(dy)
    st = {k: v for k, v in wa.state_dict().items()}
    dump("window_attention", x=x, dy=dy, y=y, dx=dx, state=st, grad=g, mask=mask, y_masked=ym.detach(),
         dx_masked=xr.grad, grad_masked=grads(wa))
    # the same module with a zero bias table IS vanilla multi-head attention (q scaled first)
    with torch.no_grad():
        wa.relative_position_bias_table.zero_()
    wa.zero_grad()
    y0, dx0, g0 = run(wa, x, dy)
    dump("mhsa_from_window_attention", x=x, dy=dy, y=y0, dx=dx0,
         state={k: v for k, v in wa.state_dict().items() if k.startswith(("qkv", "proj"))},
         grad={k: v for k, v in g0.items() if k.startswith(("qkv", "proj"))})


# 4. talking-heads and class attention, LayerScale blocks (models/cait.py:21-150)
def cait_ops():
    th = seeded_(cait.Attention_talking_head(dim=48, num_heads=4, qkv_bias=True), 15)   # hd = 12
    x, dy = rnd((3, 10, 48), 8), rnd((3, 10, 48), 9)
    y, dx, g = run(th, x, dy)
    dump("talking_heads", x=x, dy=dy, y=y, dx=dx, state=th.state_dict(), grad=g)
    ca = seeded_(cait.Class_Attention(dim=48, num_heads=4, qkv_bias=True), 16)
    u, dyc = rnd((3, 11, 48), 10), rnd((3, 1, 48), 11)
    y, du, g = run(ca, u, dyc)
    dump("class_attention", x=u, dy=dyc, y=y, dx=du, state=ca.state_dict(), grad=g)
    blk = seeded_(cait.LayerScale_Block(dim=48, num_heads=4, qkv_bias=True,
                                        norm_layer=partial(nn.LayerNorm, eps=1e-6), init_values=1e-1), 17)
    y, dx, g = run(blk, x, dy)
    dump("layerscale_block", x=x, dy=dy, y=y, dx=dx, state=blk.state_dict(), grad=g)


# 5. PatchMerging (models/swin.py:291-337)
def patch_merging():
    pm = seeded_(swin.PatchMerging((8, 8), 16), 18)
    x, dy = rnd((2, 64, 16), 12), rnd((2, 16, 32), 13)
    y, dx, g = run(pm, x, dy)
    dump("patch_merging", x=x, dy=dy, y=y, dx=dx, state=pm.state_dict(), grad=g)


# 6. whole tiny models: logits, loss, all gradients
def tiny_models():
    m = cait.cait_models(img_size=32, patch_size=8, embed_dim=64, depth=2, num_heads=4, mlp_ratio=4, qkv_bias=True,
                         norm_layer=partial(nn.LayerNorm, eps=1e-6), init_scale=1e-1, depth_token_only=2, num_classes=10)
    seeded_(m, 19)
    with torch.no_grad():
        m.cls_token.copy_(rnd(m.cls_token.shape, 20, 0.05)); m.pos_embed.copy_(rnd(m.pos_embed.shape, 21, 0.05))
    x = rnd((4, 3, 32, 32), 14)
    y = torch.randint(0, 10, (4,), generator=torch.Generator("cpu").manual_seed(15))
    logits = m(x)
    loss = F.cross_entropy(logits, y)
    loss.backward()
    dump("cait_tiny", x=x, labels=y, logits=logits, loss=loss, state=m.state_dict(), grad=grads(m))

    s = swin.SwinTransformer(img_size=56, patch_size=4, in_chans=3, num_classes=10, embed_dim=32, depths=[2, 2],
                             num_heads=[2, 4], window_size=7, drop_path_rate=0.0)
    seeded_(s, 22)
    x = rnd((2, 3, 56, 56), 16)
    y = torch.randint(0, 10, (2,), generator=torch.Generator("cpu").manual_seed(17))
    logits = s(x)
    loss = F.cross_entropy(logits, y)
    loss.backward()
    st = {k: v for k, v in s.state_dict().items()}
    dump("swin_tiny", x=x, labels=y, logits=logits, loss=loss, state=st, grad=grads(s))


if __name__ == "__main__":
    torch.set_num_threads(4)
    vit_block(); mlp(); window_attention(); cait_ops(); patch_merging(); tiny_models()
