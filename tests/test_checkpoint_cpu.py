"""Reference-format checkpoints: key handling of the three families,
exercised on CPU with checkpoints written from the oracle models."""
import pytest
import torch

from oracle import cait_ref, swin_ref, vit_ref
from oracle.vit_ref import seeded_init_


def test_cait_checkpoint_with_module_prefix_is_loaded_strictly(tmp_path):
    from vit_torch_amd import load_reference_checkpoint
    from vit_torch_amd.cait import cait_models
    from functools import partial
    cfg = dict(img_size=32, patch_size=8, embed_dim=64, depth=2, num_heads=4, qkv_bias=True, init_scale=1e-5,
               depth_token_only=2, num_classes=10)
    ref = cait_ref.CaiT(**cfg, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6))
    seeded_init_(ref, 3)
    path = tmp_path / "cait.pth"
    torch.save({"model": {"module." + k: v for k, v in ref.state_dict().items()}}, path)     # models/cait.py:381-385
    m = cait_models(**cfg)
    res = load_reference_checkpoint(m, path, "cait")
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in ref.state_dict().items():
        assert torch.equal(m.state_dict()[k], v), k
    bad = {"model": {"module." + k: v for k, v in list(ref.state_dict().items())[1:]}}
    with pytest.raises(KeyError):
        load_reference_checkpoint(m, bad, "cait")


def test_swin_checkpoint_is_loaded_non_strictly():
    from vit_torch_amd import SwinTransformer, load_reference_checkpoint
    cfg = dict(img_size=56, patch_size=4, in_chans=3, num_classes=10, embed_dim=32, depths=[2, 2], num_heads=[2, 4],
               window_size=7, drop_path_rate=0.0)
    ref = swin_ref.SwinTransformer(**cfg)
    seeded_init_(ref, 4)
    sd = dict(ref.state_dict())
    sd.pop("layers.0.blocks.1.attn_mask")                 # a buffer may be absent (strict=False, models/swin.py:838)
    sd["some.extra.key"] = torch.zeros(1)
    m = SwinTransformer(**cfg)
    res = load_reference_checkpoint(m, {"model": sd}, "swin")
    assert res.missing_keys == ["layers.0.blocks.1.attn_mask"] and res.unexpected_keys == ["some.extra.key"]
    for k, v in ref.state_dict().items():
        assert torch.equal(m.state_dict()[k], v), k


def test_dino_checkpoint_keeps_a_pos_embed_of_another_grid():
    from vit_torch_amd import VisionTransformer, load_reference_checkpoint
    big = vit_ref.VisionTransformer(img_size=64, patch_size=16, embed_dim=64, depth=2, num_heads=2)      # 4x4 grid
    seeded_init_(big, 5)
    sd = {"module.backbone." + k: v for k, v in big.state_dict().items()}      # a training-wrapper style dump
    m = VisionTransformer(img_size=32, patch_size=16, embed_dim=64, depth=2, num_heads=2)                 # 2x2 grid
    res = load_reference_checkpoint(m, sd, "dino")
    assert not res.unexpected_keys
    assert m.pos_embed.shape == big.pos_embed.shape, "the pretraining table is kept; the engine resizes it per input"
    assert torch.equal(m.blocks[1].mlp.fc2.weight, big.blocks[1].mlp.fc2.weight)
