"""GPU legs of SURVEY.md §8(f) rows 1 and 3.

(f)1  reference-format checkpoints written from the oracle models are loaded with
      `load_reference_checkpoint` (CaiT `module.`-prefixed strict load, models/cait.py:377-385;
      Swin `checkpoint['model']` non-strict, models/swin.py:831-840; DINO state dict holding a
      pos_embed of another pretraining grid, models/vision_all.py:156) and the HIP forward must
      reproduce the oracle's logits, before AND after a first forward (live engine).
(f)3  `Network.fit(log=RunLog(...))` for two tiny epochs on the GPU: the JSON it writes has the
      key layout of the reference's own log (tests/golden/ref_stats_log.json, a copy of
      /root/reference/logs/massA/stats_210715_212442.json)."""
import json
import os
from functools import partial

import pytest
import torch

from util import assert_close

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def x_for(S, B=3, seed=0):
    return torch.randn(B, 3, S, S, generator=torch.Generator("cpu").manual_seed(seed))


def test_cait_checkpoint_to_hip_forward(tmp_path):
    from oracle import cait_ref
    from oracle.vit_ref import seeded_init_
    from vit_torch_amd import load_reference_checkpoint
    from vit_torch_amd.cait import cait_models
    cfg = dict(img_size=32, patch_size=8, embed_dim=64, depth=2, num_heads=4, qkv_bias=True, init_scale=1e-1,
               depth_token_only=2, num_classes=10)
    norm = partial(torch.nn.LayerNorm, eps=1e-6)
    refs = [seeded_init_(cait_ref.CaiT(**cfg, norm_layer=norm), s) for s in (3, 4)]
    m = cait_models(**cfg, norm_layer=norm, compute_dtype="fp32").cuda()
    x = x_for(32)
    for i, ref in enumerate(refs):          # second load lands on a LIVE engine
        path = tmp_path / f"cait{i}.pth"
        torch.save({"model": {"module." + k: v for k, v in ref.state_dict().items()}}, path)
        res = load_reference_checkpoint(m, str(path), "cait")
        assert not res.missing_keys and not res.unexpected_keys
        with torch.no_grad():
            assert_close(f"cait logits after load {i}", m(x.cuda()), ref(x), 1e-4)


@pytest.mark.parametrize("compute,tol", [("fp32", 1e-4), ("bf16", 1.2e-2)])
def test_swin_checkpoint_to_hip_forward(tmp_path, compute, tol):
    from oracle import swin_ref
    from oracle.vit_ref import seeded_init_
    from vit_torch_amd import SwinTransformer, load_reference_checkpoint
    cfg = dict(img_size=56, patch_size=4, in_chans=3, num_classes=10, embed_dim=32, depths=[2, 2], num_heads=[2, 4],
               window_size=7, drop_path_rate=0.0)
    m = SwinTransformer(**cfg, compute_dtype=compute, residual_dtype="auto").cuda()
    x = x_for(56)
    for i, seed in enumerate((5, 6)):
        ref = seeded_init_(swin_ref.SwinTransformer(**cfg), seed)
        sd = dict(ref.state_dict())
        sd.pop("layers.0.blocks.1.attn_mask")              # buffers may be absent (strict=False)
        path = tmp_path / f"swin{i}.pth"
        torch.save({"model": sd}, path)
        res = load_reference_checkpoint(m, str(path), "swin")
        assert res.missing_keys == ["layers.0.blocks.1.attn_mask"]
        with torch.no_grad():
            assert_close(f"swin logits after load {i} [{compute}]", m(x.cuda()), ref(x), tol)


def test_dino_checkpoint_with_another_pretraining_grid_to_hip_forward(tmp_path):
    """A backbone pretrained at 64x64 (4x4 grid) loaded into a module built for 32x32: the stored
    table is kept and resized per input (bicubic), logits must match the oracle at both sizes."""
    from oracle import vit_ref
    from vit_torch_amd import VisionTransformer, load_reference_checkpoint
    big = vit_ref.VisionTransformer(img_size=64, patch_size=16, embed_dim=64, depth=2, num_heads=2)
    vit_ref.seeded_init_(big, 7)
    path = tmp_path / "dino.pth"
    torch.save({"module.backbone." + k: v for k, v in big.state_dict().items()}, path)
    m = VisionTransformer(img_size=32, patch_size=16, embed_dim=64, depth=2, num_heads=2, compute_dtype="fp32").cuda()
    with torch.no_grad():
        m(x_for(32).cuda())                                # engine alive before the load
    res = load_reference_checkpoint(m, str(path), "dino")
    assert not res.unexpected_keys
    m = m.cuda()
    for S in (32, 64, 48):
        x = x_for(S)
        with torch.no_grad():
            assert_close(f"dino features @{S}", m(x.cuda()), big(x), 1e-4)


def test_network_fit_writes_the_reference_log_layout_on_the_gpu(tmp_path):
    from vit_torch_amd import VisionModelZoo, VisionTransformer
    from vit_torch_amd.network import Network
    from vit_torch_amd.stats import RunLog, probe_hardware
    ref_log = json.load(open(os.path.join(HERE, "golden", "ref_stats_log.json")))
    m = VisionTransformer(img_size=32, patch_size=8, embed_dim=64, depth=2, num_heads=2, apply_head=True,
                          compute_dtype="bf16", residual_dtype="auto")
    m.head = VisionModelZoo.get_classifier_head(64, 10)
    g = torch.Generator("cpu").manual_seed(0)
    train = [(torch.randn(8, 3, 32, 32, generator=g), torch.randint(0, 10, (8,), generator=g)) for _ in range(3)]
    val = [(torch.randn(8, 3, 32, 32, generator=g), torch.randint(0, 10, (8,), generator=g)) for _ in range(2)]
    path = str(tmp_path / "logs" / "stats_run.json")
    log = RunLog(path=path, info=dict(ref_log["info"], arch="tiny_vit"),
                 telem={"sample_count_train": 24, "sample_count_val": 16, "mode": ref_log["telem"]["mode"],
                        "hardware": probe_hardware()})
    net = Network(m, opt="sgd", lr=0.05, lr_type="step", lr_step=1, lr_gamma=0.5, device="cuda")
    net.fit(train, val, epochs=2, log=log)
    got = json.load(open(path))
    assert list(got) == list(ref_log)                                  # info, telem, results, train, val
    assert set(got["results"]) == set(ref_log["results"])
    assert set(ref_log["telem"]) <= set(got["telem"]) or set(got["telem"]) == set(ref_log["telem"])
    for split, n in (("train", 24), ("val", 16)):
        assert len(got[split]) == 2
        for rec in got[split]:
            assert set(rec) == set(ref_log[split][0])
            assert rec["sample"] == n and rec["time_finish"] >= rec["time_start"]
            assert 0.0 <= rec["acc"] <= 1.0 and rec["loss"] > 0
    assert got["train"][0]["lr"] == pytest.approx(0.05) and got["train"][1]["lr"] == pytest.approx(0.025)
    assert got["val"][0]["lr"] == 0.0
    assert got["telem"]["completed"] is True
    assert got["telem"]["hardware"] == probe_hardware() and "3090" not in got["telem"]["hardware"]
