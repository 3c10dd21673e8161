"""Device-side input transform against a torch restatement of the reference's pipeline
(utils_datasets.py:553-582) with the same per-sample draws: bit-exact."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _reference(img_u8_nhwc, oy, ox, flip, mean, std, S, pad, fill=128):
    out = []
    for b in range(img_u8_nhwc.shape[0]):
        im = img_u8_nhwc[b].permute(2, 0, 1)                                  # CHW uint8
        im = F.pad(im, (pad, pad, pad, pad), value=fill)                       # RandomCrop padding, fill=128
        im = im[:, oy[b]:oy[b] + S, ox[b]:ox[b] + S]
        if flip[b]:
            im = im.flip(-1)                                                   # RandomHorizontalFlip
        x = im.float().div(255)                                                # ToTensor
        out.append((x - mean[:, None, None]) / std[:, None, None])             # Normalize
    return torch.stack(out)


@pytest.mark.parametrize("H,S,name", [(32, 32, "cifar10"), (96, 96, "stl10"), (40, 32, "cifar10")])
def test_train_transform_is_bit_exact(lib, H, S, name):
    from vit_torch_amd.data import NORM, DeviceAugment
    g = torch.Generator("cpu").manual_seed(3)
    B = 37
    img = torch.randint(0, 256, (B, H, H, 3), generator=g, dtype=torch.uint8)
    aug = DeviceAugment(S, **NORM[name], train=True, generator=torch.Generator("cpu").manual_seed(5))
    oy, ox, fl = aug.draw(B, H, H)
    assert int(oy.max()) <= H + 2 * aug.pad - S and int(fl.sum()) not in (0, B)
    got = aug(img.cuda(), oy, ox, fl)
    mean, std = torch.tensor(NORM[name]["mean"]), torch.tensor(NORM[name]["std"])
    want = _reference(img, oy.cpu().tolist(), ox.cpu().tolist(), fl.cpu().tolist(), mean, std, S, aug.pad)
    assert got.shape == (B, 3, S, S)
    assert torch.equal(got.cpu(), want), (got.cpu() - want).abs().max()


def test_test_transform_and_model_consumption(lib):
    from vit_torch_amd import VisionTransformer
    from vit_torch_amd.data import NORM, DeviceAugment
    g = torch.Generator("cpu").manual_seed(4)
    img = torch.randint(0, 256, (8, 32, 32, 3), generator=g, dtype=torch.uint8)
    aug = DeviceAugment(32, **NORM["cifar10"], train=False)
    x = aug(img)
    mean, std = torch.tensor(NORM["cifar10"]["mean"]), torch.tensor(NORM["cifar10"]["std"])
    want = (img.permute(0, 3, 1, 2).float().div(255) - mean[:, None, None]) / std[:, None, None]
    assert torch.equal(x.cpu(), want)
    m = VisionTransformer(img_size=32, patch_size=8, embed_dim=64, depth=1, num_heads=2, num_classes=10,
                          apply_head=True, compute_dtype="fp32").cuda()
    with torch.no_grad():
        assert m(x).shape == (8, 10)


@pytest.mark.parametrize("H,S,p,dt", [(32, 32, 16, torch.bfloat16), (96, 96, 8, torch.bfloat16), (40, 32, 8, torch.float32)])
def test_fused_ingest_patchify_equals_the_two_calls_bit_for_bit(lib, H, S, p, dt):
    """uint8 NHWC -> patch rows in one pass (vitmi_ingest_patchify) against vitmi_image_ingest followed by
    vitmi_patchify on the same draws: same values, same rounding, CLS placeholder rows zero."""
    from vit_torch_amd import ops
    from vit_torch_amd.data import NORM, DeviceAugment
    g = torch.Generator("cpu").manual_seed(11)
    B = 9
    img = torch.randint(0, 256, (B, H, H, 3), generator=g, dtype=torch.uint8).cuda()
    aug = DeviceAugment(S, **NORM["stl10"], train=True, generator=torch.Generator("cpu").manual_seed(6))
    oy, ox, fl = aug.draw(B, H, H)
    x = aug(img, oy, ox, fl)
    gq = S // p
    want = torch.full((B * (1 + gq * gq), 3 * p * p), float("nan"), device="cuda").to(dt)
    ops.patchify(x, want, p, cls_rows=1)
    rows = aug.patch_rows(img, p, dtype=dt, off_y=oy, off_x=ox, flip=fl)
    assert rows.rows.shape == want.shape and torch.equal(rows.rows.float().cpu(), want.float().cpu())
    assert rows.rows.view(B, 1 + gq * gq, -1)[:, 0].abs().max().item() == 0


def test_model_takes_patch_rows_in_place_of_the_image_tensor(lib):
    from vit_torch_amd import CrossEntropyLoss, VisionTransformer
    from vit_torch_amd.data import NORM, DeviceAugment
    g = torch.Generator("cpu").manual_seed(12)
    img = torch.randint(0, 256, (8, 32, 32, 3), generator=g, dtype=torch.uint8).cuda()
    y = torch.randint(0, 10, (8,), generator=g).cuda()
    aug = DeviceAugment(32, **NORM["cifar10"], train=True, generator=torch.Generator("cpu").manual_seed(7))
    oy, ox, fl = aug.draw(8, 32, 32)
    torch.manual_seed(2)
    m = VisionTransformer(img_size=32, patch_size=8, embed_dim=64, depth=2, num_heads=2, num_classes=10,
                          apply_head=True, compute_dtype="bf16", residual_dtype="auto").cuda()
    outs, grads = [], []
    for inp in (aug(img, oy, ox, fl), aug.patch_rows(img, 8, off_y=oy, off_x=ox, flip=fl)):
        m.zero_grad()
        out = m(inp)
        CrossEntropyLoss()(out, y).backward()
        outs.append(out.detach().clone()); grads.append(m.patch_embed.proj.weight.grad.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(grads[0], grads[1])
    with pytest.raises(Exception):
        m(aug.patch_rows(img, 16, off_y=oy, off_x=ox, flip=fl))          # rows of another patch size are rejected
