"""Host logic of the bicubic pos_embed resize (vit_torch_amd/posembed.py): the tap tables against
torch's own F.interpolate (the call upstream DINO makes: oracle/vit_ref.py:110-127), forward and
transpose.  The HIP kernel that applies them is checked in tests/test_posembed_gpu.py."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.vit_ref import interpolate_pos_encoding
from vit_torch_amd.posembed import axis_taps, resize_tables


def _apply(csr, src, n):
    rp, c, w = csr
    out = torch.zeros(n, src.shape[1], dtype=torch.float64)
    for r in range(n):
        for e in range(rp[r], rp[r + 1]):
            out[r] += float(w[e]) * src[c[e]].double()
    return out.float()


@pytest.mark.parametrize("side,gh,gw,D", [(14, 2, 2, 32), (28, 12, 12, 16), (14, 37, 37, 8), (14, 9, 5, 8), (7, 3, 14, 8)])
def test_tables_match_interpolate(side, gh, gw, D):
    g = torch.Generator().manual_seed(side * 100 + gh)
    pos = torch.randn(1, 1 + side * side, D, generator=g)
    leaf = pos.clone().requires_grad_(True)
    patch = leaf[:, 1:].reshape(1, side, side, D).permute(0, 3, 1, 2)
    ref = F.interpolate(patch, scale_factor=((gh + 0.1) / side, (gw + 0.1) / side), mode="bicubic")
    assert tuple(ref.shape[-2:]) == (gh, gw)
    eff = torch.cat((leaf[:, :1], ref.permute(0, 2, 3, 1).reshape(1, -1, D)), 1)[0]
    fwd, bwd = resize_tables(side, gh, gw)
    out = _apply(fwd, pos[0], 1 + gh * gw)
    assert (out - eff.detach()).abs().max() <= 1e-5 * eff.detach().abs().max()
    gout = torch.randn(eff.shape, generator=g)
    (gl,) = torch.autograd.grad(eff, leaf, gout)
    gin = _apply(bwd, gout, 1 + side * side)
    assert (gin - gl[0]).abs().max() <= 1e-5 * gl.abs().max()


def test_tables_match_oracle_function():
    """The oracle's restatement of upstream's call (square inputs: configs C1 32x32 and C3 96x96)."""
    for side, img, p in [(14, 32, 16), (28, 96, 8)]:
        D = 8
        pos = torch.randn(1, 1 + side * side, D, generator=torch.Generator().manual_seed(img))
        g = img // p
        ref = interpolate_pos_encoding(pos, g * g, img, img, p)[0]
        fwd, _ = resize_tables(side, g, g)
        out = _apply(fwd, pos[0], 1 + g * g)
        assert (out - ref).abs().max() <= 1e-5 * ref.abs().max()


def test_axis_taps_partition_of_unity_and_bounds():
    idx, w = axis_taps(14, 37, 37.1 / 14)
    assert idx.min() >= 0 and idx.max() <= 13 and idx.dtype == np.int32
    np.testing.assert_allclose(w.sum(1), 1.0, atol=1e-6)


def test_csr_rows_are_sorted_and_complete():
    (rp, c, w), (rpt, ct, wt) = resize_tables(14, 2, 2)
    assert rp[0] == 0 and rp[-1] == len(c) == len(w) == 1 + 16 * 4
    assert rpt[-1] == len(ct) and len(rpt) == 1 + 14 * 14 + 1
    for r in range(len(rp) - 1):
        assert list(c[rp[r]:rp[r + 1]]) == sorted(c[rp[r]:rp[r + 1]])
    np.testing.assert_allclose(w.sum(), 5.0, atol=1e-5)          # every output row sums to one
