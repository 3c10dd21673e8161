"""Tap tables of the bicubic position-embedding resize.

Upstream DINO's `interpolate_pos_encoding` (the module the reference loads through
`torch.hub.load`, /root/reference/models/vision_all.py:156; restated in
oracle/vit_ref.py:110-127) resizes the stored `side x side` grid of `pos_embed` to the
input's patch grid with `F.interpolate(scale_factor=((gh+0.1)/side, (gw+0.1)/side),
mode="bicubic")`.  The weights depend only on the two grids: this module restates torch's
coordinate and coefficient arithmetic (aten UpSampleBicubic2d: align_corners=False, the
caller's scale factor, A = -0.75, border taps clamped) in fp32 numpy and emits the resize
— and its transpose, which is its backward — as CSR tables for `vitmi_pos_resample`.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch

_A = np.float32(-0.75)


def _cubic1(x):            # |x| <= 1
    return ((_A + np.float32(2)) * x - (_A + np.float32(3))) * x * x + np.float32(1)


def _cubic2(x):            # 1 < |x| < 2
    return ((_A * x - np.float32(5) * _A) * x + np.float32(8) * _A) * x - np.float32(4) * _A


def axis_taps(n_in: int, n_out: int, scale_factor: float) -> Tuple[np.ndarray, np.ndarray]:
    """(index [n_out, 4] int32, weight [n_out, 4] fp32) of one axis, as aten computes them:
    src = (dst + 0.5) / scale_factor - 0.5 (no clamp for cubic), taps floor(src) - 1 .. + 2
    clamped into [0, n_in - 1]."""
    scale = np.float32(1.0 / float(scale_factor))              # area_pixel_compute_scale: double, then cast
    dst = np.arange(n_out, dtype=np.float32)
    real = scale * (dst + np.float32(0.5)) - np.float32(0.5)
    base = np.floor(real)
    t = np.clip(real - base, np.float32(0), np.float32(1)).astype(np.float32)
    w = np.stack([_cubic2(t + np.float32(1)), _cubic1(t), _cubic1(np.float32(1) - t),
                  _cubic2(np.float32(2) - t)], axis=1).astype(np.float32)
    idx = np.clip(base.astype(np.int64)[:, None] + np.arange(-1, 3)[None, :], 0, n_in - 1).astype(np.int32)
    return idx, w


def resize_tables(side: int, gh: int, gw: int):
    """CSR (row_ptr, col, w) of the map stored rows [1 + side*side] -> effective rows
    [1 + gh*gw] (row 0, the CLS position, is copied), and of its transpose."""
    iy, wy = axis_taps(side, gh, (gh + 0.1) / side)
    ix, wx = axis_taps(side, gw, (gw + 0.1) / side)
    n_out, n_in = 1 + gh * gw, 1 + side * side
    rows, cols, vals = [0], [0], [np.float32(1)]
    for oy in range(gh):
        for ox in range(gw):
            r = 1 + oy * gw + ox
            for a in range(4):
                for b in range(4):
                    rows.append(r)
                    cols.append(1 + int(iy[oy, a]) * side + int(ix[ox, b]))
                    vals.append(np.float32(wy[oy, a] * wx[ox, b]))
    rows, cols = np.asarray(rows, np.int64), np.asarray(cols, np.int64)
    vals = np.asarray(vals, np.float32)

    def csr(r, c, v, n):
        order = np.lexsort((c, r))                       # fixed entry order inside a row
        r, c, v = r[order], c[order], v[order]
        ptr = np.zeros(n + 1, np.int32)
        np.add.at(ptr, r + 1, 1)
        return np.cumsum(ptr).astype(np.int32), c.astype(np.int32), v

    return csr(rows, cols, vals, n_out), csr(cols, rows, vals, n_in)


class PosResize:
    """Device-resident tables of one (side, gh, gw); `fwd` / `bwd` = (row_ptr, col, w, rows)."""

    def __init__(self, side: int, gh: int, gw: int, device):
        f, b = resize_tables(side, gh, gw)
        dev = torch.device(device)
        self.fwd = tuple(torch.from_numpy(a).to(dev) for a in f) + (1 + gh * gw,)
        self.bwd = tuple(torch.from_numpy(a).to(dev) for a in b) + (1 + side * side,)


_CACHE: Dict[tuple, PosResize] = {}


def tables_for(n_stored: int, gh: int, gw: int, device) -> PosResize:
    side = int(math.sqrt(n_stored))
    if side * side != n_stored:
        raise ValueError(f"pos_embed holds {n_stored} patch positions: not a square grid")
    key = (side, gh, gw, str(device))
    t = _CACHE.get(key)
    if t is None:
        t = _CACHE[key] = PosResize(side, gh, gw, device)
    return t
