"""DropPath / to_2tuple / trunc_normal_ as used by models/swin.py:11, models/cait.py:10."""
import torch
import torch.nn as nn


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


trunc_normal_ = nn.init.trunc_normal_


class DropPath(nn.Module):
    """Per-sample stochastic depth: identity when drop_prob == 0 or not training, else
    x / keep * Bernoulli(keep) with one draw per sample."""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob or 0.0

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x / keep * mask
