"""nn.CrossEntropyLoss() replacement (main.py:244, utils_network.py:430) on the
fused softmax-cross-entropy HIP kernel: one launch computes the mean loss, the
logits gradient and the argmax==label count (utils_network.py:85-95)."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops


class _XentFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, owner):
        if labels.dtype != torch.int64 or labels.dim() != 1 or labels.shape[0] != logits.shape[0]:
            raise TypeError(f"CrossEntropyLoss: labels must be int64 [B] class indices, got {labels.dtype} "
                            f"{tuple(labels.shape)} for logits {tuple(logits.shape)}")
        B, K = logits.shape
        lg = logits.contiguous().float()
        loss_buf = torch.empty(1 + B, dtype=torch.float32, device=lg.device)
        correct = torch.empty(1 + B, dtype=torch.int32, device=lg.device)
        dlogits = torch.empty_like(lg)
        ops.softmax_xent(lg, labels.contiguous(), loss_buf, dlogits, correct)
        ctx.save_for_backward(dlogits)
        owner.last_correct = correct[0]          # device scalar, no sync
        owner.last_per_sample = loss_buf[1:]
        return loss_buf[0]

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        # d loss -> d logits: the upstream gradient is a device scalar (1.0 for `loss.backward()`); the product runs on
        # vitmi_scale_cast (the [B, K] gradient as one row, the scalar as that row's factor) so that no ATen math is left
        # inside the step; a B*K that is not a multiple of 4 keeps the ATen product
        n = dlogits.numel()
        if n % 4 or not dlogits.is_cuda:
            return dlogits * g, None, None
        out = torch.empty_like(dlogits)
        ops.scale_cast(dlogits, out, None, M=1, N=n, rowscale=g.detach().reshape(1).float(), rows_per_group=1)
        return out, None, None


class CrossEntropyLoss(nn.Module):
    """Mean-reduced cross entropy over int64 class labels."""

    def __init__(self):
        super().__init__()
        self.last_correct = None
        self.last_per_sample = None

    def forward(self, logits, labels):
        return _XentFn.apply(logits, labels, self)
