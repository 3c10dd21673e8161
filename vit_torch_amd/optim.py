"""optim.SGD(momentum=0.9) (utils_network.py:120, step at :442) as ONE fused HIP
kernel per model over the flat parameter / gradient / momentum buffers, which
also refreshes the bf16 weight shadow in the same pass."""
from __future__ import annotations

import torch

from . import ops
from ._lib import VitmiError


class FusedSGD(torch.optim.Optimizer):
    """Same update rule as torch.optim.SGD(lr, momentum, dampening=0, nesterov=False,
    weight_decay=0): buf = momentum*buf + g (buf starts at 0, i.e. buf_1 = g_1);
    p -= lr*buf.  `grad_scale` multiplies g first (1/world_size after a SUM
    all-reduce)."""

    def __init__(self, params, lr=1e-3, momentum=0.9, grad_scale=1.0):
        defaults = dict(lr=lr, momentum=momentum, grad_scale=grad_scale)
        super().__init__(params, defaults)
        self._mom = {}

    def _packs(self, group):
        packs = []
        for p in group["params"]:
            pack = getattr(p, "_vitmi_pack", None)
            if pack is None or not pack.is_current():
                raise VitmiError("FusedSGD needs parameters that live in a vit_torch_amd ParamPack "
                                 "(run one forward on the GPU first, or call model.engine())")
            if all(pack is not q for q in packs):
                packs.append(pack)
        return packs

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            for pack in self._packs(group):
                buf = self._mom.get(id(pack))
                if buf is None or buf.numel() != pack.total:
                    buf = torch.zeros_like(pack.flat)
                    self._mom[id(pack)] = buf
                ops.sgd_momentum(pack.flat, pack.grad, buf, pack.shadow, group["lr"],
                                 group["momentum"], group["grad_scale"])
                if pack.shadow is not None:
                    pack.mark_shadow_current()     # the kernel wrote master and shadow together
        return loss


class FusedAdamW(torch.optim.Optimizer):
    """`optim.AdamW(params, lr)` / `optim.Adam(params, lr)` of the reference's optimizer table
    (/root/reference/utils_network.py:121,124; every linear-evaluation log uses AdamW) as ONE
    kernel over the flat buffers of a ParamPack, with the bf16 shadow refreshed in the same
    pass.  torch's defaults and update order; the step count lives on the device, so the
    step can be replayed from a HIP graph.  `decoupled=False` gives Adam (L2 decay)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, decoupled=True,
                 grad_scale=1.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, decoupled=decoupled,
                        grad_scale=grad_scale)
        super().__init__(params, defaults)
        self._st = {}

    _packs = FusedSGD._packs

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            for pack in self._packs(group):
                st = self._st.get(id(pack))
                if st is None or st[0].numel() != pack.total:
                    st = (torch.zeros_like(pack.flat), torch.zeros_like(pack.flat),
                          torch.zeros(1, dtype=torch.float32, device=pack.flat.device))
                    self._st[id(pack)] = st
                b1, b2 = group["betas"]
                ops.adam(pack.flat, pack.grad, st[0], st[1], pack.shadow, st[2], group["lr"], b1, b2, group["eps"],
                         group["weight_decay"], group["decoupled"], group["grad_scale"])
                if pack.shadow is not None:
                    pack.mark_shadow_current()
        return loss
