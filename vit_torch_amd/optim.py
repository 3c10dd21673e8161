"""optim.SGD(momentum=0.9) (utils_network.py:120, step at :442) as ONE fused HIP
kernel per model over the flat parameter / gradient / momentum buffers, which
also refreshes the bf16 weight shadow in the same pass."""
from __future__ import annotations

import weakref

import torch

from . import ops
from ._lib import VitmiError


def _group_runs(group):
    """[(pack, [(start, end), ...])] for one param group: the contiguous element ranges of each
    ParamPack that cover exactly the group's trainable parameters.  A group that holds a whole
    model gives one range = the whole flat buffer (one kernel launch); a subset (head only,
    frozen layers left out, several groups over one model) touches nothing outside it."""
    by_pack = []
    for p in group["params"]:
        pack = getattr(p, "_vitmi_pack", None)
        if pack is None or not pack.is_current():
            raise VitmiError("fused optimizers need parameters that live in a vit_torch_amd ParamPack "
                             "(run one forward on the GPU first, or call model.engine())")
        if not p.requires_grad:
            continue                      # torch.optim skips parameters without a gradient
        for q, ps in by_pack:
            if q is pack:
                ps.append(p)
                break
        else:
            by_pack.append((pack, [p]))
    return [(pack, pack.runs(ps)) for pack, ps in by_pack]


def _check_disjoint(param_groups):
    seen = set()
    for g in param_groups:
        for p in g["params"]:
            if id(p) in seen:
                raise VitmiError("a parameter appears in more than one param group")
            seen.add(id(p))


class FusedSGD(torch.optim.Optimizer):
    """Same update rule as torch.optim.SGD(lr, momentum, dampening=0, nesterov=False,
    weight_decay=0): buf = momentum*buf + g (buf starts at 0, i.e. buf_1 = g_1);
    p -= lr*buf.  `grad_scale` multiplies g first (1/world_size after a SUM
    all-reduce)."""

    def __init__(self, params, lr=1e-3, momentum=0.9, grad_scale=1.0):
        defaults = dict(lr=lr, momentum=momentum, grad_scale=grad_scale)
        super().__init__(params, defaults)
        _check_disjoint(self.param_groups)
        self._mom = weakref.WeakKeyDictionary()     # pack -> momentum buffer (pack layout)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            for pack, runs in _group_runs(group):
                buf = self._mom.get(pack)
                if buf is None:
                    buf = torch.zeros_like(pack.flat)
                    self._mom[pack] = buf
                for s, e in runs:
                    ops.sgd_momentum(pack.flat[s:e], pack.grad[s:e], buf[s:e],
                                     pack.shadow[s:e] if pack.shadow is not None else None,
                                     group["lr"], group["momentum"], group["grad_scale"])
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
        _check_disjoint(self.param_groups)
        self._st = weakref.WeakKeyDictionary()      # pack -> (m, v, {run start: step counter})

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            for pack, runs in _group_runs(group):
                st = self._st.get(pack)
                if st is None:
                    st = (torch.zeros_like(pack.flat), torch.zeros_like(pack.flat), {})
                    self._st[pack] = st
                b1, b2 = group["betas"]
                for s, e in runs:
                    tick = st[2].get((s, e))
                    if tick is None:    # the kernel advances it: one counter per range it runs over
                        tick = st[2][(s, e)] = torch.zeros(1, dtype=torch.float32, device=pack.flat.device)
                    ops.adam(pack.flat[s:e], pack.grad[s:e], st[0][s:e], st[1][s:e],
                             pack.shadow[s:e] if pack.shadow is not None else None, tick, group["lr"], b1, b2,
                             group["eps"], group["weight_decay"], group["decoupled"], group["grad_scale"])
                if pack.shadow is not None:
                    pack.mark_shadow_current()
        return loss
