"""The reference's optimizer table (/root/reference/utils_network.py:119-126: sgd with momentum
0.9, adam, adadelta, adagrad, adamw, adabelief; step at :442) as ONE fused HIP kernel per
contiguous span of a model's flat parameter / gradient / state buffers, which also refreshes
the bf16 weight shadow in the same pass."""
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


class _StepCounters:
    """Device-side step counts (fp32 [1], advanced by the kernels: HIP-graph safe), one per span a
    kernel runs over.  When the spans change between steps (a parameter frozen or unfrozen, param
    groups edited) the counts the elements have ALREADY taken are carried over: the old counters
    are read back once, and a new span whose elements disagree is cut where the count changes — so
    bias corrections stay those of each element's own history (torch keeps `step` per parameter),
    and an element that joins late starts at 0 with its zero state."""

    def __init__(self):
        self.ticks = {}          # (start, end) -> tensor
        self._runs_key = None    # the `runs` the current pieces were cut for

    def spans(self, runs, device):
        """[(start, end, tick)] covering `runs`."""
        # unchanged runs -> the existing pieces, no read-back, the SAME tick tensors (a captured HIP graph keeps advancing
        # them).  The pieces may be finer than the runs (a run cut where the step count changes after an unfreeze), so
        # the test is on the runs the pieces were cut for, not on the dictionary's keys (the key test never
        # passed again after a cut, and every later step paid a host sync and fresh tick tensors)
        if self._runs_key == tuple(runs) and self.ticks:
            return [(s, e, t) for (s, e), t in sorted(self.ticks.items())]
        old = sorted((s, e, float(t.item())) for (s, e), t in self.ticks.items())    # rare: one sync

        def count_at(i):
            for s, e, c in old:
                if s <= i < e:
                    return c
            return 0.0

        cuts = sorted({b for s, e, _ in old for b in (s, e)})
        new = {}
        for s, e in runs:
            edges = [s] + [c for c in cuts if s < c < e] + [e]
            piece_lo, piece_c = edges[0], count_at(edges[0])
            for a, b in zip(edges[:-1], edges[1:]):
                c = count_at(a)
                if c != piece_c:
                    new[(piece_lo, a)] = piece_c
                    piece_lo, piece_c = a, c
            new[(piece_lo, e)] = piece_c
        self.ticks = {k: torch.full((1,), c, dtype=torch.float32, device=device) for k, c in new.items()}
        self._runs_key = tuple(runs)
        return [(s, e, t) for (s, e), t in sorted(self.ticks.items())]


class _FusedFlat(torch.optim.Optimizer):
    """Walks the param groups, finds the flat spans, keeps `n_state` state arrays per pack."""
    n_state = 0
    counted = False                 # the rule depends on the step count

    def __init__(self, params, defaults):
        super().__init__(params, defaults)
        _check_disjoint(self.param_groups)
        self._st = weakref.WeakKeyDictionary()     # pack -> ([state arrays], {group index: _StepCounters})

    def _state_init(self, flat, group):
        return [torch.zeros_like(flat) for _ in range(self.n_state)]

    def _apply(self, group, p, g, st, shadow, tick):
        raise NotImplementedError

    @torch.no_grad()
    def reset_state(self):
        """State arrays and step counts back to their initial values, in place (the device buffers a
        captured HIP graph points at stay the same)."""
        for arrays, counters in self._st.values():
            for a in arrays:
                a.fill_(getattr(self, "_fill", 0.0))
            for c in counters.values():
                for t in c.ticks.values():
                    t.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for gi, group in enumerate(self.param_groups):
            for pack, runs in _group_runs(group):
                st = self._st.get(pack)
                if st is None:
                    st = self._st[pack] = (self._state_init(pack.flat, group), {})
                arrays, counters = st
                if self.counted:
                    spans = counters.setdefault(gi, _StepCounters()).spans(runs, pack.flat.device)
                else:
                    spans = [(s, e, None) for s, e in runs]
                for s, e, tick in spans:
                    self._apply(group, pack.flat[s:e], pack.grad[s:e], [a[s:e] for a in arrays],
                                pack.shadow[s:e] if pack.shadow is not None else None, tick)
        return loss


class FusedSGD(_FusedFlat):
    """Same update rule as torch.optim.SGD(lr, momentum, dampening=0, nesterov=False,
    weight_decay=0): buf = momentum*buf + g (buf starts at 0, i.e. buf_1 = g_1);
    p -= lr*buf.  `grad_scale` multiplies g first (1/world_size after a SUM
    all-reduce)."""
    n_state = 1

    def __init__(self, params, lr=1e-3, momentum=0.9, grad_scale=1.0):
        super().__init__(params, dict(lr=lr, momentum=momentum, grad_scale=grad_scale))

    def _apply(self, group, p, g, st, shadow, tick):
        ops.sgd_momentum(p, g, st[0], shadow, group["lr"], group["momentum"], group["grad_scale"])


class FusedAdamW(_FusedFlat):
    """`optim.AdamW(params, lr)` / `optim.Adam(params, lr)` of the reference's optimizer table
    (/root/reference/utils_network.py:121,124; every linear-evaluation log uses AdamW) as ONE
    kernel over the flat buffers of a ParamPack, with the bf16 shadow refreshed in the same
    pass.  torch's defaults and update order; the step count lives on the device, so the
    step can be replayed from a HIP graph.  `decoupled=False` gives Adam (L2 decay)."""
    n_state = 2
    counted = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, decoupled=True,
                 grad_scale=1.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, decoupled=decoupled,
                                      grad_scale=grad_scale))

    def _apply(self, group, p, g, st, shadow, tick):
        b1, b2 = group["betas"]
        ops.adam(p, g, st[0], st[1], shadow, tick, group["lr"], b1, b2, group["eps"], group["weight_decay"],
                 group["decoupled"], group["grad_scale"])


class FusedAdagrad(_FusedFlat):
    """`optim.Adagrad(params, lr)` (/root/reference/utils_network.py:123), torch's defaults."""
    n_state = 1
    counted = True

    def __init__(self, params, lr=1e-2, lr_decay=0.0, weight_decay=0.0, initial_accumulator_value=0.0, eps=1e-10,
                 grad_scale=1.0):
        super().__init__(params, dict(lr=lr, lr_decay=lr_decay, weight_decay=weight_decay, eps=eps,
                                      initial_accumulator_value=initial_accumulator_value, grad_scale=grad_scale))

    def _state_init(self, flat, group):
        self._fill = float(group["initial_accumulator_value"])
        return [torch.full_like(flat, self._fill)]

    def _apply(self, group, p, g, st, shadow, tick):
        ops.adagrad(p, g, st[0], shadow, tick, group["lr"], group["lr_decay"], group["eps"], group["weight_decay"],
                    group["grad_scale"])


class FusedAdadelta(_FusedFlat):
    """`optim.Adadelta(params, lr)` (/root/reference/utils_network.py:122), torch's defaults."""
    n_state = 2

    def __init__(self, params, lr=1.0, rho=0.9, eps=1e-6, weight_decay=0.0, grad_scale=1.0):
        super().__init__(params, dict(lr=lr, rho=rho, eps=eps, weight_decay=weight_decay, grad_scale=grad_scale))

    def _apply(self, group, p, g, st, shadow, tick):
        ops.adadelta(p, g, st[0], st[1], shadow, group["lr"], group["rho"], group["eps"], group["weight_decay"],
                     group["grad_scale"])


class FusedAdaBelief(_FusedFlat):
    """`AdaBelief(params, lr, eps=1e-16, betas=(0.9, 0.999), weight_decouple=True, rectify=True)`
    (/root/reference/utils_network.py:125).  The adabelief_pytorch package is not in this container:
    the rule is restated from the published algorithm (include/vitmi.h, vitmi_adabelief)."""
    n_state = 2
    counted = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-16, weight_decay=0.0, weight_decouple=True,
                 rectify=True, grad_scale=1.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      weight_decouple=weight_decouple, rectify=rectify, grad_scale=grad_scale))

    def _apply(self, group, p, g, st, shadow, tick):
        b1, b2 = group["betas"]
        ops.adabelief(p, g, st[0], st[1], shadow, tick, group["lr"], b1, b2, group["eps"], group["weight_decay"],
                      group["weight_decouple"], group["rectify"], group["grad_scale"])
