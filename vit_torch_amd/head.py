"""Stand-alone classifier head on libvitmi kernels (linear evaluation: SURVEY §8f rank 2).

`VisionModelZoo.get_classifier_head` (/root/reference/models/vision_all.py:299-320) returns
`Sequential(Linear(bias=True), GELU, ..., Linear(out, bias=False))`.  When that head is part of a
model, the model's engine runs it; in the reference's linear-evaluation mode
(`main.py:184-201`) it is a module of its own, trained on the features of a frozen backbone
(`utils_network.py:143,202-206,413-415`).  `ClassifierHead` is that module: the same
`nn.Sequential` structure and state-dict keys ("0.weight", "0.bias", "2.weight", ...), but
`forward` / `backward` run the fp32 MFMA GEMM with fused bias+GELU / gelu' epilogues, and its
parameters live in a `ParamPack`, so `FusedSGD` updates them.  CPU tensors raise: no fallback.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from ._lib import EPI_BIAS_GELU, EPI_DGELU, VitmiError
from .packing import ParamPack


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, head, x, *params):
        ctx.head = head
        ctx.need_dx = x.requires_grad
        return head._forward(x, save=True)

    @staticmethod
    def backward(ctx, dout):
        head = ctx.head
        held = head._pack.begin_backward()      # torch's accumulation contract (packing.ParamPack.begin_backward)
        dx = head._backward(dout, ctx.need_dx)
        head._pack.end_backward(held)
        grads = []
        for p, gv in zip(head._pack.params, head._pack.fresh_grad_views()):
            if not p.requires_grad:
                grads.append(None)
            elif p.grad is not None and p.grad.data_ptr() == gv.data_ptr():
                grads.append(None)
            else:
                grads.append(gv)
        return (None, dx, *grads)


class ClassifierHead(nn.Sequential):
    def __init__(self, *layers):
        super().__init__(*layers)
        self._layers = _plan(self)
        if self._layers is None:
            raise VitmiError("ClassifierHead: layers must be Linear[, GELU], ..., Linear")
        self._pack = None
        self._saved = None
        self.reducer = None          # ddp.GradReducer (Network(ddp=...)): the head's gradients leave as one bucket after its backward

    def engine(self):
        """(Re)build the flat parameter buffers (after .to(device) / load_state_dict)."""
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise VitmiError("move the head to the GPU before the first forward")
        if self._pack is None or not self._pack.is_current() or len(self._pack.params) != sum(1 for _ in self.parameters()):
            self._pack = ParamPack(list(self.named_parameters()), dev, shadow=False)
        return self

    @property
    def pack(self):
        return self.engine()._pack

    def forward(self, x):
        if not x.is_cuda:
            raise VitmiError("vit_torch_amd heads run on an MI355X (HIP) device; got a CPU tensor "
                             "and there is no CPU fallback")
        if not self._layers:
            return x
        self.engine()
        x = x.float().contiguous()
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self._pack.params)):
            return _HeadFn.apply(self, x, *self._pack.params)
        return self._forward(x, save=False)

    # ---- kernels
    def _forward(self, x, save):
        pk = self._pack
        acts, pres, cur = [x], [], x
        for lin, gelu in self._layers:
            out = torch.empty((x.shape[0], lin.out_features), dtype=torch.float32, device=x.device)
            bias = pk.f32(lin.bias) if lin.bias is not None else None
            if gelu:
                pre = torch.empty_like(out)
                ops.gemm(cur, pk.f32(lin.weight), out, epilogue=EPI_BIAS_GELU, bias=bias, C2=pre)
                pres.append(pre)
            else:
                ops.gemm(cur, pk.f32(lin.weight), out, bias=bias)
                pres.append(None)
            acts.append(out)
            cur = out
        if save:
            self._saved = (acts, pres)
        return cur

    def _backward(self, dout, need_dx):
        try:
            dx = self._backward_impl(dout, need_dx)
        except BaseException:
            if self.reducer is not None:
                self.reducer.abort()
            raise
        if self.reducer is not None:         # data-parallel linear evaluation: all of the head's gradients are final here
            self.reducer.section_ready([p for p in self._pack.params if p.requires_grad])
            self.reducer.finish()
        return dx

    def _backward_impl(self, dout, need_dx):
        if self._saved is None:
            raise VitmiError("backward called without a saved forward (or called twice)")
        acts, pres = self._saved
        self._saved = None
        pk = self._pack
        d = dout.contiguous().float()
        if self._layers[-1][1]:
            raise VitmiError("a head ending in GELU is not supported")
        for li in range(len(self._layers) - 1, -1, -1):
            lin, _ = self._layers[li]
            ops.gemm(d, acts[li], pk.g(lin.weight), a_kmajor=False, b_kmajor=False)
            if lin.bias is not None:
                ops.colsum(d, pk.g(lin.bias))
            if li == 0 and not need_dx:
                return None
            dx = torch.empty((d.shape[0], lin.in_features), dtype=torch.float32, device=d.device)
            if li > 0 and self._layers[li - 1][1]:
                ops.gemm(d, pk.f32(lin.weight), dx, b_kmajor=False, epilogue=EPI_DGELU, aux=pres[li - 1])
            else:
                ops.gemm(d, pk.f32(lin.weight), dx, b_kmajor=False)
            d = dx
        return d


def _plan(seq):
    """[(Linear, gelu_after)] or None."""
    out, mods, i = [], list(seq), 0
    while i < len(mods):
        if not isinstance(mods[i], nn.Linear):
            return None
        gelu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.GELU)
        out.append((mods[i], gelu))
        i += 2 if gelu else 1
    return out
