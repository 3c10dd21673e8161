"""Whole-step HIP graph: forward + loss + backward + optimizer captured once, replayed per batch.

A training step of these models is 300-1500 kernel launches issued from Python; for the
small configurations (dino_vits16 at 32x32: ~10 us of GPU work per launch) the launch path,
not the GPU, paces the step.  Capturing the step in a hipGraph (through torch.cuda.graphs,
which records everything launched on the capturing stream, our ctypes launches included)
removes it.  Static shapes only: the graph owns input buffers that each call copies into.

    step = GraphedStep(model, criterion, optimizer, x0, y0)
    for x, y in loader:
        loss = step(x, y)           # device tensor, valid until the next call

`warmup` eager steps run on the example batch before the capture (they DO update the
parameters; lazy initialisation — LDS attributes, side stream, workspaces — must not happen
inside a capture, so at least one is needed the first time).  `warm_out` / `warm_loss` hold
the last warm-up step's logits and loss, for callers that account for that batch.
Limits (checked or documented): parameters must only be changed by the captured optimizer
between replays (re-capture after load_state_dict or a learning-rate change: the LR is a
kernel argument baked into the graph; `step.recapture()`).

Data parallelism: an engine with a `ddp.GradReducer` is captured WITH its gradient exchange — the
bucket all-reduces become nodes on the comm stream inside the graph (libvitmi_comm's fork event /
ncclAllReduce / join event; ProcessGroupNCCL's collectives with `VITMI_COMM=pg`), so launch-bound
configurations (dino_vits16 at 32x32: 9.5 ms eager, 3.7 ms replayed) keep the graph at N > 1.  The
warm-up steps run the exchange eagerly first (the communicator must exist before a capture starts);
every rank must capture and replay the same number of times.

Two crashes of round 4 and the contract that came out of them:
* Captures run in `capture_error_mode="thread_local"`: only the CAPTURING thread's unsafe calls invalidate the
  capture, so a helper thread of the process (ProcessGroupNCCL's watchdog polling `hipEventQuery`, a DataLoader's
  pin-memory thread) is legal beside it.  With the own RCCL communicator no helper thread holds an event of the
  exchange at all; the `time.sleep(0.35)` that used to "let the watchdog retire" the warm-up is gone.
* ONE live GraphedStep per engine, and nothing is initialised inside a capture: the engine is resolved EAGERLY and
  pinned before the capture starts (a model whose configuration changed since the last capture gets a new engine —
  new flat buffers, lazily built tables — and therefore an eager warm-up step first), and a second GraphedStep on an
  engine that still has a live one is refused (`close()` the first, or `recapture()` it).
"""
from __future__ import annotations

import weakref

import torch

from ._lib import VitmiError


class GraphedStep:
    def __init__(self, model, criterion, optimizer, x_example, y_example, warmup: int = 2):
        if not x_example.is_cuda:
            raise VitmiError("GraphedStep needs example inputs on the GPU")
        eng = model.engine() if hasattr(model, "engine") else None
        red = getattr(eng, "reducer", None) if eng is not None else None
        if red is not None and (red.world > 1 or red.force) and not red.capturable():
            raise VitmiError("GraphedStep can capture the gradient exchange on RCCL only (libvitmi_comm, or backend 'nccl'): "
                             "a gloo all_reduce of device tensors is not capturable — run the step eagerly")
        if eng is not None:
            other = getattr(eng, "_graphed_step", None)
            other = other() if other is not None else None
            if other is not None and other.graph is not None:
                raise VitmiError("this model's engine already has a live GraphedStep: two captured steps over one engine "
                                 "share its flat buffers, optimizer state and cached tables — call close() on the first "
                                 "(or recapture() it) instead of building a second")
        if red is not None and warmup < 1:
            raise VitmiError("GraphedStep with a GradReducer needs warmup >= 1: the RCCL communicator must be "
                             "created by an eager exchange before the capture starts")
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.x = x_example.detach().clone()
        self.y = y_example.detach().clone()
        self.warmup = int(warmup)
        self.graph = None
        self.loss = None
        self.out = None
        self.warm_out = None
        self.warm_loss = None
        self._lrs = None
        self._pack = None
        self._engine = None               # the engine the live graph was captured on
        self.recapture(self.warmup)

    def close(self):
        """Drop the graph (its private memory pool goes back to the allocator); the engine may be captured again."""
        self.graph = None
        self.out = self.loss = None
        eng = self._engine
        if eng is not None and getattr(eng, "_graphed_step", None) is not None and eng._graphed_step() is self:
            eng._graphed_step = None
        self._engine = None

    def _eager(self):
        self.optimizer.zero_grad(set_to_none=True)
        out = self.model(self.x)
        loss = self.criterion(out, self.y)
        loss.backward()
        self.optimizer.step()
        return out, loss

    def recapture(self, warmup: int = 0):
        """(Re)build the graph from the current parameters and optimizer settings; `warmup`
        eager steps first (0 for a re-capture: everything is initialised already)."""
        self.graph = None
        # Resolve the engine EAGERLY and pin it.  `model.engine()` rebuilds the engine when the model's configuration moved
        # (a DropPath rate edited, a head installed, parameters replaced): new flat buffers, new lazily built tables.  If that
        # happened since the last capture, or this is the first capture, the new engine's first step must run EAGERLY — lazy
        # initialisation (LDS attributes, workspaces, host-built index tables = synchronous copies) is illegal inside a
        # capture, raises there, and the capture_end that follows an invalidated capture is where round 4's segfault sat.
        eng = self.model.engine() if hasattr(self.model, "engine") else None
        if eng is not None and eng is not self._engine and warmup < 1:
            warmup = 1
        # the warm-up passes run on a side stream (torch's capture recipe), so AccumulateGrad
        # nodes of earlier eager steps live on another stream: expected here, not a hazard
        quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if quiet is not None:
            quiet(False)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                 # warm-up off the default stream: lazy
            for _ in range(warmup):                   # initialisation (LDS attributes, side
                out, loss = self._eager()             # streams, workspaces) happens eagerly
                self.warm_out, self.warm_loss = out.detach().clone(), loss.detach().clone()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if eng is not None:
            if self.model.engine() is not eng:
                raise VitmiError("the model's engine changed during the warm-up steps: cannot capture")
            self._engine = eng
            eng._graphed_step = weakref.ref(self)
        # The engine's fp32 -> bf16 weight cast stays OUT of the graph when the captured optimizer is one of the fused ones
        # over ALL of the pack's trainable parameters: those kernels write master and shadow in one pass, so inside the
        # replay loop the shadow is always current, and __call__ checks the pack's version key eagerly before every replay
        # (weights loaded or modified between replays are cast then).  Anything else keeps the recorded cast.
        self._pack = self._vouched_pack()
        if self._pack is not None:
            self._pack.refresh_shadow()                # current at capture time (version check; usually a no-op)
            self._pack.capture_skips_cast = True
        g = torch.cuda.CUDAGraph()
        self.optimizer.zero_grad(set_to_none=True)
        try:
            # thread_local: only this thread's unsafe calls invalidate the capture (module docstring)
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                out = self.model(self.x)
                loss = self.criterion(out, self.y)
                loss.backward()
                self.optimizer.step()
        finally:
            if self._pack is not None:
                self._pack.capture_skips_cast = False
        self.graph, self.out, self.loss = g, out, loss
        self._lrs = [grp["lr"] for grp in self.optimizer.param_groups]

    def _vouched_pack(self):
        """The engine's ParamPack if the optimizer keeps its bf16 shadow current by itself, else None."""
        from .optim import _FusedFlat
        eng = self.model.engine() if hasattr(self.model, "engine") else None
        pack = getattr(eng, "pack", None)
        if pack is None or getattr(pack, "shadow", None) is None or not isinstance(self.optimizer, _FusedFlat):
            return None
        covered = {id(p) for grp in self.optimizer.param_groups for p in grp["params"]}
        if any(p.requires_grad and id(p) not in covered for p in pack.params):
            return None
        return pack

    def __call__(self, x, y):
        if [grp["lr"] for grp in self.optimizer.param_groups] != self._lrs:
            self.recapture()                          # the learning rate is baked into the graph
        # `self.x` / `self.y` are the graph's static inputs: a loader that writes its batch straight into them (the device
        # ingest kernel's output, a resident synthetic batch) passes them back and pays no copy (154 MB per ViT-B/16 step)
        if x is not self.x:
            self.x.copy_(x, non_blocking=True)
        if y is not self.y:
            self.y.copy_(y, non_blocking=True)
        if self._pack is not None:
            self._pack.refresh_shadow()                # eager: casts only if a parameter was written since (version key)
        self.graph.replay()
        return self.loss
