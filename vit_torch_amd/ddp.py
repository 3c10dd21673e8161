"""Data-parallel gradient exchange: one process per GPU, RCCL over xGMI.

The reference is single-process; this is the build's
multi-GPU path.  Gradients live in ONE flat fp32 buffer laid out in
named_parameters order, and the backward pass finishes whole sections of it
(head+norm, then blocks depth-1 .. 0, then the embeddings), so a bucket is a
contiguous slice: it is all-reduced IN PLACE (no packing copy) with
`async_op=True` the moment its section is complete.  ProcessGroupNCCL runs the
collective on its own HIP stream, ordered after the kernels already queued on
the compute stream, so the exchange of block i overlaps the backward of blocks
i-1.. .  `finish()` makes the compute stream wait for every bucket; the SUM is
turned into the mean by the optimizer's `grad_scale = 1/world` (one fewer pass
over the gradients).  xGMI is point-to-point, so buckets are kept large
(one transformer block = 28 MB fp32 for ViT-B) rather than many small ones.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, pack, group=None, min_bucket_elems: int = 4 << 20, force: bool = False):
        """force: exchange even in a world of one (exercises the RCCL stream ordering on a
        single GPU; tests only)."""
        self.pack = pack
        self.group = group
        self.force = bool(force)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # RCCL's all-reduce kernels hold a few dozen CUs while a bucket is in flight, and a 256x256-tile
        # GEMM / attention-backward workgroup needs a whole CU: with the persistent grids (a FIXED list of
        # tiles / pairs per workgroup) the workgroups that find their CU taken would start only when
        # another one has walked its whole list.  So from the first bucket's launch until finish() the
        # engine passes VITMI_LAUNCH_SHARED_DEVICE with every GEMM / attention-backward call (one tile per
        # workgroup: the dispatcher balances over the free CUs); the forward pass, the start of the
        # backward and the optimizer — ordered before / after the exchange on the compute stream — keep
        # the persistent form.  The flag travels with each call (`launch_flags`, ABI 105): nothing
        # process-wide is switched, so an exception in backward cannot leave another engine degraded.
        self.comm_active = False
        self.min_bucket = int(min_bucket_elems)
        self._pending_lo: Optional[int] = None
        self._pending_hi: Optional[int] = None
        self._works: List = []
        self.launched: List[tuple] = []      # (lo, hi) of every bucket, for tests
        # bench.py's dp_proxy: with `timing` set, an event is recorded on the compute stream in front of every bucket's
        # launch (the last one is kept) and another when finish() has joined them all: tail_ms() = what a step waits
        # for between handing over its last bucket and being free to run the optimizer
        self.timing = False
        self._ev_last = self._ev_done = None

    # called by the engine when the gradients of `params` are final
    def section_ready(self, params) -> None:
        if (self.world == 1 and not self.force) or not params:
            return
        lo, hi = self.pack.span(params)
        if self._pending_lo is None:
            self._pending_lo, self._pending_hi = lo, hi
        else:
            if hi != self._pending_lo and lo != self._pending_hi:
                self._flush()                 # not adjacent: send what we have
                self._pending_lo, self._pending_hi = lo, hi
            else:
                self._pending_lo = min(lo, self._pending_lo)
                self._pending_hi = max(hi, self._pending_hi)
        if self._pending_hi - self._pending_lo >= self.min_bucket:
            self._flush()

    def _flush(self) -> None:
        if self._pending_lo is None:
            return
        lo, hi = self._pending_lo, self._pending_hi
        self._pending_lo = self._pending_hi = None
        buf = self.pack.grad[lo:hi]
        if self.timing:
            self._ev_last = torch.cuda.Event(enable_timing=True)
            self._ev_last.record()
        self.comm_active = True
        self._works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self.launched.append((lo, hi))

    def finish(self) -> None:
        """Order the current stream after every outstanding bucket."""
        self._flush()
        for w in self._works:
            w.wait()                      # the compute stream waits; kernels launched from here on run after the exchange
        self._works.clear()
        self.comm_active = False
        if self.timing and self._ev_last is not None:
            self._ev_done = torch.cuda.Event(enable_timing=True)
            self._ev_done.record()

    def tail_ms(self):
        """Milliseconds between the last bucket's launch and finish() returning on the compute stream (timing mode)."""
        if self._ev_last is None or self._ev_done is None:
            return None
        torch.cuda.synchronize()
        return round(self._ev_last.elapsed_time(self._ev_done), 4)

    def launch_flags(self) -> int:
        """What the engine passes as `launch_flags` right now (vitmi.h VITMI_LAUNCH_*)."""
        from ._lib import LAUNCH_SHARED_DEVICE
        return LAUNCH_SHARED_DEVICE if self.comm_active else 0

    def abort(self) -> None:
        """After an exception inside backward: join what was launched, drop what was pending."""
        self._pending_lo = self._pending_hi = None
        try:
            for w in self._works:
                w.wait()
        finally:
            self._works.clear()
            self.comm_active = False

    def broadcast_parameters(self, src: int = 0) -> None:
        if self.world > 1:
            dist.broadcast(self.pack.flat, src=src, group=self.group)
            self.pack.invalidate_shadow()     # the master was written behind the version counters
