"""The reference's training harness, reduced to the hot path it drives
(/root/reference/utils_network.py): `LRSchedule` (:35-73), the optimizer table (:119-126),
`get_lr_scheduler` (:529-544) and the per-batch step of `run_one_epoch` (:406-453):

    inputs.to(device); labels.to(device)
    outputs = model(inputs)                         :418
    loss = loss_fn(outputs, labels)                 :430
    optimizer.zero_grad(); loss.backward(); optimizer.step()   :440-442
    correct = (argmax(outputs) == labels)           :85-95
    loss.item()                                     :452

`run_one_epoch` returns the per-batch losses and the per-sample correct flags the reference
feeds to its Stats object; `fit(log=RunLog(...))` writes them in the reference's JSON schema
(vit_torch_amd.stats; the progress bars of utils_stats.py are out of scope).  Unlike
the reference's two device->host syncs per step, loss and correct count stay on the device
and are read once per epoch.

Data parallelism (the hook the reference sketches at utils_datasets.py:876-891 as
`ddp={'size', 'rank'}`): `Network(..., ddp={'size': W, 'rank': r})` in every rank of an initialised
`torch.distributed` job (one process per GPU, backend "nccl" = RCCL; "gloo" on CPU) shards the data by
rank (`distributed_loader` = the reference's DistributedSampler hook; `ShardedLoader` = rank r's rows
of every global batch of an existing loader), broadcasts rank 0's parameters, attaches a
`ddp.GradReducer` to the engine (bucketed all-reduce overlapped with the backward), lets the optimizer
turn the SUM into the mean (`grad_scale = 1/W`) and replaces the reference's two host syncs per STEP
(utils_network.py:94, :452) by ONE 2-float all-reduce (loss sum, correct count) per EPOCH.
"""
from __future__ import annotations

import math
from typing import Iterable, List, Optional

import numpy as np
import torch
import torch.nn as nn

from .loss import CrossEntropyLoss
from .optim import FusedAdaBelief, FusedAdadelta, FusedAdagrad, FusedAdamW, FusedSGD, _FusedFlat


class LRSchedule:
    """Multiplicative LR factors as functions of the epoch index (utils_network.py:35-73)."""

    @classmethod
    def get_base_fn(cls):
        return lambda e: 1.0

    @classmethod
    def get_step_fn(cls, step=10, gamma=0.5):
        assert step > 0 and 1 >= gamma >= 0
        return lambda e: gamma ** np.floor(e / step)

    @classmethod
    def get_exp_fn(cls, gamma=0.99, step=1):
        assert 1 >= gamma >= 0 and step > 0
        return lambda e: gamma ** float(e / step)

    @classmethod
    def get_cosine(cls, step=20, min_scale=0.1):
        assert 1 >= min_scale >= 0
        return lambda e: (1.0 - min_scale) / 2 * (np.cos(np.mod(e / step, 0.5) * np.pi * 2) + 1) + min_scale

    @classmethod
    def get_cosine_exp(cls, step=20, min_scale=0.1, gamma=0.5):
        assert 1 >= min_scale >= 0 and 1 >= gamma >= 0
        cos = cls.get_cosine(step, min_scale)
        return lambda e: cos(e) * gamma ** float(e / step)


def get_lr_scheduler(optimizer, type="step", step=10, gamma=0.5, scale=0.1):
    """utils_network.py:529-544, including its quirk that type 'none' maps to `lambda e: e`
    (LR x epoch, i.e. LR 0 in epoch 0 — SURVEY Appendix C)."""
    if type == "none" or not isinstance(type, str):
        fn = lambda e: e
    elif type == "step":
        fn = LRSchedule.get_step_fn(step=step, gamma=gamma)
    elif type == "exp":
        fn = LRSchedule.get_exp_fn(gamma=gamma)
    elif type == "cos":
        fn = LRSchedule.get_cosine(step=step, min_scale=scale)
    elif type == "cos_exp":
        fn = LRSchedule.get_cosine_exp(step=step, min_scale=scale, gamma=gamma)
    else:
        raise NotImplementedError(f"lr scheduler {type} has not been implemented")
    return torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=fn)


def classification_count_correct(outputs, labels):
    """utils_network.py:85-95 without the host round trip: bool [B] on the device."""
    with torch.no_grad():
        return torch.argmax(outputs, dim=-1) == labels


class ShardedLoader:
    """Rank `rank`'s rows of every batch of `loader` (an iterable of (inputs, labels) GLOBAL batches): rows
    [rank * B/size, (rank + 1) * B/size).  Every rank must iterate the same global batches (same seed / same file order);
    a global batch whose size is not a multiple of `size` is refused — with unequal shards the mean of the ranks' mean
    losses is not the global mean the reference computes (the DistributedSampler of `Network.distributed_loader` pads
    instead, as the reference's hook does)."""

    def __init__(self, loader, rank: int, size: int):
        self.loader, self.rank, self.size = loader, int(rank), int(size)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for inputs, labels in self.loader:
            B = labels.shape[0]
            if B % self.size:
                raise ValueError(f"ShardedLoader: a global batch of {B} samples does not split over {self.size} ranks")
            per = B // self.size
            yield inputs[self.rank * per:(self.rank + 1) * per], labels[self.rank * per:(self.rank + 1) * per]


class Network:
    """Minimal counterpart of utils_network.Network for the models of this package."""

    # name -> f(params, lr, grad_scale); grad_scale = 1 / world in a data-parallel job (the all-reduce SUMs), else 1
    optimizer_fns = {
        "sgd": lambda params, lr, gs=1.0: FusedSGD(params, lr=lr, momentum=0.9, grad_scale=gs),      # utils_network.py:120
        "torch_sgd": lambda params, lr, gs=1.0: torch.optim.SGD(params, lr=lr, momentum=0.9),
        "adam": lambda params, lr, gs=1.0: FusedAdamW(params, lr=lr, weight_decay=0.0, decoupled=False, grad_scale=gs),   # :121
        "adamw": lambda params, lr, gs=1.0: FusedAdamW(params, lr=lr, grad_scale=gs),                                     # :124
        "torch_adamw": lambda params, lr, gs=1.0: torch.optim.AdamW(params, lr=lr),
        "adadelta": lambda params, lr, gs=1.0: FusedAdadelta(params, lr=lr, grad_scale=gs),                               # :122
        "adagrad": lambda params, lr, gs=1.0: FusedAdagrad(params, lr=lr, grad_scale=gs),                                 # :123
        "adabelief": lambda params, lr, gs=1.0: FusedAdaBelief(params, lr=lr, grad_scale=gs),                             # :125
    }

    def __init__(self, model, opt="sgd", loss_fn=None, lr=1e-3, lr_type="step", lr_step=10, lr_gamma=0.5,
                 lr_scale=0.1, device="cuda", epochs=1, hip_graph=False, frozen_model_bottom=None, ddp=None,
                 earlystop_epoch=0):
        if not isinstance(model, nn.Module):
            raise ValueError("`model` must be a torch.nn.Module")          # utils_network.py:167-170
        self.model = model.to(device)
        self.device = device
        # linear evaluation (utils_network.py:143,202-206): frozen backbones run first, under
        # no_grad (:413-415), and only `model` (the head) is trained
        if isinstance(frozen_model_bottom, nn.Module):
            frozen_model_bottom = [frozen_model_bottom]
        self.frozen_model_bottom = [m.to(device) for m in frozen_model_bottom] if isinstance(frozen_model_bottom, list) else []
        self.loss_fn = loss_fn if loss_fn is not None else CrossEntropyLoss()
        self.epochs = epochs
        if opt not in self.optimizer_fns:
            raise ValueError(f"optimizer `{opt}` is not supported")
        if not opt.startswith("torch_"):     # the flat buffers exist after the engine is built
            if not hasattr(self.model, "engine"):
                raise ValueError(f"optimizer `{opt}` (fused) needs a vit_torch_amd model or ClassifierHead")
            self.model.engine()
        # `earlystop_epoch` is accepted and IGNORED, as in the reference (utils_network.py:161 stores nothing; `fit` has its
        # own default of 10, :233, and main.py never passes the CLI value on — SURVEY Appendix C)
        self.earlystop_epoch = earlystop_epoch
        self.world, self.rank, self.reducer, self._ddp_pack = 1, 0, None, None
        if ddp is not None:
            self._setup_ddp(ddp)
        self.optimizer = self.optimizer_fns[opt](self.model.parameters(), lr, 1.0 / self.world)
        self._grad_scale_in_optimizer = isinstance(self.optimizer, _FusedFlat)
        self.lr_scheduler = get_lr_scheduler(self.optimizer, lr_type, lr_step, lr_gamma, lr_scale)
        # hip_graph: training steps replay a captured HIP graph (vit_torch_amd.graph.GraphedStep);
        # batches of another shape (the last, short batch of an epoch) run eagerly
        self.hip_graph = bool(hip_graph)
        self._graphed = None

    # ---- data parallelism ----------------------------------------------------------------------------------------
    def _setup_ddp(self, ddp):
        import torch.distributed as dist
        from .ddp import GradReducer
        from .packing import ParamPack
        if not dist.is_available() or not dist.is_initialized():
            raise ValueError("Network(ddp=...) needs an initialised torch.distributed process group "
                             "(one process per GPU: backend 'nccl' = RCCL; 'gloo' on CPU)")
        group = ddp.get("group")
        self.world, self.rank = int(ddp["size"]), int(ddp["rank"])
        if self.world != dist.get_world_size(group) or self.rank != dist.get_rank(group):
            raise ValueError(f"ddp={{'size': {self.world}, 'rank': {self.rank}}} does not match the process group "
                             f"(size {dist.get_world_size(group)}, rank {dist.get_rank(group)})")
        self._group = group
        if self.frozen_model_bottom:
            for m in self.frozen_model_bottom:            # frozen backbones: same weights everywhere, no gradients
                for t in list(m.parameters()) + list(m.buffers()):
                    dist.broadcast(t.data, src=0, group=group)
        if hasattr(self.model, "engine"):
            eng = self.model.engine()
            self.reducer = GradReducer(eng.pack, group=group, force=bool(ddp.get("force", False)))
            eng.reducer = self.reducer                    # buckets leave while the backward is still running
        else:
            # any other nn.Module (the reference's Network takes every nn.Module, utils_network.py:167-170): the same
            # reducer over a flat copy of the gradients, fed after backward() in reverse module order
            self._ddp_pack = ParamPack([(n, p) for n, p in self.model.named_parameters() if p.requires_grad],
                                       next(self.model.parameters()).device, shadow=False)
            self.reducer = GradReducer(self._ddp_pack, group=group, force=bool(ddp.get("force", False)))
        self.reducer.broadcast_parameters(0)
        for b in self.model.buffers():
            dist.broadcast(b.data, src=0, group=group)

    def distributed_loader(self, dataset, batch_size, shuffle=False, num_workers=0, seed=0):
        """The reference's hook (utils_datasets.py:876-891): `DistributedSampler(num_replicas=ddp['size'],
        rank=ddp['rank'], shuffle=...)` under a DataLoader with the PER-RANK batch size, drop_last=False.  Without `ddp` the
        plain loader of the reference's else-branch (:893-898)."""
        from torch.utils.data import DataLoader, DistributedSampler
        if self.world == 1 and self.reducer is None:
            return DataLoader(dataset, batch_size=batch_size, shuffle=bool(shuffle), num_workers=num_workers)
        sampler = DistributedSampler(dataset, num_replicas=self.world, rank=self.rank, shuffle=bool(shuffle), seed=seed)
        return DataLoader(dataset, sampler=sampler, batch_size=batch_size, num_workers=num_workers, pin_memory=False,
                          drop_last=False)

    def shard(self, loader):
        """This rank's rows of every global batch of `loader` (identity without ddp)."""
        return loader if self.reducer is None or self.world == 1 else ShardedLoader(loader, self.rank, self.world)

    def _exchange_module_grads(self):
        """Generic nn.Module in a data-parallel job: gradients -> flat buffer -> GradReducer (sections in reverse module
        order, like a backward pass finishes them) -> back, averaged."""
        pack, red = self._ddp_pack, self.reducer
        for p in pack.params:
            g = pack.g(p)
            if p.grad is None:
                g.zero_()
            else:
                g.copy_(p.grad)
        seen = set()
        for child in reversed(list(self.model.children())):
            ps = [p for p in child.parameters() if id(p) in pack.index and id(p) not in seen]
            seen.update(id(p) for p in ps)
            red.section_ready(ps)
        red.section_ready([p for p in pack.params if id(p) not in seen])       # parameters owned by the root module
        red.finish()
        for p in pack.params:
            if p.grad is None:
                p.grad = torch.empty_like(p)
            torch.mul(pack.g(p), 1.0 / self.world, out=p.grad)

    def _train_step(self, inputs, labels):
        """utils_network.py:418-442 for one batch; in a data-parallel job the gradients every rank steps on are the
        global-batch mean."""
        outputs = self.model(inputs)
        loss = self.loss_fn(outputs, labels)
        self.optimizer.zero_grad()
        loss.backward()                   # engine models: the reducer's buckets leave in here and are joined at its end
        if self.reducer is not None and (self.world > 1 or self.reducer.force):
            if self._ddp_pack is not None:
                self._exchange_module_grads()
            elif not self._grad_scale_in_optimizer:
                # stock torch.optim has no grad_scale argument: SUM -> mean over the flat gradient buffer (vitmi_scale_cast
                # in place, the whole buffer as one row with the factor as its row scale)
                from . import ops
                g = self.model.engine().pack.grad
                if g.is_cuda:
                    if getattr(self, "_inv_world", None) is None:
                        self._inv_world = torch.full((1,), 1.0 / self.world, dtype=torch.float32, device=g.device)
                    ops.scale_cast(g, g, None, M=1, N=g.numel(), rowscale=self._inv_world, rows_per_group=1)
                else:
                    g.mul_(1.0 / self.world)
        self.optimizer.step()
        return outputs, loss

    def run_one_epoch(self, dataloader: Iterable, training: bool = True):
        losses: List[torch.Tensor] = []
        corrects: List[torch.Tensor] = []
        for inputs, labels in dataloader:
            inputs = inputs.to(self.device)
            labels = labels.to(self.device)
            if self.frozen_model_bottom:
                with torch.no_grad():
                    for m in self.frozen_model_bottom:
                        inputs = m(inputs)
            if training and self.hip_graph:
                g = self._graphed
                if g is None and inputs.is_cuda:
                    from .graph import GraphedStep
                    # one eager step on this batch (it counts as the batch's update), then the capture
                    g = self._graphed = GraphedStep(self.model, self.loss_fn, self.optimizer, inputs, labels, warmup=1)
                    outputs, loss = g.warm_out, g.warm_loss
                elif g is not None and inputs.shape == g.x.shape and labels.shape == g.y.shape:
                    loss = g(inputs, labels).clone()
                    outputs = g.out.clone()
                else:
                    outputs, loss = self._train_step(inputs, labels)
            elif training:
                outputs, loss = self._train_step(inputs, labels)
            else:
                with torch.no_grad():
                    outputs = self.model(inputs)
                    loss = self.loss_fn(outputs, labels)
            corrects.append(classification_count_correct(outputs, labels))
            losses.append(loss.detach())
        if self.reducer is not None and (self.world > 1 or self.reducer.force) and losses:
            # ONE 2-float all-reduce per epoch (loss sum, correct count) instead of the reference's two host syncs per step
            # (utils_network.py:94, :452).  Shards are equal-sized (DistributedSampler pads, ShardedLoader refuses anything
            # else), so the global figures are the sums over ranks divided by world x the local counts.
            loss_vec = torch.stack(losses).float()
            corr_vec = torch.cat(corrects)
            both = torch.stack([loss_vec.sum(), corr_vec.sum().float()]).contiguous()
            both = self.reducer.allreduce_metrics(both).cpu().tolist()
            loss_values = loss_vec.cpu().tolist()
            correct = corr_vec.cpu().numpy().reshape(-1)
            return {"loss": loss_values, "loss_avg": both[0] / (self.world * len(loss_values)),
                    "correct": correct, "acc": both[1] / (self.world * correct.size),
                    "samples_global": self.world * int(correct.size)}
        loss_values = torch.stack(losses).float().cpu().tolist() if losses else []
        correct = torch.cat(corrects).cpu().numpy().reshape(-1) if corrects else np.zeros(0, dtype=bool)
        return {"loss": loss_values, "loss_avg": float(np.mean(loss_values)) if loss_values else math.nan,
                "correct": correct, "acc": float(correct.mean()) if correct.size else math.nan}

    def fit(self, train_loader, val_loader=None, epochs: Optional[int] = None, log=None, verbose=False,
            earlystop_epoch: int = 10):
        """`log`: a vit_torch_amd.stats.RunLog; gets one entry per split and epoch (train with
        the epoch's LR, val with lr 0.0 as in the reference's files) and is saved after each.

        Early stopping as the reference does it (utils_network.py:322-328, checked at the top of the next epoch :256-259):
        after a validation round, once at least `earlystop_epoch` validation accuracies exist, training stops if none of
        the last `earlystop_epoch` reaches the best one so far.  The default is the reference's `fit` default of 10 (:233) —
        the constructor's / CLI's `earlystop_epoch` never reaches it there (main.py:253, :275), and does not here.
        In a data-parallel job the accuracies are the all-reduced global ones, so every rank stops in the same epoch."""
        history = []
        val_accs: List[float] = []
        stopping = False
        self.stopped_early_after = None
        for epoch in range(epochs if epochs is not None else self.epochs):
            if stopping:
                self.stopped_early_after = epoch
                if verbose and self.rank == 0:
                    print(f"Stopped Early after {epoch} epochs! Training Fishished.", flush=True)      # (sic, :258)
                break
            lr = self.optimizer.param_groups[0]["lr"]
            rec = {"epoch": epoch, "lr": lr}
            for split, loader in (("train", train_loader), ("val", val_loader)):
                if loader is None:
                    continue
                if log is not None:
                    log.new_round(split)
                r = rec[split] = self.run_one_epoch(loader, training=(split == "train"))
                if split == "train":
                    self.lr_scheduler.step()                               # utils_network.py:311-313
                if log is not None:
                    log.finish_round(split, epoch=epoch, lr=lr if split == "train" else 0.0, loss=r["loss_avg"],
                                     acc=r["acc"], sample=int(r.get("samples_global", r["correct"].size)))
                    if verbose:
                        print(log.console_line(split), flush=True)
                if split == "val":
                    val_accs.append(r["acc"])
                    best = max(val_accs)
                    if len(val_accs) >= earlystop_epoch and max(val_accs[-earlystop_epoch:] or [best]) < best:
                        stopping = True
            history.append(rec)
        if log is not None:
            log.finish()
        return history
