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
"""
from __future__ import annotations

import math
from typing import Iterable, List, Optional

import numpy as np
import torch
import torch.nn as nn

from .loss import CrossEntropyLoss
from .optim import FusedAdaBelief, FusedAdadelta, FusedAdagrad, FusedAdamW, FusedSGD


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


class Network:
    """Minimal counterpart of utils_network.Network for the models of this package."""

    optimizer_fns = {
        "sgd": lambda params, lr: FusedSGD(params, lr=lr, momentum=0.9),      # utils_network.py:120
        "torch_sgd": lambda params, lr: torch.optim.SGD(params, lr=lr, momentum=0.9),
        "adam": lambda params, lr: FusedAdamW(params, lr=lr, weight_decay=0.0, decoupled=False),   # utils_network.py:121
        "adamw": lambda params, lr: FusedAdamW(params, lr=lr),                                     # utils_network.py:124
        "torch_adamw": lambda params, lr: torch.optim.AdamW(params, lr=lr),
        "adadelta": lambda params, lr: FusedAdadelta(params, lr=lr),                               # utils_network.py:122
        "adagrad": lambda params, lr: FusedAdagrad(params, lr=lr),                                 # utils_network.py:123
        "adabelief": lambda params, lr: FusedAdaBelief(params, lr=lr),                             # utils_network.py:125
    }

    def __init__(self, model, opt="sgd", loss_fn=None, lr=1e-3, lr_type="step", lr_step=10, lr_gamma=0.5,
                 lr_scale=0.1, device="cuda", epochs=1, hip_graph=False, frozen_model_bottom=None):
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
        self.optimizer = self.optimizer_fns[opt](self.model.parameters(), lr)
        self.lr_scheduler = get_lr_scheduler(self.optimizer, lr_type, lr_step, lr_gamma, lr_scale)
        # hip_graph: training steps replay a captured HIP graph (vit_torch_amd.graph.GraphedStep);
        # batches of another shape (the last, short batch of an epoch) run eagerly
        self.hip_graph = bool(hip_graph)
        self._graphed = None

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
                    outputs = self.model(inputs)
                    loss = self.loss_fn(outputs, labels)
                    self.optimizer.zero_grad()
                    loss.backward()
                    self.optimizer.step()
            elif training:
                outputs = self.model(inputs)
                loss = self.loss_fn(outputs, labels)
                self.optimizer.zero_grad()
                loss.backward()
                self.optimizer.step()
            else:
                with torch.no_grad():
                    outputs = self.model(inputs)
                    loss = self.loss_fn(outputs, labels)
            corrects.append(classification_count_correct(outputs, labels))
            losses.append(loss.detach())
        loss_values = torch.stack(losses).float().cpu().tolist() if losses else []
        correct = torch.cat(corrects).cpu().numpy().reshape(-1) if corrects else np.zeros(0, dtype=bool)
        return {"loss": loss_values, "loss_avg": float(np.mean(loss_values)) if loss_values else math.nan,
                "correct": correct, "acc": float(correct.mean()) if correct.size else math.nan}

    def fit(self, train_loader, val_loader=None, epochs: Optional[int] = None, log=None, verbose=False):
        """`log`: a vit_torch_amd.stats.RunLog; gets one entry per split and epoch (train with
        the epoch's LR, val with lr 0.0 as in the reference's files) and is saved after each."""
        history = []
        for epoch in range(epochs if epochs is not None else self.epochs):
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
                                     acc=r["acc"], sample=int(r["correct"].size))
                    if verbose:
                        print(log.console_line(split), flush=True)
            history.append(rec)
        if log is not None:
            log.finish()
        return history
