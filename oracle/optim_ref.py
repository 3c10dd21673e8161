"""TEST INFRASTRUCTURE (oracle): CPU restatement of the optimizer rules of the reference's table
(/root/reference/utils_network.py:119-126) that torch itself does not ship.

AdaBelief: the reference imports `adabelief_pytorch.AdaBelief` (utils_network.py:17) and builds it with
eps=1e-16, betas=(0.9, 0.999), weight_decouple=True, rectify=True (:125).  That package is a third-party
dependency (unpinned in requirements.txt) that is NOT in this container, so this is a restatement of the
published algorithm (Zhuang et al., "AdaBelief Optimizer", NeurIPS 2020, Algorithm 2 + the RAdam
rectification of its reference implementation; defaults amsgrad=False, fixed_decay=False,
degenerated_to_sgd=True, weight_decay=0) — PARITY UNPINNED: no fixture of the package's output exists.
Adagrad / Adadelta / Adam / AdamW / SGD are checked against torch.optim directly."""
import math

import torch


class AdaBeliefRef(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-16, weight_decay=0.0, weight_decouple=True,
                 rectify=True, degenerated_to_sgd=True):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      weight_decouple=weight_decouple, rectify=rectify,
                                      degenerated_to_sgd=degenerated_to_sgd))

    @torch.no_grad()
    def step(self):
        for g in self.param_groups:
            b1, b2 = g["betas"]
            for p in g["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["m"] = torch.zeros_like(p)
                    st["s"] = torch.zeros_like(p)
                grad = p.grad
                if g["weight_decouple"]:
                    p.mul_(1.0 - g["lr"] * g["weight_decay"])
                elif g["weight_decay"] != 0:
                    grad = grad.add(p, alpha=g["weight_decay"])
                st["step"] += 1
                t = st["step"]
                bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
                m, s = st["m"], st["s"]
                m.mul_(b1).add_(grad, alpha=1 - b1)
                r = grad - m
                s.mul_(b2).addcmul_(r, r, value=1 - b2).add_(g["eps"])          # eps enters the state
                if not g["rectify"]:
                    denom = (s.sqrt() / math.sqrt(bc2)).add_(g["eps"])
                    p.addcdiv_(m, denom, value=-g["lr"] / bc1)
                    continue
                rho_inf = 2.0 / (1.0 - b2) - 1.0
                rho_t = rho_inf - 2.0 * t * (b2 ** t) / bc2
                if rho_t >= 5:
                    rt = math.sqrt(bc2 * (rho_t - 4) / (rho_inf - 4) * (rho_t - 2) / rho_t * rho_inf / (rho_inf - 2))
                    p.addcdiv_(m, s.sqrt().add_(g["eps"]), value=-g["lr"] * rt / bc1)
                elif g["degenerated_to_sgd"]:
                    p.add_(m, alpha=-g["lr"] / bc1)
