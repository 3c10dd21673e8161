"""Pretrained-checkpoint ingestion from LOCAL files: the reference
downloads its weights (`torch.hub.load_state_dict_from_url`), this build never touches the
network — the key handling is what is reproduced.

* CaiT (/root/reference/models/cait.py:377-385): the file holds `{"model": {...}}` whose keys
  carry a `module.` prefix; every key of the model's own state dict must be present (strict).
* Swin (/root/reference/models/swin.py:831-840): `checkpoint["model"]`, `strict=False`
  (buffers such as `attn_mask` / `relative_position_index` may be absent or extra).
* DINO (/root/reference/models/vision_all.py:156): torch.hub returns a plain state dict for the
  backbone; `head` keys are optional, and a `pos_embed` of another grid is loaded as stored —
  the engine resizes it per input (bicubic, as upstream DINO does).
"""
from __future__ import annotations

import torch


def _as_dict(ckpt):
    if isinstance(ckpt, (str, bytes)) or hasattr(ckpt, "__fspath__"):
        ckpt = torch.load(ckpt, map_location="cpu")
    return ckpt


def load_reference_checkpoint(model, checkpoint, family: str):
    """Load a reference-format checkpoint (path or already-loaded object) into `model`.
    family: "cait" | "swin" | "dino".  Returns what `load_state_dict` returns."""
    ckpt = _as_dict(checkpoint)
    if family == "cait":
        src = ckpt["model"]
        own = model.state_dict()
        missing = [k for k in own if "module." + k not in src]
        if missing:
            raise KeyError(f"CaiT checkpoint lacks {len(missing)} keys, e.g. module.{missing[0]}")
        res = model.load_state_dict({k: src["module." + k] for k in own}, strict=True)
    elif family == "swin":
        res = model.load_state_dict(ckpt["model"], strict=False)
    elif family == "dino":
        sd = ckpt.get("state_dict", ckpt) if isinstance(ckpt, dict) else ckpt
        sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}
        sd = {k[len("backbone."):] if k.startswith("backbone.") else k: v for k, v in sd.items()}
        own = model.state_dict()
        if "pos_embed" in sd and "pos_embed" in own and sd["pos_embed"].shape != own["pos_embed"].shape:
            with torch.no_grad():      # other pretraining grid: keep the stored table, resized per input
                model.pos_embed = torch.nn.Parameter(sd["pos_embed"].to(own["pos_embed"].device).clone())
        res = model.load_state_dict(sd, strict=False)
    else:
        raise ValueError(f"unknown checkpoint family [{family}]")
    eng = getattr(model, "_engine", None)
    if eng is not None:
        model._engine = None               # parameters may have been re-created: rebuild the flat buffers
    return res
