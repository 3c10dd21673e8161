"""VisionModelZoo: the reference's model factory (/root/reference/models/vision_all.py:30-376)
for the families on the MI355X hot path.  Same classmethods, argument names
and error behaviour; the returned nn.Modules run on libvitmi HIP kernels.

Differences, on purpose:
  * `pretrained=True` raises: the reference downloads weights through
    torch.hub (vision_all.py:156), this build has no network; load a
    state_dict into the returned module instead (names/shapes match upstream).
  * the dino branch applies the installed classifier (`apply_head=True`);
    upstream DINO's forward ignores `.head` ([recall]) — pass
    `apply_head=False` to get the raw upstream behaviour.
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.nn import GELU

from .vit import VisionTransformer


class VisionModelZoo:
    archs_types = {
        "dino": ["dino_vits16", "dino_vits8", "dino_vitb16", "dino_vitb8"],
        # models/vision_all.py:44-49 / models/cait.py:13-18
        "cait": ["cait_M48", "cait_M36", "cait_S36", "cait_S24", "cait_S24_224", "cait_XS24", "cait_XXS24",
                 "cait_XXS24_224", "cait_XXS36", "cait_XXS36_224"],
        # models/vision_all.py:50-69 (the 224 / window-7 classification variants)
        "swin": ["swin_tiny_patch4_window7_224", "swin_small_patch4_window7_224", "swin_base_patch4_window7_224",
                 "swin_large_patch4_window7_224"],
    }
    # name: (patch, embed_dim, depth, heads) — DINO vit_small / vit_base
    dino_cfg = {
        "dino_vits16": (16, 384, 12, 6),
        "dino_vits8": (8, 384, 12, 6),
        "dino_vitb16": (16, 768, 12, 12),
        "dino_vitb8": (8, 768, 12, 12),
    }

    @classmethod
    def get_model(cls, arch=None, pretrained=True, image_channels=3, classifier=None,
                  classifier_act=GELU(), root_path=None, *args, **kwargs):
        if arch is None:
            if isinstance(classifier, (int, list)):
                return cls.get_classifier_head(in_features=image_channels, classifier_units=classifier,
                                               classifier_act=classifier_act)
            return nn.Identity()
        _type = None
        for k, v in cls.archs_types.items():
            if arch in v:
                _type = k
                break
        if _type is None:
            raise ValueError("arch [{}] not found!".format(arch))
        fn = getattr(cls, "get_model_" + _type)
        return fn(arch=arch, pretrained=pretrained, image_channels=image_channels,
                  classifier=classifier, classifier_act=classifier_act, *args, **kwargs)

    @classmethod
    def get_model_dino(cls, arch="dino_vits16", pretrained=True, image_channels=3, classifier=None,
                       classifier_act=GELU(), return_separate=False, apply_head=True, **model_kwargs):
        if pretrained:
            raise RuntimeError("pretrained weights cannot be downloaded here; build with "
                               "pretrained=False and load_state_dict() a local DINO checkpoint")
        p, d, depth, heads = cls.dino_cfg[arch]
        in_ch = 3 if image_channels is None else image_channels
        assert isinstance(in_ch, int) and in_ch > 0
        model = VisionTransformer(patch_size=p, in_chans=in_ch, embed_dim=d, depth=depth,
                                  num_heads=heads, mlp_ratio=4.0, qkv_bias=True, eps=1e-6,
                                  **model_kwargs)
        # vision_all.py:157-158 re-initialises every sub-module with its PyTorch default
        cls.reset_parameters(model)
        if isinstance(classifier, (int, list)):
            backbone_features = model.norm.weight.data.shape[-1]
            model.head = cls.get_classifier_head(in_features=backbone_features,
                                                 classifier_units=classifier,
                                                 classifier_act=classifier_act)
            model.apply_head = bool(apply_head)
        if return_separate:
            _head = model.head
            model.head = nn.Identity()
            model.apply_head = False
            return model, _head
        return model

    @classmethod
    def get_model_cait(cls, arch="cait_M36", pretrained=True, image_channels=3, classifier=None,
                       classifier_act=GELU(), return_separate=False, **model_kwargs):
        """models/vision_all.py:184-221 (timm create_model -> models/cait.py factory)."""
        from .cait import create_cait
        model = create_cait(arch, pretrained=bool(pretrained), num_classes=1000,
                            in_chans=3 if image_channels is None else image_channels, **model_kwargs)
        if classifier is False:
            model.head = nn.Identity()
            model.head_dist = model.head
        elif isinstance(classifier, (int, list)):
            backbone_features = model.norm.weight.data.shape[-1]
            model.head = cls.get_classifier_head(in_features=backbone_features, classifier_units=classifier,
                                                 classifier_act=classifier_act)
            model.head_dist = model.head
        if return_separate:
            _head = model.head
            model.head = nn.Identity()
            return model, _head
        return model

    @classmethod
    def get_model_swin(cls, arch="swin_base_patch4_window7_224", pretrained=True, image_channels=3,
                       classifier=None, classifier_act=GELU(), return_separate=False, **model_kwargs):
        """models/vision_all.py:223-297.  The model is built with its config's DropPath rate
        (models/swin.py:768-820), active while `model.training` (see swin.py's docstring)."""
        from .swin import get_swin_model
        assert image_channels == 3
        model = get_swin_model(arch, pretrained=pretrained, **model_kwargs)
        if isinstance(classifier, (int, list)):
            backbone_features = model.norm.weight.data.shape[-1]
            model.head = cls.get_classifier_head(in_features=backbone_features, classifier_units=classifier,
                                                 classifier_act=classifier_act)
        if return_separate:
            _head = model.head
            model.head = nn.Identity()
            return model, _head
        return model

    @classmethod
    def get_classifier_head(cls, in_features, classifier_units=None, classifier_act=GELU()):
        linear_layers = []
        if isinstance(classifier_units, int):
            classifier_units = [classifier_units]
        if isinstance(classifier_units, list):
            for i, v in enumerate(classifier_units):
                fin = in_features if i == 0 else classifier_units[i - 1]
                is_not_last = i < len(classifier_units) - 1
                linear_layers.append(nn.Linear(in_features=fin, out_features=v, bias=is_not_last))
                if is_not_last:
                    linear_layers.append(classifier_act)
        # same structure / state-dict keys as the reference's nn.Sequential; stand-alone
        # (linear evaluation) it runs on libvitmi kernels too (head.py)
        if all(isinstance(m, (nn.Linear, nn.GELU)) for m in linear_layers):
            from .head import ClassifierHead
            return ClassifierHead(*linear_layers)
        return nn.Sequential(*linear_layers)

    @classmethod
    def reset_parameters(cls, m=[]):
        if isinstance(m, list):
            _ = [cls.reset_parameters(v) for v in m]
        if hasattr(m, "children"):
            cls.reset_parameters(list(m.children()))
        if hasattr(m, "reset_parameters"):
            m.reset_parameters()

    @classmethod
    def get_output_shape(cls, model, input_shape=(1, 3, 224, 224), device="cuda"):
        with torch.no_grad():
            out = model.to(device)(torch.zeros(*input_shape, device=device))
        return list(out.shape)
