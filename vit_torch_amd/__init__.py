"""vit_torch_amd — MI355X-native ViT forward/backward training path (libvitmi)."""
from ._lib import VitmiError  # noqa: F401
from .loss import CrossEntropyLoss  # noqa: F401
from .optim import FusedAdaBelief, FusedAdadelta, FusedAdagrad, FusedAdamW, FusedSGD  # noqa: F401
from .vision_all import VisionModelZoo  # noqa: F401
from .vit import VisionTransformer  # noqa: F401
from .cait import cait_models  # noqa: F401
from .swin import SwinTransformer  # noqa: F401
from .graph import GraphedStep  # noqa: F401
from .head import ClassifierHead  # noqa: F401
from .checkpoint import load_reference_checkpoint  # noqa: F401
from .stats import RunLog  # noqa: F401

__all__ = ["VisionModelZoo", "VisionTransformer", "CrossEntropyLoss", "FusedSGD", "FusedAdamW", "FusedAdagrad", "FusedAdadelta", "FusedAdaBelief", "GraphedStep", "ClassifierHead", "load_reference_checkpoint", "RunLog", "VitmiError"]
