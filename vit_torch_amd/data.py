"""Device-side input pipeline.

The reference transforms every sample on CPU DataLoader workers
(/root/reference/utils_datasets.py:553-582: `RandomCrop(S, padding=max(2, S//12), fill=128)`,
`RandomHorizontalFlip`, `ToTensor`, `Normalize(mean, std)`; test: `ToTensor`, `Normalize`).
`DeviceAugment` takes the raw uint8 NHWC batch on the GPU and produces the normalised fp32
NCHW tensor in one kernel; the per-sample crop offsets and flip flags are drawn with a torch
generator (pass them explicitly to pin a batch).  Resize (bicubic, only when the requested
size differs from the stored one) is not part of this kernel.

`DeviceAugment.patch_rows(images, patch_size, dtype)` goes one step further (round 3): the same
transform written straight into the patch rows the patch-embedding GEMM contracts
(`vitmi_ingest_patchify`: uint8 NHWC -> bf16 patch rows in one pass, no fp32 NCHW intermediate); the
returned `PatchRows` is accepted by `VisionTransformer.forward` in place of the image tensor.
"""
from __future__ import annotations

import torch

from . import ops

NORM = {   # utils_datasets.py:586-589 (STL-10), :644-647 (CIFAR-10)
    "stl10": dict(mean=[0.44671062065972217, 0.43980983983523964, 0.40664644709967324],
                  std=[0.2603409782662331, 0.25657727311344447, 0.27126738145225493]),
    "cifar10": dict(mean=[0.4914, 0.4822, 0.4465], std=[0.247, 0.243, 0.261]),
}


class DeviceAugment:
    def __init__(self, image_size: int, mean=None, std=None, train: bool = True, device="cuda", generator=None):
        self.S = int(image_size)
        self.pad = max(2, self.S // 12) if train else 0          # utils_datasets.py:567
        self.train = train
        self.device = torch.device(device)
        self.mean = torch.tensor(mean, dtype=torch.float32, device=self.device) if mean is not None else None
        self.std = torch.tensor(std, dtype=torch.float32, device=self.device) if std is not None else None
        self.generator = generator

    def draw(self, B, H, W):
        """(off_y, off_x, flip) as torchvision draws them: top-left uniform in [0, H+2p-S], flip p=0.5."""
        if not self.train:
            return None, None, None
        g = self.generator
        oy = torch.randint(0, H + 2 * self.pad - self.S + 1, (B,), generator=g, dtype=torch.int32)
        ox = torch.randint(0, W + 2 * self.pad - self.S + 1, (B,), generator=g, dtype=torch.int32)
        fl = (torch.rand(B, generator=g) < 0.5).to(torch.uint8)
        return oy.to(self.device), ox.to(self.device), fl.to(self.device)

    def __call__(self, images_u8_nhwc, off_y=None, off_x=None, flip=None):
        x = images_u8_nhwc
        if not x.is_cuda:
            x = x.to(self.device, non_blocking=True)
        B, H, W, C = x.shape
        if self.train and off_y is None:
            off_y, off_x, flip = self.draw(B, H, W)
        out = torch.empty((B, C, self.S, self.S), dtype=torch.float32, device=x.device)
        return ops.image_ingest(x.contiguous(), out, off_y, off_x, flip, self.mean, self.std, self.pad)

    def patch_rows(self, images_u8_nhwc, patch_size: int, dtype=torch.bfloat16, cls_rows: int = 1,
                   off_y=None, off_x=None, flip=None) -> "PatchRows":
        x = images_u8_nhwc
        if not x.is_cuda:
            x = x.to(self.device, non_blocking=True)
        B, H, W, C = x.shape
        if self.train and off_y is None:
            off_y, off_x, flip = self.draw(B, H, W)
        g = self.S // patch_size
        rows = torch.empty((B * (cls_rows + g * g), C * patch_size * patch_size), dtype=dtype, device=x.device)
        ops.ingest_patchify(x.contiguous(), rows, off_y, off_x, flip, self.mean, self.std, self.S, self.pad, patch_size,
                            cls_rows)
        return PatchRows(rows, B, C, self.S, self.S, patch_size, cls_rows)


class PatchRows:
    """A batch already in patch-row form ([B*(cls_rows + gh*gw), C*p*p], the GEMM operand dtype): what
    `VisionTransformer.forward` takes instead of an fp32 [B,C,H,W] tensor when the input pipeline runs on
    the device (the engine then skips its own patch gather)."""
    is_cuda = True

    def __init__(self, rows, B, C, H, W, p, cls_rows):
        self.rows, self.B, self.C, self.H, self.W, self.p, self.cls_rows = rows, B, C, H, W, p, cls_rows
        self.device = rows.device
