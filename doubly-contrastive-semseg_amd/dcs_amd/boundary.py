"""Boundary-aware label weights on the device: mirror of ``LabelBoundaryTransform``
(dataloaders/custom_transforms_acdc.py:656-693; the same class in custom_transforms.py:1696 / custom_transforms2.py:659).

The reference runs, per sample on the CPU worker, one ``cv2.distanceTransform(mask, cv2.DIST_L2, maskSize=3)`` per present
class and combines them into ``label_distance_weight = exp(-d / (2 std(d)))``.  Here the whole batch is transformed by
``dcs_label_boundary_weights`` (one chamfer transform for all classes, see csrc/label_boundary.hip); the result feeds
``BoundaryAwareFocalLoss`` (utils/loss.py:44) without leaving HBM.

Accepted ``example['label']``: an int64/uint8 device tensor [H,W] (one sample, as in the reference) or [B,H,W]."""
from __future__ import annotations

import torch

from . import ops


class LabelBoundaryTransform:
    def __init__(self, num_classes, reduce=False, ignore_id=255):
        self.num_classes = num_classes
        self.reduce = reduce
        self.ignore_id = ignore_id

    def __call__(self, example):
        labels = example["label"]
        if not torch.is_tensor(labels):
            raise RuntimeError("LabelBoundaryTransform (MI355X): 'label' must be a device tensor (no CPU fallback)")
        single = labels.dim() == 2
        lab = (labels[None] if single else labels).to(torch.int64).contiguous()
        weight, dist = ops.label_boundary_weights(lab, self.num_classes, self.ignore_id)
        if self.reduce:
            example["label_distance_weight"] = weight[0] if single else weight
        else:
            # [num_classes, H, W]: the class's distance inside its mask, -1 elsewhere (reference layout)
            d = dist.to(torch.float32) * (1.0 / 65536.0)
            cls = torch.arange(self.num_classes, device=lab.device).view(1, -1, 1, 1)
            full = torch.where(lab[:, None] == cls, d[:, None], torch.full_like(d[:, None], -1.0))
            example["label_distance_transform"] = full[0] if single else full
        return example
