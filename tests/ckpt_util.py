"""Synthetic torchvision-style ResNet-18 ImageNet state dict (keys / shapes of torchvision.models.resnet18, the file
resnet_pyramid.py:14-20 points at), filled from a seeded numpy stream.  Shared by tests/golden/make_golden_ckpt.py
(which feeds it to the REFERENCE) and tests/test_checkpoint.py (which feeds it to this repository's model)."""
import hashlib
from collections import OrderedDict

import numpy as np
import torch


def tv_resnet18_spec():
    spec = OrderedDict()

    def bn(prefix, c):
        for k in ("weight", "bias", "running_mean", "running_var"):
            spec[f"{prefix}.{k}"] = (c,)
        spec[f"{prefix}.num_batches_tracked"] = ()

    spec["conv1.weight"] = (64, 3, 7, 7)
    bn("bn1", 64)
    inpl = 64
    for li, planes in enumerate((64, 128, 256, 512), start=1):
        for b in range(2):
            p = f"layer{li}.{b}"
            spec[p + ".conv1.weight"] = (planes, inpl if b == 0 else planes, 3, 3)
            bn(p + ".bn1", planes)
            spec[p + ".conv2.weight"] = (planes, planes, 3, 3)
            bn(p + ".bn2", planes)
            if b == 0 and li > 1:
                spec[p + ".downsample.0.weight"] = (planes, inpl, 1, 1)
                bn(p + ".downsample.1", planes)
        inpl = planes
    spec["fc.weight"] = (1000, 512)
    spec["fc.bias"] = (1000,)
    return spec


def tv_resnet18_state(seed=11):
    g = np.random.default_rng(seed)
    sd = OrderedDict()
    for k, shape in tv_resnet18_spec().items():
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.tensor(7, dtype=torch.int64)
        elif k.endswith("running_var"):
            sd[k] = torch.from_numpy((0.5 + g.random(shape)).astype(np.float32))
        else:
            sd[k] = torch.from_numpy((g.standard_normal(shape) * 0.05).astype(np.float32))
    return sd


def digest(t: torch.Tensor) -> str:
    t = t.detach().cpu().contiguous()
    return hashlib.sha1(t.numpy().tobytes()).hexdigest() + ":" + "x".join(map(str, t.shape)) + ":" + str(t.dtype)
