"""CPU restatement (pure PyTorch, fp32) of the reference's DeepLabV3+ / ResNet path (BASELINE config 5).

TEST INFRASTRUCTURE: checker only (see oracle/__init__.py).  State-dict keys equal the reference's
``network.modeling.deeplabv3plus_resnet101(...)`` model (``backbone.*`` from IntermediateLayerGetter,
``classifier.*`` from DeepLabHeadV3Plus).  Citations are paths relative to the reference root.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def resnet_plan(layers=(3, 4, 23, 3), output_stride=16):
    """[(layer name, block index, inplanes, planes, stride, dilation, has_downsample)] following
    network/backbone/resnet.py:173-195 (_make_layer) and network/modeling.py:46-51."""
    rswd = [False, True, True] if output_stride == 8 else [False, False, True]
    plan, inplanes, dilation = [], 64, 1
    for li, (planes, blocks) in enumerate(zip((64, 128, 256, 512), layers)):
        stride = 1 if li == 0 else 2
        dilate = False if li == 0 else rswd[li - 1]
        prev_dil = dilation
        if dilate:
            dilation *= stride
            stride = 1
        ds = stride != 1 or inplanes != planes * 4
        plan.append((f"layer{li + 1}", 0, inplanes, planes, stride, prev_dil, ds))
        inplanes = planes * 4
        for b in range(1, blocks):
            plan.append((f"layer{li + 1}", b, inplanes, planes, 1, dilation, False))
    return plan


def state_spec(num_classes=19, layers=(3, 4, 23, 3), output_stride=16):
    spec = OrderedDict()

    def bn(prefix, c):
        spec[prefix + ".weight"] = ((c,), "bn_w")
        spec[prefix + ".bias"] = ((c,), "bn_b")
        spec[prefix + ".running_mean"] = ((c,), "bn_rm")
        spec[prefix + ".running_var"] = ((c,), "bn_rv")
        spec[prefix + ".num_batches_tracked"] = ((), "bn_nbt")

    spec["backbone.conv1.weight"] = ((64, 3, 7, 7), "conv")
    bn("backbone.bn1", 64)
    for lname, b, inpl, planes, stride, dil, ds in resnet_plan(layers, output_stride):
        p = f"backbone.{lname}.{b}"
        spec[p + ".conv1.weight"] = ((planes, inpl, 1, 1), "conv")
        bn(p + ".bn1", planes)
        spec[p + ".conv2.weight"] = ((planes, planes, 3, 3), "conv")
        bn(p + ".bn2", planes)
        spec[p + ".conv3.weight"] = ((planes * 4, planes, 1, 1), "conv")
        bn(p + ".bn3", planes * 4)
        if ds:
            spec[p + ".downsample.0.weight"] = ((planes * 4, inpl, 1, 1), "conv")
            bn(p + ".downsample.1", planes * 4)
    c = "classifier."
    spec[c + "project.0.weight"] = ((48, 256, 1, 1), "conv")
    bn(c + "project.1", 48)
    spec[c + "aspp.convs.0.0.weight"] = ((256, 2048, 1, 1), "conv")
    bn(c + "aspp.convs.0.1", 256)
    for i in (1, 2, 3):
        spec[c + f"aspp.convs.{i}.0.weight"] = ((256, 2048, 3, 3), "conv")
        bn(c + f"aspp.convs.{i}.1", 256)
    spec[c + "aspp.convs.4.1.weight"] = ((256, 2048, 1, 1), "conv")
    bn(c + "aspp.convs.4.2", 256)
    spec[c + "aspp.project.0.weight"] = ((256, 1280, 1, 1), "conv")
    bn(c + "aspp.project.1", 256)
    spec[c + "classifier.0.weight"] = ((256, 304, 3, 3), "conv")
    bn(c + "classifier.1", 256)
    spec[c + "classifier.3.weight"] = ((num_classes, 256, 1, 1), "conv")
    spec[c + "classifier.3.bias"] = ((num_classes,), "bias")
    return spec


def make_state(seed=7, num_classes=19, layers=(3, 4, 23, 3), output_stride=16, residual_gain=1.0):
    """Seeded random state.  residual_gain scales the last BatchNorm weight of every bottleneck (bn3): 1.0 is the plain
    random init (an ill-conditioned 101-layer net: fp32 and fp64 runs of the reference differ by 2e-3 on the logits), 0.25
    the small-residual-gain init of the well-conditioned fixture (same random stream, so all other entries are equal)."""
    rng = np.random.default_rng(seed)
    st = OrderedDict()
    for name, (shape, kind) in state_spec(num_classes, layers, output_stride).items():
        if kind == "conv":
            fan_out = shape[0] * shape[2] * shape[3]
            v = rng.standard_normal(shape, dtype=np.float32) * np.float32(math.sqrt(2.0 / fan_out))
        elif kind == "bn_w":
            v = (1.0 + 0.1 * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)
            if residual_gain != 1.0 and name.endswith(".bn3.weight"):
                v = (v * np.float32(residual_gain)).astype(np.float32)
        elif kind in ("bn_b", "bias"):
            v = (0.1 * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)
        elif kind == "bn_rm":
            v = (0.05 * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)
        elif kind == "bn_rv":
            v = (1.0 + 0.2 * rng.random(shape, dtype=np.float32)).astype(np.float32)
        else:
            st[name] = torch.zeros((), dtype=torch.int64)
            continue
        st[name] = torch.from_numpy(np.ascontiguousarray(v))
    return st


def trainable_names(state):
    return [k for k in state if not (k.endswith("running_mean") or k.endswith("running_var")
                                     or k.endswith("num_batches_tracked"))]


def batch_norm(x, state, prefix, training):
    w, b = state[prefix + ".weight"], state[prefix + ".bias"]
    if not training:
        return F.batch_norm(x, state[prefix + ".running_mean"], state[prefix + ".running_var"], w, b, False, 0.0, BN_EPS)
    with torch.no_grad():
        n = x.numel() // x.shape[1]
        mean = x.mean(dim=(0, 2, 3))
        var_unb = x.var(dim=(0, 2, 3), unbiased=False) * (n / max(n - 1, 1))
        state[prefix + ".running_mean"].mul_(1 - BN_MOMENTUM).add_(mean, alpha=BN_MOMENTUM)
        state[prefix + ".running_var"].mul_(1 - BN_MOMENTUM).add_(var_unb, alpha=BN_MOMENTUM)
        state[prefix + ".num_batches_tracked"] += 1
    return F.batch_norm(x, None, None, w, b, True, 0.0, BN_EPS)


def bottleneck(x, state, p, stride, dil, ds, training):
    """network/backbone/resnet.py:97-119."""
    out = F.relu(batch_norm(F.conv2d(x, state[p + ".conv1.weight"]), state, p + ".bn1", training))
    out = F.conv2d(out, state[p + ".conv2.weight"], None, stride, dil, dil)
    out = F.relu(batch_norm(out, state, p + ".bn2", training))
    out = batch_norm(F.conv2d(out, state[p + ".conv3.weight"]), state, p + ".bn3", training)
    idt = x
    if ds:
        idt = batch_norm(F.conv2d(x, state[p + ".downsample.0.weight"], None, stride), state, p + ".downsample.1", training)
    return F.relu(out + idt)


def backbone(img, state, training, layers=(3, 4, 23, 3), output_stride=16):
    """IntermediateLayerGetter over resnet (network/utils.py:240-251): {'low_level': layer1, 'out': layer4}.
    No input normalisation inside the model."""
    x = F.conv2d(img, state["backbone.conv1.weight"], None, 2, 3)
    x = F.relu(batch_norm(x, state, "backbone.bn1", training))
    x = F.max_pool2d(x, 3, 2, 1)
    feats = {}
    for lname, b, inpl, planes, stride, dil, ds in resnet_plan(layers, output_stride):
        x = bottleneck(x, state, f"backbone.{lname}.{b}", stride, dil, ds, training)
        feats[lname] = x
    return feats["layer1"], feats["layer4"]


def aspp(x, state, training, rates=(6, 12, 18), dropout_p=0.1):
    """network/_deeplab.py:140-169."""
    c = "classifier.aspp."
    res = [F.relu(batch_norm(F.conv2d(x, state[c + "convs.0.0.weight"]), state, c + "convs.0.1", training))]
    for i, r in enumerate(rates, start=1):
        y = F.conv2d(x, state[c + f"convs.{i}.0.weight"], None, 1, r, r)
        res.append(F.relu(batch_norm(y, state, c + f"convs.{i}.1", training)))
    size = x.shape[-2:]
    p = F.adaptive_avg_pool2d(x, 1)
    p = F.relu(batch_norm(F.conv2d(p, state[c + "convs.4.1.weight"]), state, c + "convs.4.2", training))
    res.append(F.interpolate(p, size=size, mode="bilinear", align_corners=False))
    y = torch.cat(res, dim=1)
    y = F.relu(batch_norm(F.conv2d(y, state[c + "project.0.weight"]), state, c + "project.1", training))
    return F.dropout(y, dropout_p, training)


def head(low, out, state, training, rates=(6, 12, 18)):
    """DeepLabHeadV3Plus.forward, network/_deeplab.py:47-56."""
    c = "classifier."
    ll = F.relu(batch_norm(F.conv2d(low, state[c + "project.0.weight"]), state, c + "project.1", training))
    o = aspp(out, state, training, rates)
    o = F.interpolate(o, size=ll.shape[2:], mode="bilinear", align_corners=False)
    y = torch.cat([ll, o], dim=1)
    y = F.relu(batch_norm(F.conv2d(y, state[c + "classifier.0.weight"], None, 1, 1), state, c + "classifier.1", training))
    return F.conv2d(y, state[c + "classifier.3.weight"], state[c + "classifier.3.bias"])


def deeplab_forward(img, state, training=True, return_supcon_feature=False, layers=(3, 4, 23, 3), output_stride=16):
    """_SimpleSegmentationModel.forward, network/utils.py:166-194 -> the reference 4-tuple."""
    rates = (12, 24, 36) if output_stride == 8 else (6, 12, 18)
    low, out = backbone(img, state, training, layers, output_stride)
    fine_feat = out
    if return_supcon_feature:
        bsz = out.shape[0] // 2
        out, low = out[:bsz], low[:bsz]
    fine_feat0 = out
    before = head(low, out, state, training, rates)
    fine_feat0 = F.interpolate(fine_feat0, size=before.shape[-2:], mode="bilinear", align_corners=False)
    seg = F.interpolate(before, size=img.shape[2:], mode="bilinear", align_corners=False)
    return seg, before, fine_feat, fine_feat0
