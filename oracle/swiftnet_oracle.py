"""CPU restatement (pure PyTorch, fp32) of the reference hot path.

TEST INFRASTRUCTURE: checker + CPU baseline only (see oracle/__init__.py).

Every function cites the reference file:line it restates (paths relative to
the reference repo root).  The model is held as a flat ``state`` dict whose
keys/shapes equal the reference ``WeatherNet(resnet18).state_dict()``
(SURVEY.md 9.1), so fixtures generated from the reference load directly.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

IMG_MEAN = (73.15, 82.90, 72.3)     # network/weathernet.py:37
IMG_STD = (47.67, 48.49, 47.73)     # network/weathernet.py:38
NUM_FEATURES = 128                  # network/backbone/resnet_pyramid.py:130
BN_EPS = 1e-5
BN_MOMENTUM = 0.1

# (name, planes, stride) of the four ResNet-18 stages, two BasicBlocks each
# (network/backbone/resnet_pyramid.py:170-179, :401).
LAYERS = (("layer1", 64, 1), ("layer2", 128, 2), ("layer3", 256, 2), ("layer4", 512, 2))


# --------------------------------------------------------------------------- #
# state construction
# --------------------------------------------------------------------------- #
def state_spec(num_classes: int = 19) -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """name -> (shape, kind) for every state_dict entry, in the reference's
    registration order (network/backbone/resnet_pyramid.py:131-237,
    network/weathernet.py:61-62).  kind in {conv, bn_w, bn_b, bn_rm, bn_rv,
    bn_nbt, bias, buf_mean, buf_std}."""
    spec: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()
    fe = "feature_extractor."

    def bn(prefix, c):
        spec[prefix + ".weight"] = ((c,), "bn_w")
        spec[prefix + ".bias"] = ((c,), "bn_b")
        spec[prefix + ".running_mean"] = ((c,), "bn_rm")
        spec[prefix + ".running_var"] = ((c,), "bn_rv")
        spec[prefix + ".num_batches_tracked"] = ((), "bn_nbt")

    spec[fe + "img_mean"] = ((1, 3, 1, 1), "buf_mean")
    spec[fe + "img_std"] = ((1, 3, 1, 1), "buf_std")
    spec[fe + "conv1.weight"] = ((64, 3, 7, 7), "conv")
    for l in range(3):
        bn(fe + f"bn1_{l}", 64)
    inpl = 64
    for li, (lname, planes, stride) in enumerate(LAYERS):
        for b in range(2):
            p = fe + f"{lname}.{b}"
            cin = inpl if b == 0 else planes
            spec[p + ".conv1.weight"] = ((planes, cin, 3, 3), "conv")
            bn(p + ".bn1", planes)
            spec[p + ".conv2.weight"] = ((planes, planes, 3, 3), "conv")
            bn(p + ".bn2", planes)
            if b == 0 and (stride != 1 or inpl != planes):
                spec[p + ".downsample.0.weight"] = ((planes, inpl, 1, 1), "conv")
                bn(p + ".downsample.1", planes)
        inpl = planes
        spec[fe + f"upsample_bottlenecks{li + 1}.weight"] = ((NUM_FEATURES, planes, 1, 1), "conv")
    for i in range(1, 6):
        bn(fe + f"upsample_blends{i}.blend_conv.norm", NUM_FEATURES)
        spec[fe + f"upsample_blends{i}.blend_conv.conv.weight"] = ((NUM_FEATURES, NUM_FEATURES, 3, 3), "conv")
    bn("segmentation.norm", NUM_FEATURES)
    spec["segmentation.conv.weight"] = ((num_classes, NUM_FEATURES, 1, 1), "conv")
    spec["segmentation.conv.bias"] = ((num_classes,), "bias")
    return spec


def make_state(seed: int = 1, num_classes: int = 19) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic, platform-stable random state (numpy PCG64 stream, not the
    torch generator) with the magnitudes of the reference's init
    (kaiming fan_out for convs, resnet_pyramid.py:249-254) but non-trivial BN
    affine / running statistics so that every term is exercised."""
    rng = np.random.default_rng(seed)
    st: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, (shape, kind) in state_spec(num_classes).items():
        if kind == "conv":
            fan_out = shape[0] * shape[2] * shape[3]
            v = rng.standard_normal(shape, dtype=np.float32) * np.float32(math.sqrt(2.0 / fan_out))
        elif kind == "bn_w":
            v = (1.0 + 0.1 * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)
        elif kind in ("bn_b", "bias"):
            v = (0.1 * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)
        elif kind == "bn_rm":
            v = (0.05 * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)
        elif kind == "bn_rv":
            v = (1.0 + 0.2 * rng.random(shape, dtype=np.float32)).astype(np.float32)
        elif kind == "bn_nbt":
            st[name] = torch.zeros((), dtype=torch.int64)
            continue
        elif kind == "buf_mean":
            v = np.asarray(IMG_MEAN, dtype=np.float32).reshape(shape)
        elif kind == "buf_std":
            v = np.asarray(IMG_STD, dtype=np.float32).reshape(shape)
        else:  # pragma: no cover
            raise KeyError(kind)
        st[name] = torch.from_numpy(np.ascontiguousarray(v))
    return st


def trainable_names(state) -> List[str]:
    return [k for k in state if not (k.endswith("running_mean") or k.endswith("running_var")
                                     or k.endswith("num_batches_tracked")
                                     or k.endswith("img_mean") or k.endswith("img_std"))]


def param_groups(state) -> Tuple[List[str], List[str]]:
    """(random_init, fine_tune) parameter names -- the two ADAM groups of
    utils/init_trainer.py:169-177 via network/weathernet.py:100-104 and
    network/backbone/resnet_pyramid.py:187-188,:239-246.  The segmentation
    head is in neither (SURVEY.md note N1)."""
    rnd, fine = [], []
    for k in trainable_names(state):
        if k.startswith("segmentation."):
            continue
        body = k[len("feature_extractor."):]
        if body.startswith("upsample_bottlenecks") or body.startswith("upsample_blends"):
            rnd.append(k)
        else:
            fine.append(k)
    return rnd, fine


# --------------------------------------------------------------------------- #
# model forward
# --------------------------------------------------------------------------- #
class BNLog:
    """Records train-mode BN calls so that the reference's activation
    checkpointing side effect (block BNs update their running statistics a
    second time when the segment is recomputed in backward) can
    be replayed: ``replay_recompute`` applies the second update in reverse
    call order, which is the order autograd re-runs the checkpointed segments."""

    def __init__(self):
        self.calls: List[Tuple[str, torch.Tensor, torch.Tensor]] = []

    def replay_recompute(self, state):
        for prefix, mean, var_unb in reversed(self.calls):
            _ema(state, prefix, mean, var_unb)
        self.calls.clear()


def _ema(state, prefix, mean, var_unb):
    state[prefix + ".running_mean"].mul_(1 - BN_MOMENTUM).add_(mean, alpha=BN_MOMENTUM)
    state[prefix + ".running_var"].mul_(1 - BN_MOMENTUM).add_(var_unb, alpha=BN_MOMENTUM)
    state[prefix + ".num_batches_tracked"] += 1


def batch_norm(x, state, prefix, training, log: Optional[BNLog] = None, checkpointed=False):
    """nn.BatchNorm2d semantics: batch mean / biased variance
    in training, running-stat EMA with the unbiased variance."""
    w, b = state[prefix + ".weight"], state[prefix + ".bias"]
    if not training:
        return F.batch_norm(x, state[prefix + ".running_mean"], state[prefix + ".running_var"],
                            w, b, False, 0.0, BN_EPS)
    with torch.no_grad():
        n = x.numel() // x.shape[1]
        mean = x.mean(dim=(0, 2, 3))
        var_b = x.var(dim=(0, 2, 3), unbiased=False)
        var_unb = var_b * (n / max(n - 1, 1))
        _ema(state, prefix, mean, var_unb)
        if log is not None and checkpointed:
            log.calls.append((prefix, mean.clone(), var_unb.clone()))
    return F.batch_norm(x, None, None, w, b, True, 0.0, BN_EPS)


def basic_block(x, state, p, stride, has_ds, training, log, ckpt):
    """network/backbone/resnet_pyramid.py:71-89."""
    out = F.conv2d(x, state[p + ".conv1.weight"], None, stride, 1)
    out = F.relu(batch_norm(out, state, p + ".bn1", training, log, ckpt))
    out = F.conv2d(out, state[p + ".conv2.weight"], None, 1, 1)
    out = batch_norm(out, state, p + ".bn2", training, log, ckpt)
    if has_ds:
        res = F.conv2d(x, state[p + ".downsample.0.weight"], None, stride, 0)
        res = batch_norm(res, state, p + ".downsample.1", training, None, False)
    else:
        res = x
    return F.relu(out + res)


def bn_relu_conv(x, state, prefix, k, training, bias=None):
    """network/utils.py:35-49 (_BNReluConv: norm -> relu -> conv)."""
    x = F.relu(batch_norm(x, state, prefix + ".norm", training))
    return F.conv2d(x, state[prefix + ".conv.weight"], bias, 1, k // 2)


def feature_extractor(img, state, training, log: Optional[BNLog] = None, ckpt: bool = False,
                      keep: Optional[dict] = None):
    """network/backbone/resnet_pyramid.py:295-379 (pyramid SwiftNet forward)."""
    fe = "feature_extractor."
    x0 = (img - state[fe + "img_mean"]) / state[fe + "img_std"]        # :303-304
    pyramid = [x0]
    for l in (1, 2):                                                  # :306-314
        pyramid.append(F.interpolate(x0, scale_factor=1 / 2 ** l, mode="bicubic", align_corners=None))
    skips: List[List[torch.Tensor]] = [[] for _ in range(6)]
    for idx, p in enumerate(pyramid):                                 # :318-348
        x = F.conv2d(p, state[fe + "conv1.weight"], None, 2, 3)
        x = F.relu(batch_norm(x, state, fe + f"bn1_{idx}", training))
        x = F.max_pool2d(x, 3, 2, 1)
        inpl = 64
        for li, (lname, planes, stride) in enumerate(LAYERS):
            for b in range(2):
                has_ds = b == 0 and (stride != 1 or inpl != planes)
                x = basic_block(x, state, fe + f"{lname}.{b}", stride if b == 0 else 1, has_ds,
                                training, log, ckpt)
            inpl = planes
            skips[idx + li].append(F.conv2d(x, state[fe + f"upsample_bottlenecks{li + 1}.weight"]))
    skips = skips[::-1]                                               # :361
    x = skips[0][0]
    if keep is not None:
        keep["skips_0"] = x
    for i in range(1, 6):                                             # :372-376, network/utils.py:92-102
        skip = 0
        for s in skips[i]:
            skip = skip + s
        x = F.interpolate(x, skip.shape[2:4], mode="bilinear", align_corners=False)
        x = x + skip
        x = bn_relu_conv(x, state, fe + f"upsample_blends{i}.blend_conv", 3, training)
    return x


def weathernet_forward(img, state, training=True, return_supcon_feature=False,
                       log: Optional[BNLog] = None, ckpt: bool = False):
    """network/weathernet.py:76-98.  Returns the reference 4-tuple."""
    fine_feat = feature_extractor(img, state, training, log, ckpt)
    if return_supcon_feature:
        bsz = fine_feat.shape[0] // 2
        fine_feat0 = fine_feat[:bsz]
    else:
        fine_feat0 = fine_feat
    before = bn_relu_conv(fine_feat0, state, "segmentation", 1, training,
                          bias=state["segmentation.conv.bias"])
    seg = F.interpolate(before, img.shape[2:], mode="bilinear", align_corners=False)
    return seg, before, fine_feat, fine_feat0


def weather_classifier(x, fc_w, fc_b):
    """network/classifier.py:24-32."""
    return F.linear(x.mean(dim=(2, 3)), fc_w, fc_b)


# --------------------------------------------------------------------------- #
# losses (utils/loss.py)
# --------------------------------------------------------------------------- #
def boundary_aware_focal_loss(logits, target, ldw, class_weight, gamma=0.5, ignore_id=255,
                              variant="full"):
    """utils/loss.py:39-80.  Mutates ``target`` in place like the reference
    (:43).  variant: full | plain_focal | no_class_weights | no_EDT (:65-72)."""
    if logits.shape[-2:] != target.shape[-2:]:
        logits = F.interpolate(logits, target.shape[-2:], mode="bilinear", align_corners=False)
    target[target == ignore_id] = 0
    n = (ldw > 0).sum()
    if int(n) <= 0:
        return torch.zeros((0,), requires_grad=True).sum()
    c = logits.shape[1]
    x = logits.permute(0, 2, 3, 1).reshape(-1, c)
    t = target.reshape(-1, 1)
    alphas = ldw.reshape(-1)
    w = class_weight[t].reshape(-1)
    logpt = F.log_softmax(x.float(), dim=-1).gather(1, t).reshape(-1)
    pt = logpt.detach().exp()
    mod = torch.exp(gamma * (1 - pt))
    if variant == "plain_focal":
        loss = -mod * logpt
    elif variant == "no_class_weights":
        loss = -alphas * mod * logpt
    elif variant == "no_EDT":
        loss = -w * mod * logpt
    else:
        loss = -w * alphas * mod * logpt
    return loss.sum() / n


def cross_entropy_loss(logits, target, ignore_index=255):
    """nn.CrossEntropyLoss(ignore_index=255), utils/init_trainer.py:223."""
    return F.cross_entropy(logits, target, ignore_index=ignore_index)


def _masked_contrast_rows(feat, temperature):
    """Shared head of both contrastive losses: S = C C^T / T, minus the detached
    row max, then row L2-normalised (utils/loss.py:175-180,:194 and :361-366)."""
    s = torch.matmul(feat, feat.T) / temperature
    s = s - s.max(dim=1, keepdim=True)[0].detach()
    return F.normalize(s)


def supcon_loss(features, proj, class_labels=None, temperature=0.07, base_temperature=0.07):
    """utils/loss.py:114-205.  ``proj`` = (w1, b1, w2, b2) of the projection MLP
    (:105-109).  class_labels None -> SimCLR (eye mask)."""
    w1, b1, w2, b2 = proj
    f = features.mean(dim=(2, 3))
    bsz = f.shape[0] // 2
    f = torch.stack([f[:bsz], f[bsz:]], dim=1)                        # [B,2,C]
    f = F.linear(F.relu(F.linear(f, w1, b1)), w2, b2)
    if class_labels is None:
        mask = torch.eye(bsz, dtype=torch.float32)
    else:
        lab = class_labels.reshape(-1, 1)
        if lab.shape[0] != bsz:
            raise ValueError("Num of labels does not match num of features")
        mask = torch.eq(lab, lab.T).float()
    contrast = torch.cat(torch.unbind(f, dim=1), dim=0)               # [2B,128], view-major
    logits = _masked_contrast_rows(contrast, temperature)
    mask = mask.repeat(2, 2)
    logits_mask = 1.0 - torch.eye(2 * bsz)
    mask = mask * logits_mask
    exp_logits = torch.exp(logits) * logits_mask
    log_prob = logits - torch.log(exp_logits.sum(1, keepdim=True))
    mean_log_prob_pos = (mask * log_prob).sum(1) / mask.sum(1)
    loss = -(temperature / base_temperature) * mean_log_prob_pos
    return loss.view(2, bsz).mean()


def hard_anchor_sampling_indices(labels_lr, predict, max_samples=1024, max_views=2, ignore_label=255,
                                 generator: Optional[torch.Generator] = None):
    """Index-only restatement of utils/loss.py:264-337.

    labels_lr, predict: int64 [B, h*w].  Returns (img_idx [T], cls [T],
    pix_idx [T, n_view]) or None when no class qualifies (:287-288).  Consumes
    the CPU generator exactly like the reference: per (image, class) one
    ``randperm(num_hard)`` then one ``randperm(num_easy)`` (:327-330)."""
    bsz = labels_lr.shape[0]
    classes = []
    total = 0
    for ii in range(bsz):
        y = labels_lr[ii]
        cs = [int(c) for c in torch.unique(y) if int(c) != ignore_label]
        cs = [c for c in cs if int((y == c).sum()) > max_views]
        classes.append(cs)
        total += len(cs)
    if total == 0:
        return None
    n_view = min(max_samples // total, max_views)
    img_idx, cls_out, pix = [], [], []
    for ii in range(bsz):
        y, p = labels_lr[ii], predict[ii]
        for c in classes[ii]:
            hard = ((y == c) & (p != c)).nonzero().reshape(-1)
            easy = ((y == c) & (p == c)).nonzero().reshape(-1)
            nh, ne = hard.numel(), easy.numel()
            if nh >= n_view / 2 and ne >= n_view / 2:
                kh = n_view // 2
                ke = n_view - kh
            elif nh >= n_view / 2:
                ke = ne
                kh = n_view - ke
            elif ne >= n_view / 2:
                kh = nh
                ke = n_view - kh
            else:
                raise Exception("this shoud be never touched! {} {} {}".format(nh, ne, n_view))
            perm = torch.randperm(nh, generator=generator)
            hsel = hard[perm[:kh]]
            perm = torch.randperm(ne, generator=generator)
            esel = easy[perm[:ke]]
            img_idx.append(ii)
            cls_out.append(c)
            pix.append(torch.cat([hsel, esel]))
    return (torch.tensor(img_idx, dtype=torch.int64), torch.tensor(cls_out, dtype=torch.int64),
            torch.stack(pix))


def pixel_contrastive(x_, y_, temperature=0.07, base_temperature=0.07):
    """utils/loss.py:339-389 on sampled anchors x_ [T, n_view, C], y_ [T]."""
    t, n_view = x_.shape[0], x_.shape[1]
    y = y_.reshape(-1, 1)
    mask = torch.eq(y, y.T).float()
    contrast = torch.cat(torch.unbind(x_, dim=1), dim=0)              # [A, C], view-major
    logits = _masked_contrast_rows(contrast, temperature)
    mask = mask.repeat(n_view, n_view)
    neg_mask = 1 - mask
    a = t * n_view
    mask = mask * (1.0 - torch.eye(a))
    neg = (torch.exp(logits) * neg_mask).sum(1, keepdim=True)
    log_prob = logits - torch.log(torch.exp(logits) + neg)
    mean_log_prob_pos = (mask * log_prob).sum(1) / mask.sum(1)
    return (-(temperature / base_temperature) * mean_log_prob_pos).mean()


def downsample_labels_nearest(labels, h, w):
    """utils/loss.py:400-403."""
    return F.interpolate(labels.unsqueeze(1).float(), (h, w), mode="nearest").squeeze(1).long()


def pixel_contrast_loss(feats, labels, predict_logits, generator=None, return_indices=False):
    """utils/loss.py:391-415."""
    b, c, h, w = feats.shape
    predict = predict_logits.max(1)[1].reshape(b, -1)
    lab = downsample_labels_nearest(labels, h, w).reshape(b, -1)
    x = feats.permute(0, 2, 3, 1).reshape(b, h * w, c)
    sel = hard_anchor_sampling_indices(lab, predict, generator=generator)
    if sel is None:
        raise AttributeError("'NoneType' object has no attribute 'shape'")   # loss.py:341 on (None, None)
    img_idx, cls, pix = sel
    x_ = x[img_idx.unsqueeze(1), pix]                                 # [T, n_view, C]
    loss = pixel_contrastive(x_, cls.float())
    if return_indices:
        return loss, sel
    return loss


# --------------------------------------------------------------------------- #
# train step (trainer.py:62-215) and optimizer (utils/init_trainer.py:169-177)
# --------------------------------------------------------------------------- #
CRITERIA = ("supcon_focal", "supcon_simclr_focal", "pixelcontrast_focal", "supcon_pixelcontrast_focal",
            "supcon_simclr_pixelcontrast_focal", "crossentropy", "supcon_crossentropy",
            "supcon_simclr_cross_entropy", "focal")


def combine_losses(criterion, batch_size, seg=None, supcon=None, simclr=None, pixel=None, ce=None):
    """trainer.py:116-203 (including the supcon_simclr_cross_entropy quirk that
    adds the still-zero supcon term)."""
    if criterion == "supcon_focal":
        return supcon * 1 / batch_size + seg * 1.2
    if criterion == "supcon_simclr_focal":
        return simclr * 1 / batch_size + seg * 1.2
    if criterion == "pixelcontrast_focal":
        return pixel * 1 / batch_size + seg * 1.2
    if criterion == "supcon_pixelcontrast_focal":
        return 1 / batch_size * (supcon + pixel) + seg * 1.2
    if criterion == "supcon_simclr_pixelcontrast_focal":
        return 1 / batch_size * (simclr + pixel) + seg * 1.2
    if criterion == "crossentropy":
        return ce
    if criterion == "supcon_crossentropy":
        return ce + supcon
    if criterion == "supcon_simclr_cross_entropy":
        return ce + torch.tensor([0.])
    return seg


def train_step_losses(state, proj, img, labels, ldw, weather, class_weight, criterion, batch_size,
                      generator=None, log: Optional[BNLog] = None, ckpt=True):
    """One forward of trainer.py:62-203.  ``img`` already holds both crops for
    supcon criteria (:66-72).  Returns dict of scalar losses and tensors."""
    supcon_flag = "supcon" in criterion
    seg, before, fine_feat, fine_feat0 = weathernet_forward(img, state, True, supcon_flag, log, ckpt)
    out = {"seg_logits": seg, "before": before, "fine_feat": fine_feat}
    sup = sim = pix = segl = ce = None
    if criterion in ("supcon_focal", "supcon_pixelcontrast_focal", "supcon_crossentropy"):
        sup = supcon_loss(fine_feat, proj, weather)
    if criterion in ("supcon_simclr_focal", "supcon_simclr_pixelcontrast_focal", "supcon_simclr_cross_entropy"):
        sim = supcon_loss(fine_feat, proj, None)
    if "pixelcontrast" in criterion:
        pix, sel = pixel_contrast_loss(fine_feat0, labels, before, generator, return_indices=True)
        out["anchors"] = sel
    if "crossentropy" in criterion or "cross_entropy" in criterion:
        ce = cross_entropy_loss(seg, labels)
    else:
        segl = boundary_aware_focal_loss(seg, labels, ldw, class_weight)
    total = combine_losses(criterion, batch_size, segl, sup, sim, pix, ce)
    out.update(supcon=sup, simclr=sim, pixel=pix, seg=segl, ce=ce, total=total.reshape(()))
    return out


class Adam:
    """torch.optim.Adam restated (L2 weight decay folded into the gradient,
    bias-corrected moments), groups per utils/init_trainer.py:169-177."""

    def __init__(self, state, lr=4e-4, weight_decay=1e-4, betas=(0.9, 0.99), eps=1e-8, fine_tune_factor=4):
        rnd, fine = param_groups(state)
        self.groups = [dict(names=rnd, lr=lr, wd=weight_decay),
                       dict(names=fine, lr=lr / fine_tune_factor, wd=weight_decay / fine_tune_factor)]
        self.betas, self.eps, self.t = betas, eps, 0
        self.m = {k: torch.zeros_like(state[k]) for g in self.groups for k in g["names"]}
        self.v = {k: torch.zeros_like(state[k]) for g in self.groups for k in g["names"]}

    @torch.no_grad()
    def step(self, state, grads):
        self.t += 1
        b1, b2 = self.betas
        bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        for g in self.groups:
            for k in g["names"]:
                if grads.get(k) is None:
                    continue
                gr = grads[k] + g["wd"] * state[k]
                self.m[k].mul_(b1).add_(gr, alpha=1 - b1)
                self.v[k].mul_(b2).addcmul_(gr, gr, value=1 - b2)
                denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
                state[k].addcdiv_(self.m[k], denom, value=-g["lr"] / bc1)


def train_step(state, proj, opt: Optional[Adam], img, labels, ldw, weather, class_weight, criterion,
               batch_size, generator=None):
    """Full trainer.py:62-215 iteration on CPU: forward, backward, Adam step,
    plus the checkpoint-recompute BN side effect.  Returns (losses, grads)."""
    names = trainable_names(state)
    for k in names:
        state[k].requires_grad_(True)
    pw = [p.requires_grad_(True) for p in proj]
    log = BNLog()
    out = train_step_losses(state, pw, img, labels, ldw, weather, class_weight, criterion, batch_size,
                            generator, log, ckpt=True)
    params = [state[k] for k in names] + list(pw)
    g = torch.autograd.grad(out["total"], params, allow_unused=True)
    for k in names:
        state[k].requires_grad_(False)
    log.replay_recompute(state)
    grads = {k: gi for k, gi in zip(names, g[:len(names)])}
    grads_proj = list(g[len(names):])
    if opt is not None:
        opt.step(state, grads)
    det = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}
    return det, grads, grads_proj


# --------------------------------------------------------------------------- #
# synthetic inputs
# --------------------------------------------------------------------------- #
def synthetic_batch(b, h, w, seed=0, two_crops=False, cell=32, num_classes=19, weather_num=4):
    """What dataloaders/ would deliver: raw 0-255 image(s), blocky labels with
    an ignore border, label-distance weights in (0,1] (0 on ignore), weather id."""
    rng = np.random.default_rng(seed)
    bm = 2 * b if two_crops else b
    img = rng.random((bm, 3, h, w), dtype=np.float32) * np.float32(255.0)
    gh, gw = -(-h // cell), -(-w // cell)
    cells = rng.integers(0, num_classes, size=(b, gh, gw))
    lab = np.repeat(np.repeat(cells, cell, axis=1), cell, axis=2)[:, :h, :w].astype(np.int64)
    # salt a fraction of pixels with a second class so hard/easy both occur
    salt = rng.random((b, h, w)) < 0.1
    lab = np.where(salt, rng.integers(0, num_classes, size=(b, h, w)), lab)
    bw = max(2, cell // 8)
    lab[:, :bw, :] = 255
    lab[:, :, -bw:] = 255
    ldw = (0.05 + 0.95 * rng.random((b, h, w), dtype=np.float32)).astype(np.float32)
    ldw[lab == 255] = 0.0
    weather = rng.integers(0, weather_num, size=(b, 1)).astype(np.int64)
    freq = rng.random(num_classes) * 0.2
    cw = (1.0 / np.log(1.1 + freq)).astype(np.float32)               # utils/init_trainer.py:204-208
    return (torch.from_numpy(img), torch.from_numpy(lab), torch.from_numpy(ldw),
            torch.from_numpy(weather), torch.from_numpy(cw))


def make_proj(seed=2, dim_in=128, feat_dim=128):
    """SupConLoss.projection parameters (utils/loss.py:105-109) from a stable stream."""
    rng = np.random.default_rng(seed)
    k = 1.0 / math.sqrt(dim_in)
    mk = lambda *s: torch.from_numpy(((rng.random(s, dtype=np.float32) * 2 - 1) * np.float32(k)).astype(np.float32))
    return [mk(dim_in, dim_in), mk(dim_in), mk(feat_dim, dim_in), mk(feat_dim)]
