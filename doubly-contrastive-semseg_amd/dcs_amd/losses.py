"""Drop-in loss modules with the reference's names and call signatures
(utils/loss.py:27-80, :84-205, :208-247, :250-415), backed by HIP kernels.

Each loss is one ``torch.autograd.Function``: forward launches the fused
kernels (which already produce the gradient), backward only rescales it by
the incoming scalar.  Host code that the reference also runs on the host
(class filtering, ``torch.randperm`` on the CPU generator) stays on the host
with identical RNG consumption.
"""
from __future__ import annotations

import os

from abc import ABC
from typing import List, Optional

import torch
import torch.nn as nn

from . import ops


# --------------------------------------------------------------------------- #
# layout helpers
# --------------------------------------------------------------------------- #
def nhwc(x: torch.Tensor) -> torch.Tensor:
    """[N,C,H,W]-shaped tensor -> contiguous [N,H,W,C] (free for channels_last inputs)."""
    ops.require_device(x, "loss input")
    v = x.permute(0, 2, 3, 1).contiguous()
    return v if v.is_floating_point() else v.float()


def _scalar(g: torch.Tensor) -> torch.Tensor:
    return g.detach().reshape(1).contiguous()


_POOL_W = {}


def _upsample_colsum(n_in: int, n_out: int) -> torch.Tensor:
    """sum over outputs o of the bilinear weight with which input i enters o (align_corners=False, fp32 like the kernels)."""
    w = torch.zeros(n_in, dtype=torch.float32)
    scale = torch.tensor(n_in, dtype=torch.float32) / torch.tensor(n_out, dtype=torch.float32)
    s = (scale * (torch.arange(n_out, dtype=torch.float32) + 0.5) - 0.5).clamp_min(0.0)
    i0 = s.floor().long().clamp_max(n_in - 1)
    i1 = i0 + (i0 < n_in - 1).long()
    w1 = s - i0.float()
    w.index_add_(0, i0, 1.0 - w1)
    w.index_add_(0, i1, w1)
    return w


def global_avg_pool(x) -> torch.Tensor:
    """nn.AdaptiveAvgPool2d((1,1)) + flatten on an [N,C,H,W]-shaped tensor -> [N,C].  For a LazyUpsampled handle the
    mean of the (virtual) upsampled map is a weighted sum of the low-resolution map: no upsampling."""
    if isinstance(x, LazyUpsampled):
        v = nhwc(x.lowres.detach())
        N, hf, wf, Cc = v.shape
        key = (hf, wf) + x.size + (str(v.device),)
        if key not in _POOL_W:
            wy, wx = _upsample_colsum(hf, x.size[0]), _upsample_colsum(wf, x.size[1])
            w4 = torch.zeros((hf * wf, 4), dtype=torch.float32)
            w4[:, 0] = (wy[:, None] * wx[None, :]).reshape(-1) / float(x.size[0] * x.size[1])
            _POOL_W[key] = w4.to(v.device)
        out = torch.empty((N, 4, Cc), device=v.device, dtype=v.dtype)
        for n in range(N):
            ops.linear_wgrad(v[n].reshape(hf * wf, Cc), _POOL_W[key].to(v.dtype), out[n])
        return out[:, 0, :].contiguous()
    v = nhwc(x.detach())
    N, H, W, Cc = v.shape
    return ops.colsum(v.reshape(N * H * W, Cc), B=N, scale=1.0 / (H * W))[:, 0, :].contiguous()


# --------------------------------------------------------------------------- #
# segmentation losses
# --------------------------------------------------------------------------- #
class _SegLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, ldw, cw, mode, gamma, ignore, reduce_fn=None):
        lg = logits.detach()
        if not lg.is_contiguous():
            lg = lg.contiguous()
        out, grad = ops.seg_loss(lg, target, ldw, cw, mode, gamma, ignore)
        if reduce_fn is not None:
            # data parallel: loss = (sum over all ranks) / (count over all ranks); see dcs_amd/dist.py
            out = reduce_fn(out)
        ctx.grad, ctx.out = grad, out
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        grad = ctx.grad
        if grad is None:
            raise RuntimeError("seg loss backward called twice")
        ctx.grad = None
        ops.scale_inplace(grad, _scalar(g), ctx.out[2:3])
        return grad, None, None, None, None, None, None, None


class _UpsampleNCHWFn(torch.autograd.Function):
    """upsample = F.interpolate(bilinear, align_corners=False) (utils/loss.py:5) on logits."""

    @staticmethod
    def forward(ctx, x, size):
        v = nhwc(x.detach())
        ctx.in_hw = v.shape[1:3]
        ctx.cs = v.shape[3]
        return ops.upsample_to_nchw(v, v.shape[3], size[0], size[1])

    @staticmethod
    def backward(ctx, g):
        gx = ops.upsample_to_nchw_bwd(g.contiguous(), ctx.in_hw[0], ctx.in_hw[1], ctx.cs)
        return gx.permute(0, 3, 1, 2), None


class LazyLogits(torch.Tensor):
    """``pred_segmap`` = F.interpolate(pred_segmap_beforeup, (H, W), bilinear, align_corners=False) (network/utils.py:8,
    weathernet.py:93) that is only built if somebody touches it.

    The reference materialises the full-resolution logits [B,19,H,W] (2.55 GB at C3) in the model and the focal / CE
    loss reads them back, then autograd folds a gradient of the same size.  The training loop hands ``left_seg`` to the
    criterion and to nothing else (trainer.py:121-201), and the criteria here (BoundaryAwareFocalLoss, FocalLoss2,
    SemsegCrossEntropy) recognise this handle and run ONE fused kernel on the low-resolution logits instead
    (dcs_seg_loss_fused): upsampling, log-softmax, loss and the adjoint of the upsampling, nothing full-resolution in
    HBM.  Everything else -- ``left_seg.detach().max(1)``, slicing, torch's own ``nn.CrossEntropyLoss`` (the reference's
    ce_criterion, utils/init_trainer.py:223), ``.cpu()`` -- triggers ``__torch_function__``, which materialises the
    dense tensor once (differentiably: gradients flow back into the low-resolution logits) and carries on with it, so
    any caller sees an ordinary [B,C,H,W] tensor.  Shape / dtype / device queries do not materialise."""

    @staticmethod
    def __new__(cls, before_raw, num_classes, size):
        B = before_raw.shape[0]
        r = torch.Tensor._make_wrapper_subclass(cls, (B, num_classes, int(size[0]), int(size[1])), dtype=before_raw.dtype,
                                                device=before_raw.device, requires_grad=False)
        r._before_raw = before_raw                  # [B,h,w,cs] NHWC, autograd output of the model node
        r._nc, r._size, r._dense = num_classes, (int(size[0]), int(size[1])), None
        return r

    _META = None

    def materialize(self):
        if self._dense is None:
            v = self._before_raw[..., :self._nc].permute(0, 3, 1, 2)
            self._dense = _UpsampleNCHWFn.apply(v, self._size)
        return self._dense

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if cls._META is None:
            T = torch.Tensor
            cls._META = {T.shape.__get__, T.dtype.__get__, T.device.__get__, T.size, T.dim, T.ndim.__get__, T.is_cuda.__get__,
                         T.requires_grad.__get__, T.is_floating_point, T.numel, T.layout.__get__, T.ndimension,
                         T.grad_fn.__get__, T.is_leaf.__get__, T.element_size}
        if func in cls._META:
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        from torch.utils._pytree import tree_map
        dense = lambda a: a.materialize() if isinstance(a, LazyLogits) else a
        with torch._C.DisableTorchFunctionSubclass():
            return func(*tree_map(dense, args), **tree_map(dense, kwargs))

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        # only reached by callers that bypass __torch_function__ (C++ entry points): same answer, dense tensor
        from torch.utils._pytree import tree_map
        dense = lambda a: a.materialize() if isinstance(a, LazyLogits) else a
        return func(*tree_map(dense, args), **tree_map(dense, kwargs or {}))

    def __repr__(self):
        return f"LazyLogits(shape={tuple(self._size)}, materialized={self._dense is not None})"


class _SegLossFusedFn(torch.autograd.Function):
    """Loss on a LazyLogits handle: one kernel from the low-resolution NHWC logits to the loss and their gradient."""

    @staticmethod
    def forward(ctx, before_raw, nc, target, ldw, cw, mode, gamma, ignore, reduce_fn=None):
        out, grad = ops.seg_loss_fused(before_raw.detach(), nc, target, ldw, cw, mode, gamma, ignore)
        if reduce_fn is not None:
            out = reduce_fn(out)
        ctx.grad, ctx.out = grad, out
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        grad = ctx.grad
        if grad is None:
            raise RuntimeError("seg loss backward called twice")
        ctx.grad = None
        ops.scale_inplace(grad, _scalar(g), ctx.out[2:3])
        return grad, None, None, None, None, None, None, None, None


def _seg_loss(input, target, ldw, cw, mode, gamma, ignore, reduce_fn):
    """Focal / CE loss of ``input`` = full-resolution logits, low-resolution logits (upsampled first like
    utils/loss.py:41-42) or a LazyLogits handle (fused: never upsampled in memory)."""
    target = _prep_target(target)
    if isinstance(input, LazyLogits) and input._dense is None:
        raw = input._before_raw
        if raw.is_contiguous() and ops.seg_loss_fused_ok(raw.shape[1], raw.shape[2], target.shape[-2], target.shape[-1], input._nc) \
                and tuple(target.shape[-2:]) == input._size:
            return _SegLossFusedFn.apply(raw, input._nc, target, ldw, cw, mode, gamma, ignore, reduce_fn)
    if isinstance(input, LazyLogits):
        input = input.materialize()
    if input.shape[-2:] != target.shape[-2:]:
        input = _UpsampleNCHWFn.apply(input, tuple(target.shape[-2:]))
    return _SegLossFn.apply(input, target, ldw, cw, mode, gamma, ignore, reduce_fn)


def _prep_target(target):
    ops.require_device(target, "target")
    if target.dtype != torch.int64:
        raise RuntimeError("target must be an int64 tensor")
    if not target.is_contiguous():
        raise RuntimeError("target must be contiguous (it is rewritten in place like the reference)")
    return target


class BoundaryAwareFocalLoss(nn.Module):
    """utils/loss.py:27-80.  ``target`` is rewritten in place (255 -> 0) like the reference."""

    def __init__(self, gamma=0, num_classes=19, ignore_id=19, print_each=20, weight=None, device=None, opts=None):
        super().__init__()
        self.num_classes = num_classes
        self.ignore_id = ignore_id
        self.print_each = print_each
        self.step_counter = 0
        self.gamma = gamma
        self.weight = weight
        self.device = device
        self.opts = opts
        self._cw = None
        self.dist_reduce = None            # set by dcs_amd.dist.DataParallelStep

    def _class_weight(self, dev):
        if self.weight is None:
            return None
        if self._cw is None or self._cw.device != dev or self._cw.dtype != self.weight.dtype:
            self._cw = self.weight.to(dev).contiguous()
        return self._cw

    def _mode(self):
        o = self.opts
        if getattr(o, "criterion", None) == "plain_focal":
            return "plain_focal"
        if getattr(o, "no_class_weights", False):
            return "no_class_weights"
        if getattr(o, "no_EDT", False):
            return "no_EDT"
        return "full"

    def forward(self, input, target, batch, **kwargs):
        ldw = batch["label_distance_weight"].to(input.device, input.dtype).contiguous()
        loss = _seg_loss(input, target, ldw, self._class_weight(input.device), self._mode(), float(self.gamma),
                         int(self.ignore_id), self.dist_reduce)
        self.step_counter += 1
        return loss


class FocalLoss2(BoundaryAwareFocalLoss):
    """utils/loss.py:208-247: the un-weighted variant (-exp(gamma(1-pt)) logpt / N)."""

    def __init__(self, gamma=0, num_classes=19, ignore_id=19, print_each=20, weight=None, device=None):
        super().__init__(gamma, num_classes, ignore_id, print_each, weight, device, opts=None)

    def _mode(self):
        return "plain_focal"


class SemsegCrossEntropy(nn.Module):
    """nn.CrossEntropyLoss(ignore_index) on the HIP kernel (utils/loss.py:6-24, init_trainer.py:223)."""

    def __init__(self, num_classes=19, ignore_id=255, print_each=20):
        super().__init__()
        self.num_classes, self.ignore_id, self.step_counter, self.print_each = num_classes, ignore_id, 0, print_each
        self.dist_reduce = None

    def forward(self, logits, labels, **kwargs):
        self.step_counter += 1
        return _seg_loss(logits, labels, None, None, "ce", 0.0, int(self.ignore_id), self.dist_reduce)


# --------------------------------------------------------------------------- #
# image-level contrastive loss
# --------------------------------------------------------------------------- #
_ident_bn = {}


def _identity_bn(Cc, dev, dtype=torch.float32):
    key = (Cc, dev, dtype)
    if key not in _ident_bn:
        t = torch.zeros((4, Cc), device=dev, dtype=dtype)
        t[0].fill_(1.0)
        _ident_bn[key] = t
    return _ident_bn[key]


class _SupConFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, labels, w1, b1, w2, b2, temperature, row_gather=None, cap=0, mask=None):
        v = nhwc(features.detach())
        N, H, W, Cc = v.shape
        pooled = ops.colsum(v.reshape(N * H * W, Cc), B=N, scale=1.0 / (H * W))[:, 0, :].contiguous()
        w1c, w2c = w1.detach().contiguous(), w2.detach().contiguous()
        h1 = ops.linear(pooled, w1c, b1.detach().contiguous())
        a1 = ops.bn_act(h1, _identity_bn(h1.shape[1], h1.device, h1.dtype), relu=True)
        f = ops.linear(a1, w2c, b2.detach().contiguous())
        if row_gather is None:
            loss, dF = ops.contrast_fwd_bwd(f, labels, 1, temperature, mask=mask)
        else:                                   # global-batch denominator: every rank evaluates all rows
            buf, start = row_gather(f, labels, cap)
            Cf = f.shape[1]
            loss, dF_all = ops.contrast_fwd_bwd(buf[:, :Cf], buf[:, Cf], 1, temperature)
            dF = dF_all[start:start + f.shape[0]].contiguous()
        ctx.saved = (pooled, a1, dF, w1c, w2c, (N, H, W, Cc))
        return loss.reshape(()).clone()

    @staticmethod
    def backward(ctx, g):
        pooled, a1, dF, w1c, w2c, (N, H, W, Cc) = ctx.saved
        ctx.saved = None
        ops.scale_inplace(dF, _scalar(g))
        dw2 = torch.empty_like(w2c)
        ops.linear_wgrad(a1, dF, dw2)
        db2 = ops.colsum(dF)[0, 0].contiguous()
        da1 = ops.linear(dF, ops.transpose(w2c))
        dh1 = ops.relu_bwd(da1, a1)
        dw1 = torch.empty_like(w1c)
        ops.linear_wgrad(pooled, dh1, dw1)
        db1 = ops.colsum(dh1)[0, 0].contiguous()
        dpool = ops.linear(dh1, ops.transpose(w1c))
        gfeat = torch.empty((N, H, W, Cc), device=dF.device, dtype=dF.dtype)
        ops.add_rowvec_bcast(gfeat, dpool, 1.0 / (H * W), accumulate=False)
        return gfeat.permute(0, 3, 1, 2), None, dw1, db1, dw2, db2, None, None, None, None


class SupConLoss(nn.Module):
    """utils/loss.py:84-205 (Supervised Contrastive / SimCLR on pooled, projected features)."""

    def __init__(self, temperature=0.07, contrast_mode="all", base_temperature=0.07, weight=None, device=None,
                 opts=None):
        super().__init__()
        self.temperature = temperature
        self.base_temperature = base_temperature
        self.device = device
        self.weight = weight
        self.opts = opts
        feat_dim = 128
        dim_in = 2048 if getattr(opts, "deeplab", False) else 128
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.projection = nn.Sequential(nn.Linear(dim_in, dim_in), nn.ReLU(inplace=True),
                                        nn.Linear(dim_in, feat_dim)).to(self.device)
        self.contrast_mode = "all"
        self.row_gather = None             # set by dcs_amd.dist.DataParallelStep
        self.gather_cap = 0                # rows every rank contributes to the all-gather (2 x its batch)

    def forward(self, features, class_labels=None, mask=None):
        if features.dim() != 4:
            raise ValueError("`features` needs to be [2*bsz, C, H, W]")
        bsz = features.shape[0] // 2
        if class_labels is not None and mask is not None:
            raise ValueError("Cannot define both `labels` and `mask`")
        dev = features.device
        if mask is not None:
            # explicit contrastive mask [bsz, bsz], mask_ij = 1 if sample j is a positive of sample i; may be asymmetric
            # (utils/loss.py:148-159).  Single-process only: the reference has no notion of a mask across ranks.
            if self.row_gather is not None:
                raise ValueError("an explicit `mask` cannot be combined with the data-parallel row gather")
            mask = mask.to(dev, torch.float32)
            if mask.shape != (bsz, bsz):
                raise ValueError("`mask` needs to be [bsz, bsz]")
            lab = torch.arange(bsz, device=dev, dtype=features.dtype)
        elif class_labels is None:
            lab = torch.arange(bsz, device=dev, dtype=features.dtype)
            if self.row_gather is not None:           # instance ids must be unique across ranks in the gathered set
                lab = lab + float(self.row_gather.instance_offset())
        else:
            lab = class_labels.contiguous().view(-1).to(dev, features.dtype)
            if lab.shape[0] != bsz:
                raise ValueError("Num of labels does not match num of features")
        lab2 = lab.repeat(2).contiguous()                       # mask.repeat(anchor_count, contrast_count)
        p = self.projection
        if self.temperature != self.base_temperature:
            raise NotImplementedError("temperature != base_temperature")
        return _SupConFn.apply(features, lab2, p[0].weight, p[0].bias, p[2].weight, p[2].bias,
                               float(self.temperature), self.row_gather, int(self.gather_cap or 2 * bsz), mask)


# --------------------------------------------------------------------------- #
# pixel-level contrastive loss
# --------------------------------------------------------------------------- #
def _contrast_rows(X, y, temperature, row_gather, cap):
    """Pixel-contrast loss of the sampled rows X [A,C] (A may be 0 on a data-parallel rank whose shard holds no class
    with enough pixels: it then contributes only padding to the gather and receives a zero gradient)."""
    if row_gather is None:
        return ops.contrast_fwd_bwd(X, y, 0, temperature)
    buf, start = row_gather(X, y, cap)
    Cc = X.shape[1]
    loss, dX_all = ops.contrast_fwd_bwd(buf[:, :Cc], buf[:, Cc], 0, temperature)
    return loss, dX_all[start:start + X.shape[0]].contiguous()


def _row_sink(feats):
    """The autograd node that produced ``feats`` if it accepts the gradient of a few rows directly (model._SwiftNetFn for
    its fine_feat0 output, reached through the NHWC -> NCHW permute of WeatherNet.forward), else None.  The pixel
    contrast touches <= 608 of the 2 million pixels of fine_feat0: handing the node those rows saves a 1 GB zero fill and
    a 3 GB add per step at C3; any other producer gets the dense gradient."""
    if os.environ.get("DCS_SPARSE_FF0", "1") == "0":
        return None
    gf = getattr(feats, "grad_fn", None)
    if gf is None or type(gf).__name__ != "PermuteBackward0" or len(gf.next_functions) != 1:
        return None
    node, index = gf.next_functions[0]
    fwd = getattr(node, "_forward_cls", None)           # custom autograd.Function nodes name their Function class
    if fwd is None or not hasattr(fwd, "accept_rows") or index != getattr(fwd, "rows_output", -1):
        return None
    return node


class _PixelContrastFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, rowidx, y, temperature, row_gather=None, cap=0, sink=None):
        v = nhwc(feats.detach())
        N, H, W, Cc = v.shape
        if rowidx.numel():
            X = ops.gather_rows(v.reshape(N * H * W, Cc), rowidx)
        else:
            X = torch.empty((0, Cc), device=v.device, dtype=v.dtype)
        loss, dX = _contrast_rows(X, y, temperature, row_gather, cap)
        ctx.saved = (dX, rowidx, (N, H, W, Cc), sink)
        return loss.reshape(()).clone()

    @staticmethod
    def backward(ctx, g):
        dX, rowidx, (N, H, W, Cc), sink = ctx.saved
        ctx.saved = None
        if rowidx.numel():
            ops.scale_inplace(dX, _scalar(g))
        if sink is not None:
            # the rows go to the producer directly; autograd gets a zero that occupies no memory (and stays correct if it
            # is summed with another consumer's gradient)
            if rowidx.numel():
                sink._forward_cls.accept_rows(sink, rowidx, dX)
            return dX.new_zeros(()).expand(N, Cc, H, W), None, None, None, None, None, None
        gfeat = torch.zeros((N, H, W, Cc), device=dX.device, dtype=dX.dtype)
        if rowidx.numel():
            ops.scatter_add_rows(dX, rowidx, gfeat)
        return gfeat.permute(0, 3, 1, 2), None, None, None, None, None, None


class LazyUpsampled:
    """``F.interpolate(lowres, size, mode="bilinear", align_corners=False)`` that is never materialised
    (SURVEY.md 8(f) rank 4).  network/utils.py:190 upsamples DeepLab's 2048-channel feature to the logits' resolution
    (4.3 GB per 4 images at 1024x2048, plus the same again for its gradient) only so that PixelContrastLoss can read
    <= 608 pixels of it; bilinear interpolation is linear per pixel, so those rows are interpolated on demand and the
    gradient goes straight back to the low-resolution map.  Quacks like the tensor for what the loss needs."""

    def __init__(self, lowres: torch.Tensor, size):
        self.lowres = lowres                      # logical NCHW [B,C,hf,wf] (autograd-connected)
        self.size = (int(size[0]), int(size[1]))

    @property
    def shape(self):
        return torch.Size((self.lowres.shape[0], self.lowres.shape[1]) + self.size)

    @property
    def device(self):
        return self.lowres.device

    @property
    def dtype(self):
        return self.lowres.dtype


class _PixelContrastLazyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lowres, rowidx, y, temperature, row_gather, size, cap=0):
        v = nhwc(lowres.detach())
        if rowidx.numel():
            X = ops.gather_rows_bilinear(v, rowidx, size[0], size[1])
        else:
            X = torch.empty((0, v.shape[-1]), device=v.device, dtype=v.dtype)
        loss, dX = _contrast_rows(X, y, temperature, row_gather, cap)
        ctx.saved = (dX, rowidx, tuple(v.shape), size)
        return loss.reshape(()).clone()

    @staticmethod
    def backward(ctx, g):
        dX, rowidx, shape, size = ctx.saved
        ctx.saved = None
        gfeat = torch.zeros(shape, device=dX.device, dtype=dX.dtype)
        if rowidx.numel():
            ops.scale_inplace(dX, _scalar(g))
            ops.scatter_rows_bilinear(dX, rowidx, gfeat, size[0], size[1])
        return gfeat.permute(0, 3, 1, 2), None, None, None, None, None, None


def plan_anchor_requests(counts: torch.Tensor, num_classes: int, max_samples=1024, max_views=2):
    """Host half of utils/loss.py:264-337: see _plan_anchor_requests (the statement in torch) for the contract.

    The plan runs in the library's host function dcs_sampler_plan on the state of torch's default CPU generator: the
    reference's torch.randperm(n) calls shuffle 2.1 million elements per step at BASELINE config 3 to keep ~600 of them
    (6-7 ms of host time, the device idle for ~4 ms of it); the first entries of a permutation need only its first draws,
    the rest of the draws just advance the generator (csrc/sampler_host.cpp).  Same anchors, same generator state
    afterwards (tests/test_host_logic_cpu.py).  DCS_SAMPLER_PLAN=torch, a non-default generator layout, or a case the
    library leaves to the general path (n_view = 0, the reference's exception) take _plan_anchor_requests."""
    if os.environ.get("DCS_SAMPLER_PLAN", "lib") != "torch" and max_views <= 8:
        plan = _plan_in_library(counts, num_classes, max_samples, max_views)
        if plan is not NotImplemented:
            return plan
    # torch.randperm on CPU is thread-count independent in its RESULT but, with several intra-op threads, takes
    # 30-150 ms per call for 32K < n < 100K (measured, torch 2.10) instead of 0.3 ms: one intra-op thread
    nt = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        return _plan_anchor_requests(counts, num_classes, max_samples, max_views)
    finally:
        torch.set_num_threads(nt)


_MT_N = 624
_RNG_STATE_BYTES = 5056          # at::CPUGeneratorImplStateLegacy: seed u64, left i32, seeded i32, next u64, state u64[624], ...


def _plan_in_library(counts, num_classes, max_samples, max_views):
    import ctypes as C
    import numpy as np
    from . import lib as _lib
    st = torch.get_rng_state()
    if st.numel() != _RNG_STATE_BYTES:
        return NotImplemented
    raw = st.numpy()
    left, seeded = (int(v) for v in raw[8:16].view(np.int32))
    nxt = int(raw[16:24].view(np.uint64)[0])
    words = raw[24:24 + 8 * _MT_N].view(np.uint64)
    if not seeded or not (1 <= left <= _MT_N) or nxt > _MT_N:
        return NotImplemented
    key = words.astype(np.uint32)
    pos0 = _MT_N if left == 1 else nxt                    # left == 1: the next draw regenerates the block
    pos = C.c_int(pos0)
    cnt = counts.to(torch.int64).contiguous().numpy()
    B = cnt.shape[0]
    req = np.empty((max_samples + max_views, 3), dtype=np.int32)
    cls = np.empty((max_samples,), dtype=np.int32)
    img = np.empty((max_samples,), dtype=np.int32)
    n_view = C.c_int(0)
    T = _lib.load().dcs_sampler_plan(cnt.ctypes.data, B, num_classes, max_samples, max_views, key.ctypes.data, C.byref(pos),
                                     req.ctypes.data, cls.ctypes.data, img.ctypes.data, C.byref(n_view))
    if T == -3:
        return NotImplemented                             # generator untouched
    if T < 0:
        _lib.check(T, "dcs_sampler_plan")
    if T == 0:
        return None
    # write the advanced generator back: after a draw, left + next = 625 (at::mt19937::operator())
    if pos.value != pos0 or not np.array_equal(words, key):
        words[:] = key
        raw[8:12].view(np.int32)[0] = _MT_N + 1 - pos.value
        raw[16:24].view(np.uint64)[0] = pos.value
        torch.set_rng_state(st)
    nv = n_view.value
    return nv, T, req[:T * nv].tolist(), cls[:T].tolist(), img[:T].tolist()


def _plan_anchor_requests(counts: torch.Tensor, num_classes: int, max_samples=1024, max_views=2):
    """Host half of utils/loss.py:264-337 given per-(image, class, hard|easy) pixel counts
    (int tensor [B, C, 2], index 0 = hard, 1 = easy).  Consumes the default CPU generator exactly
    like the reference (randperm(num_hard) then randperm(num_easy) per kept class).

    Returns None if no class qualifies, else (n_view, T, req [T*n_view, 3], cls [T], img [T]) with
    req rows (image, key = 2*class + is_easy, rank) in (t, view) order."""
    cl = counts.tolist()                                  # plain Python ints: no per-element tensor indexing
    B = len(cl)
    classes = [[c for c in range(num_classes) if cl[ii][c][0] + cl[ii][c][1] > max_views] for ii in range(B)]
    total_classes = sum(len(c) for c in classes)
    if total_classes == 0:
        return None
    n_view = min(max_samples // total_classes, max_views)
    req: List[List[int]] = []
    cls: List[int] = []
    img: List[int] = []
    randperm = torch.randperm
    for ii in range(B):
        for c in classes[ii]:
            num_hard, num_easy = cl[ii][c]
            if num_hard >= n_view / 2 and num_easy >= n_view / 2:
                num_hard_keep = n_view // 2
                num_easy_keep = n_view - num_hard_keep
            elif num_hard >= n_view / 2:
                num_easy_keep = num_easy
                num_hard_keep = n_view - num_easy_keep
            elif num_easy >= n_view / 2:
                num_hard_keep = num_hard
                num_easy_keep = n_view - num_hard_keep
            else:
                raise Exception(f"class with fewer than n_view/2 hard and easy pixels: {num_hard} {num_easy} {n_view}")
            # int32 output: the same draws and the same permutation as the reference's int64 call (the generator is consumed
            # per element, whatever the output type; checked against randperm(n) incl. the generator state), 15-25 % faster
            perm = randperm(num_hard, dtype=torch.int32)   # consumed even when nothing is kept, like the reference
            if num_hard_keep > 0:
                for r in perm[:num_hard_keep].tolist():
                    req.append([ii, 2 * c, r])
            perm = randperm(num_easy, dtype=torch.int32)
            if num_easy_keep > 0:
                for r in perm[:num_easy_keep].tolist():
                    req.append([ii, 2 * c + 1, r])
            cls.append(c)
            img.append(ii)
    return n_view, total_classes, req, cls, img


class PixelContrastLoss(nn.Module, ABC):
    """utils/loss.py:250-415."""

    def __init__(self, device=None):
        super().__init__()
        self.device = device
        self.temperature = 0.07
        self.base_temperature = 0.07
        self.ignore_label = 255
        self.max_samples = 1024
        self.max_views = 2
        self.loss_weight = 1
        self.contrast_mode = "all"
        self.last_anchors = None          # (img [T], cls [T], pix [T, n_view]) of the last call, for tests
        self.row_gather = None            # set by dcs_amd.dist.DataParallelStep
        self.gather_cap = 0               # rows every rank contributes to the all-gather (<= max_samples)
        # parity-test hook: (img [T], pix [T, n_view], cls [T]) -- evaluate the loss on THESE anchors instead of sampling
        # (what tests/golden/make_golden.py::pixel_loss does to the reference for its float64 run): decouples the loss /
        # gradient comparison from argmax near-ties of the forward pass that redirect the sampler.  Consumed by one call.
        self.forced_anchors = None

    def _count(self, feats, labels, predict):
        """Device half of the sampler (argmax, nearest label downsample, per-class hard/easy histogram) and the
        asynchronous D2H copy of the [B, C, 2] counts."""
        B, Cf, h, w = feats.shape
        nc = predict.shape[1]
        assert predict.shape[-1] == feats.shape[-1], "{} {}".format(predict.shape, feats.shape)
        pl = predict.detach().permute(0, 2, 3, 1)
        if not (pl.stride(3) == 1 and pl.stride(1) == w * pl.stride(2) and pl.stride(0) == h * w * pl.stride(2)
                and pl.is_floating_point()):
            pl = pl.contiguous()
        cs = pl.stride(2)
        lab = labels.detach()
        if lab.dtype != torch.int64 or not lab.is_contiguous():
            lab = lab.long().contiguous()
        key, hist = ops.anchor_keys_raw(pl, B, h, w, cs, nc, lab, self.ignore_label)
        counts_dev = hist.sum(1).view(B, nc, 2)
        if counts_dev.is_cuda:
            counts = torch.empty(counts_dev.shape, dtype=counts_dev.dtype, pin_memory=True)
            counts.copy_(counts_dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:
            counts, ev = counts_dev, None
        return dict(key=key, hist=hist, counts=counts, event=ev, labels_ptr=labels.data_ptr(), nc=nc)

    def prefetch(self, feats, labels, predict):
        """Optional: launch the counting kernel early (must see the labels BEFORE the focal loss rewrites 255 -> 0,
        trainer.py:151-156).  ``forward`` picks the result up if it is called with the same labels tensor."""
        self._pre = self._count(feats, labels, predict)

    def forward(self, feats, labels=None, predict=None):
        B, Cf, h, w = feats.shape
        pre = getattr(self, "_pre", None)
        self._pre = None
        if pre is None or pre["labels_ptr"] != labels.data_ptr():
            pre = self._count(feats, labels, predict)
        key, hist, nc = pre["key"], pre["hist"], pre["nc"]
        if pre["event"] is not None:
            pre["event"].synchronize()                        # one host wait per step (reference: ~3 per class)
        counts = pre["counts"]
        forced, self.forced_anchors = self.forced_anchors, None
        plan = plan_anchor_requests(counts, nc, self.max_samples, self.max_views) if forced is None else None
        if self.temperature != self.base_temperature:
            raise NotImplementedError("temperature != base_temperature")
        cap = int(self.gather_cap or self.max_samples)
        if forced is not None:
            img, pixv, cls = forced
            pix_t = torch.as_tensor(pixv, dtype=torch.int32).reshape(len(img), -1)          # [T, n_view]
            T, n_view = pix_t.shape
            base = torch.as_tensor(img, dtype=torch.int32).reshape(T, 1) * (h * w)
            pix = pix_t.t().contiguous().to(feats.device)                                   # view-major, like below
            rowidx = (pix_t + base).t().contiguous().view(-1).to(feats.device)
            cls = [float(c) for c in cls]
            y = torch.tensor(cls * n_view, dtype=torch.float32).to(feats.device)
            self.last_anchors = ([int(i) for i in img], cls, pix, n_view)
        elif plan is None:
            if self.row_gather is None:
                raise AttributeError("'NoneType' object has no attribute 'shape'")   # loss.py:341 on (None, None)
            # data parallel: THIS rank's shard has no class with enough pixels, the global batch may well have.  The rank
            # must still take part in the collective (the others are waiting in it): it contributes padding only, gets
            # the global loss and a zero gradient.  (All ranks empty -> NaN loss: 0 valid rows, like an empty mean.)
            rowidx = torch.empty((0,), device=feats.device, dtype=torch.int32)
            y = torch.empty((0,), device=feats.device, dtype=torch.float32)
            self.last_anchors = ([], [], torch.empty((0, 0), device=feats.device, dtype=torch.int32), 0)
        else:
            n_view, T, req, cls, img = plan
            HW = h * w
            # view-major anchor order of torch.cat(torch.unbind(X_, dim=1)) (loss.py:347): a = v*T + t
            order = [t * n_view + v for v in range(n_view) for t in range(T)]
            req_vm = [req[i] for i in order]
            host = torch.tensor([r + [r[0] * HW] for r in req_vm], dtype=torch.int32)
            dev = host.to(feats.device, non_blocking=False)
            pix = ops.anchor_select(key, hist, dev[:, :3].contiguous(), nc)
            rowidx = (pix + dev[:, 3]).contiguous()
            y = torch.tensor([float(cls[t]) for _ in range(n_view) for t in range(T)], dtype=torch.float32).to(feats.device)
            self.last_anchors = (img, cls, pix.view(n_view, T), n_view)
        if isinstance(feats, LazyUpsampled):
            return _PixelContrastLazyFn.apply(feats.lowres, rowidx, y, float(self.temperature), self.row_gather,
                                              feats.size, cap)
        return _PixelContrastFn.apply(feats, rowidx, y, float(self.temperature), self.row_gather, cap, _row_sink(feats))
