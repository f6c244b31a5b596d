"""DeepLabV3+ / ResNet backbone (BASELINE config 5) on the same HIP kernels.

Drop-in for ``network.modeling.deeplabv3plus_resnet101`` (network/modeling.py:212-220, :44-72): same factory
signature, same 4-tuple forward (network/utils.py:166-194) and the same state_dict keys (``backbone.*`` of the
IntermediateLayerGetter over network/backbone/resnet.py, ``classifier.*`` of DeepLabHeadV3Plus,
network/_deeplab.py:28-64).  The nn.Module tree only owns parameters; ``DeepLabEngine`` runs forward and an
explicit reverse pass through ``dcs_amd.ops``.

Channel concatenations (ASPP 5x256 -> 1280, decoder 48+256 -> 304) are never materialised: the consuming
convolution runs once per source tensor on the matching input-channel slice of its weight and accumulates.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import ops
from .model import FlatBuffers, LOGIT_CS, _cl


def _conv(inp, out, k=1, stride=1, pad=0, dil=1, bias=False):
    return _cl(nn.Conv2d(inp, out, k, stride, pad, dil, bias=bias))


class Bottleneck(nn.Module):
    """Parameter container for network/backbone/resnet.py:74-96."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = _conv(inplanes, planes)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _conv(planes, planes, 3, stride, dilation, dilation)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = _conv(planes, planes * 4)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride, self.dilation = stride, dilation


def _resnet_backbone(layers, output_stride) -> nn.ModuleDict:
    """conv1..layer4 as the ModuleDict the reference's IntermediateLayerGetter exposes (network/utils.py:226-238)."""
    rswd = [False, True, True] if output_stride == 8 else [False, False, True]      # network/modeling.py:46-51
    mods = OrderedDict()
    mods["conv1"] = _cl(nn.Conv2d(3, 64, 7, 2, 3, bias=False))
    mods["bn1"] = nn.BatchNorm2d(64)
    mods["relu"] = nn.ReLU(inplace=True)
    mods["maxpool"] = nn.MaxPool2d(3, 2, 1)
    inplanes, dilation = 64, 1
    for li, (planes, blocks) in enumerate(zip((64, 128, 256, 512), layers)):      # resnet.py:173-195
        stride = 1 if li == 0 else 2
        prev = dilation
        if li > 0 and rswd[li - 1]:
            dilation *= stride
            stride = 1
        ds = None
        if stride != 1 or inplanes != planes * 4:
            ds = nn.Sequential(_conv(inplanes, planes * 4, 1, stride), nn.BatchNorm2d(planes * 4))
        blks = [Bottleneck(inplanes, planes, stride, ds, prev)]
        inplanes = planes * 4
        blks += [Bottleneck(inplanes, planes, dilation=dilation) for _ in range(1, blocks)]
        mods[f"layer{li + 1}"] = nn.Sequential(*blks)
    bb = nn.ModuleDict(mods)
    for m in bb.modules():                                                          # resnet.py:144-149
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
    return bb


class _ASPP(nn.Module):
    def __init__(self, in_channels, rates):
        super().__init__()
        oc = 256
        mods = [nn.Sequential(_conv(in_channels, oc), nn.BatchNorm2d(oc), nn.ReLU(inplace=True))]
        for r in rates:
            mods.append(nn.Sequential(_conv(in_channels, oc, 3, 1, r, r), nn.BatchNorm2d(oc), nn.ReLU(inplace=True)))
        mods.append(nn.Sequential(nn.AdaptiveAvgPool2d(1), _conv(in_channels, oc), nn.BatchNorm2d(oc), nn.ReLU(inplace=True)))
        self.convs = nn.ModuleList(mods)
        self.project = nn.Sequential(_conv(5 * oc, oc), nn.BatchNorm2d(oc), nn.ReLU(inplace=True), nn.Dropout(0.1))
        self.rates = tuple(rates)


class DeepLabHeadV3Plus(nn.Module):
    """Parameter container for network/_deeplab.py:28-64."""

    def __init__(self, in_channels, low_level_channels, num_classes, aspp_dilate):
        super().__init__()
        self.project = nn.Sequential(_conv(low_level_channels, 48), nn.BatchNorm2d(48), nn.ReLU(inplace=True))
        self.aspp = _ASPP(in_channels, aspp_dilate)
        self.classifier = nn.Sequential(_conv(304, 256, 3, 1, 1), nn.BatchNorm2d(256), nn.ReLU(inplace=True),
                                        _conv(256, num_classes, 1, bias=True))
        for m in self.modules():                                                    # _init_weight, :58-64
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)


class _Saved:
    __slots__ = ("tape", "training", "B", "Bm", "shapes", "lazy")


class DeepLabEngine:
    def __init__(self, backbone: nn.ModuleDict, head: DeepLabHeadV3Plus, num_classes: int):
        self.bb, self.head, self.num_classes = backbone, head, num_classes
        self.flat: Optional[FlatBuffers] = None
        self.dropout_noise = None          # test hook: callable(shape_nchw) -> 0/1 float tensor drawn on the host
        self.lazy_ff0 = False              # set by DeepLabV3._get_engine
        self._seed = 0

    def _galloc(self, p):
        """Gradient destination of parameter p: its view of the flat gradient buffer, unless ``p.grad`` already IS
        that view (an earlier backward wrote it and nobody reset it to None): autograd semantics are then to ADD,
        so the engine writes to a scratch tensor that autograd accumulates (gradient accumulation over micro-batches,
        zero_grad(set_to_none=False), and the segmentation head, which is in no ADAM group -- SURVEY.md N1 -- and
        therefore never reset by ``optimizer.zero_grad()``, exactly like in the reference)."""
        v = self.flat.grad_view.get(p) if self.flat is not None else None
        if v is not None and (p.grad is None or p.grad.data_ptr() != v.data_ptr()):
            return v
        return torch.empty_like(p)

    def _bn(self, x, m, training, rows=None, sums=None):
        Cc = x.shape[-1]
        rows = rows if rows is not None else x.numel() // Cc
        if training:
            if sums is None:
                sums = ops.colsum(x.reshape(-1, Cc)[:rows], moments=True)
            bn = ops.bn_finalize(sums, m.weight, m.bias, m.running_mean, m.running_var, rows, True, momentum=m.momentum)
            self._nbt.append(m)
            return bn
        return ops.bn_finalize(None, m.weight, m.bias, m.running_mean, m.running_var, rows, False)

    def _conv(self, x, conv, training, **kw):
        s, p, d = conv.stride[0], conv.padding[0], conv.dilation[0]
        if training:
            return ops.conv_fwd(x, conv.weight, s, p, want_stats=True, dil=d, **kw)
        return ops.conv_fwd(x, conv.weight, s, p, dil=d, **kw), None

    # ------------------------------------------------------------------ forward
    def forward(self, img, training, supcon, need_grad, lazy_seg=False):
        bb, hd = self.bb, self.head
        self._nbt = []
        ops.new_step(training)         # split-weight images of the previous step are stale (optimizer)
        tape = [] if need_grad else None
        parts = list(img) if isinstance(img, (list, tuple)) else [img]
        parts = [t if t.is_floating_point() else t.float() for t in parts]
        Bm = sum(t.shape[0] for t in parts)
        H, W = parts[0].shape[2:]
        dev = parts[0].device
        zero3, one3 = torch.zeros(3, device=dev, dtype=parts[0].dtype), torch.ones(3, device=dev, dtype=parts[0].dtype)
        p0 = ops.normalize_pyramid(parts if len(parts) > 1 else parts[0], zero3, one3, levels=1)[0]   # raw image -> NHWC4
        wst = ops.pack_stem_weight(bb["conv1"].weight)
        y, st = ops.stem_conv(p0, wst, want_stats=True) if training else (ops.stem_conv(p0, wst), None)
        bn = self._bn(y, bb["bn1"], training, sums=st)
        x, pidx = ops.bn_relu_maxpool(y, bn)
        if need_grad:
            tape.append(("stem", p0, y, bn, pidx))
        low = None
        for lname in ("layer1", "layer2", "layer3", "layer4"):
            for blk in bb[lname]:
                x = self._block_fwd(x, blk, training, tape)
            if lname == "layer1":
                low = x
                if need_grad:
                    tape.append(("low",))
        out = x                                                  # [Bm, H/16, W/16, 2048]
        B = Bm // 2 if supcon else Bm
        f, lo = out[:B], low[:B]
        asp = hd.aspp
        hf, wf = f.shape[1:3]
        # ---- ASPP (network/_deeplab.py:140-169) ----
        branches = []
        for i in range(4):
            conv, bnm = asp.convs[i][0], asp.convs[i][1]
            yb, stb = self._conv(f, conv, training)
            bnb = self._bn(yb, bnm, training, sums=stb)
            zb = ops.bn_act(yb, bnb, relu=True)
            branches.append((conv, bnm, yb, bnb, zb))
        pconv, pbn = asp.convs[4][1], asp.convs[4][2]
        pooled = ops.colsum(f.reshape(B * hf * wf, f.shape[-1]), B=B, scale=1.0 / (hf * wf))[:, 0, :].contiguous()
        if training and B == 1:              # same error as nn.BatchNorm2d on a [1,256,1,1] input (torch/nn/functional.py)
            raise ValueError(f"Expected more than 1 value per channel when training, got input size {(B, 256, 1, 1)}")
        yp = ops.linear(pooled, pconv.weight.reshape(pconv.weight.shape[0], -1))
        bnp = self._bn(yp, pbn, training, rows=B)
        zp = ops.bn_act(yp, bnp, relu=True)                                    # [B,256]
        zpb = ops.upsample_add(zp.reshape(B, 1, 1, -1), [], hf, wf)             # bilinear from 1x1 = broadcast
        prj, prj_bn = asp.project[0], asp.project[1]
        srcs = [b[4] for b in branches] + [zpb]
        yj = None
        for i, z in enumerate(srcs):                                           # virtual concat: slice i*256 of the 1280 inputs
            if yj is None:
                yj = ops.conv_fwd(z, prj.weight, 1, 0, koff=256 * i)
            else:
                ops.conv_fwd(z, prj.weight, 1, 0, koff=256 * i, out=yj)
        bnj = self._bn(yj, prj_bn, training)
        zj = ops.bn_act(yj, bnj, relu=True)
        dmask = None
        if training:
            noise = None
            if self.dropout_noise is not None:
                noise = self.dropout_noise((B, zj.shape[3], hf, wf)).to(zj.device, zj.dtype).permute(0, 2, 3, 1).contiguous()
            self._seed += 1
            dj, dmask = ops.dropout(zj, 0.1, noise, seed=self._seed)
        else:
            dj = zj
        # ---- decoder (network/_deeplab.py:47-56) ----
        lconv, lbn = hd.project[0], hd.project[1]
        yl, stl = self._conv(lo, lconv, training)
        bnl = self._bn(yl, lbn, training, sums=stl)
        zl = ops.bn_act(yl, bnl, relu=True)                                    # [B,h,w,48]
        h, w = zl.shape[1:3]
        up = ops.upsample_add(dj, [], h, w)                                    # [B,h,w,256]
        c0, cbn, c3 = hd.classifier[0], hd.classifier[1], hd.classifier[3]
        yc = ops.conv_fwd(zl, c0.weight, 1, 1, koff=0)
        ops.conv_fwd(up, c0.weight, 1, 1, koff=48, out=yc)
        bnc = self._bn(yc, cbn, training)
        zc = ops.bn_act(yc, bnc, relu=True)
        before = ops.conv_fwd(zc, c3.weight, 1, 0, bias=c3.bias, dst_cs=LOGIT_CS)
        seg = before.new_zeros(1) if lazy_seg else ops.upsample_to_nchw(before, self.num_classes, H, W)
        if self.lazy_ff0:       # hand out the low-resolution map, rows are interpolated on demand
            ff0 = out[:B] if B != Bm else out.view(out.shape)
        else:
            ff0 = ops.upsample_add(f.contiguous(), [], h, w)                   # network/utils.py:190
        if self._nbt and not (self.flat is not None and self.flat.bump_counters(self._nbt)):
            for m in self._nbt:
                m.num_batches_tracked += 1
        self._nbt = []
        saved = None
        if need_grad:
            saved = _Saved()
            tape.append(("head", f, lo, branches, (pconv, pbn, pooled, yp, bnp, zp), (prj, prj_bn, srcs, yj, bnj, zj, dmask),
                         (lconv, lbn, yl, bnl, zl), (c0, cbn, yc, bnc, zc, up, c3), (hf, wf, h, w, H, W)))
            saved.tape, saved.training, saved.B, saved.Bm = tape, training, B, Bm
            saved.shapes = (out.shape, low.shape)
            saved.lazy = self.lazy_ff0
        return seg, before, out, ff0, saved

    def _block_fwd(self, x, blk: Bottleneck, training, tape):
        y1, s1 = self._conv(x, blk.conv1, training)
        bn1 = self._bn(y1, blk.bn1, training, sums=s1)
        z1 = ops.bn_act(y1, bn1, relu=True)
        y2, s2 = self._conv(z1, blk.conv2, training)
        bn2 = self._bn(y2, blk.bn2, training, sums=s2)
        z2 = ops.bn_act(y2, bn2, relu=True)
        y3, s3 = self._conv(z2, blk.conv3, training)
        bn3 = self._bn(y3, blk.bn3, training, sums=s3)
        yd = bnd = None
        if blk.downsample is not None:
            yd, sd = self._conv(x, blk.downsample[0], training)
            bnd = self._bn(yd, blk.downsample[1], training, sums=sd)
            out = ops.bn_act(y3, bn3, r=yd, bn2=bnd, relu=True)
        else:
            out = ops.bn_act(y3, bn3, r=x, relu=True)
        if tape is not None:
            tape.append(("block", blk, x, y1, bn1, z1, y2, bn2, z2, y3, bn3, yd, bnd, out))
        return out

    # ------------------------------------------------------------------ backward
    def backward(self, saved, g_seg, g_before, g_ff, g_ff0):
        grads: Dict[nn.Parameter, torch.Tensor] = {}
        packed = {}
        training = saved.training
        B, Bm = saved.B, saved.Bm

        def wgrad(conv, x, dy, koff=None):      # every weight (slice) is used exactly once per forward: plain write
            w = conv.weight
            if w not in grads:
                grads[w] = self._galloc(w)
            ops.conv_wgrad(x, dy, grads[w], conv.stride[0], conv.padding[0], False, dil=conv.dilation[0], koff=koff)

        def dgrad(conv, dy, in_hw, koff=0, kw=None, out=None, accumulate=False):
            key = (conv.weight, koff, kw)
            if key not in packed:
                packed[key] = ops.pack_dgrad_weight(conv.weight, koff, kw)
            return ops.conv_dgrad(dy, packed[key], in_hw, conv.stride[0], conv.padding[0], out=out,
                                  accumulate=accumulate, dil=conv.dilation[0])

        def bn_bwd(m, g, y, bn, **kw):
            grads[m.weight] = self._galloc(m.weight)
            grads[m.bias] = self._galloc(m.bias)
            return ops.bn_bwd(g, y, bn, m.weight, dgamma=grads[m.weight], dbeta=grads[m.bias], training=training, **kw)

        tape = saved.tape
        _, f, lo, branches, pool, proj, lowp, cls, dims = tape[-1]
        hf, wf, h, w, H, W = dims
        out_shape, low_shape = saved.shapes
        dev = f.device
        g_out = g_ff if g_ff is not None else torch.zeros(out_shape, device=dev, dtype=f.dtype)
        if g_ff0 is not None:
            if saved.lazy:
                ops.axpy(g_out[:B], g_ff0.contiguous(), 1.0)
            else:
                ops.upsample_bwd(g_ff0.contiguous(), hf, wf, out=g_out[:B], accumulate=True)
        g_low = None
        gb = None
        if g_seg is not None:
            gb = ops.upsample_to_nchw_bwd(g_seg, h, w, LOGIT_CS)
        if g_before is not None:
            if gb is None:
                gb = g_before.contiguous()
            else:
                ops.axpy(gb, g_before.contiguous(), 1.0)
        if gb is not None:
            c0, cbn, yc, bnc, zc, up, c3 = cls
            wgrad(c3, zc, gb)
            grads[c3.bias] = self._galloc(c3.bias)
            grads[c3.bias].copy_(ops.colsum(gb.reshape(-1, LOGIT_CS))[0, 0, :self.num_classes])
            wpad = torch.zeros((zc.shape[-1], 1, 1, LOGIT_CS), device=dev, dtype=f.dtype)
            wpad[..., :self.num_classes] = ops.pack_dgrad_weight(c3.weight)
            g_zc = ops.conv_dgrad(gb, wpad, (h, w), 1, 0)
            dyc, _ = bn_bwd(cbn, g_zc, yc, bnc, relu=True)
            lconv, lbn, yl, bnl, zl = lowp
            wgrad(c0, zl, dyc, koff=0)
            wgrad(c0, up, dyc, koff=48)
            g_zl = dgrad(c0, dyc, (h, w), 0, 48)
            g_up = dgrad(c0, dyc, (h, w), 48, 256)
            dyl, _ = bn_bwd(lbn, g_zl, yl, bnl, relu=True)
            wgrad(lconv, lo, dyl)
            g_lo = dgrad(lconv, dyl, lo.shape[1:3])
            g_low = torch.zeros(low_shape, device=dev, dtype=f.dtype) if Bm != B else None
            if g_low is None:
                g_low = g_lo
            else:
                g_low[:B].copy_(g_lo)
            # ASPP
            prj, prj_bn, srcs, yj, bnj, zj, dmask = proj
            g_dj = ops.upsample_bwd(g_up, hf, wf)
            g_zj = ops.dropout_bwd(g_dj, dmask, 0.1) if dmask is not None else g_dj
            dyj, _ = bn_bwd(prj_bn, g_zj, yj, bnj, relu=True)
            g_src = []
            for i, z in enumerate(srcs):
                wgrad(prj, z, dyj, koff=256 * i)
                g_src.append(dgrad(prj, dyj, (hf, wf), 256 * i, 256))
            pconv, pbn, pooled, yp, bnp, zp = pool
            g_zp = ops.upsample_bwd(g_src[4], 1, 1).reshape(B, -1)                       # adjoint of the broadcast
            dyp, _ = bn_bwd(pbn, g_zp, yp, bnp, relu=True)
            wp2 = pconv.weight.reshape(pconv.weight.shape[0], -1)
            gw = torch.empty_like(wp2)
            ops.linear_wgrad(pooled, dyp, gw)
            grads[pconv.weight] = self._galloc(pconv.weight)
            grads[pconv.weight].copy_(gw.reshape(pconv.weight.shape))
            g_pool = ops.linear(dyp, ops.transpose(wp2.contiguous()))                   # [B,2048]
            g_f = None
            for i, (conv, bnm, yb, bnb, zb) in enumerate(branches):
                dyb, _ = bn_bwd(bnm, g_src[i], yb, bnb, relu=True)
                wgrad(conv, f, dyb)
                g_f = dgrad(conv, dyb, (hf, wf)) if g_f is None else dgrad(conv, dyb, (hf, wf), out=g_f, accumulate=True)
            ops.add_rowvec_bcast(g_f, g_pool, 1.0 / (hf * wf))
            ops.axpy(g_out[:B], g_f, 1.0)
        # ---- backbone ----
        g_cur = g_out
        pos = len(tape) - 2
        while pos >= 0:
            item = tape[pos]
            pos -= 1
            if item[0] == "low":
                if g_low is not None:
                    ops.axpy(g_cur, g_low, 1.0)                                          # low-level feature branch
                continue
            if item[0] == "block":
                _, blk, x, y1, bn1, z1, y2, bn2, z2, y3, bn3, yd, bnd, outt = item
                dy3, gm = bn_bwd(blk.bn3, g_cur, y3, bn3, masksrc=outt, want_gm=True)
                wgrad(blk.conv3, z2, dy3)
                g_z2 = dgrad(blk.conv3, dy3, z2.shape[1:3])
                dy2, _ = bn_bwd(blk.bn2, g_z2, y2, bn2, relu=True)
                wgrad(blk.conv2, z1, dy2)
                g_z1 = dgrad(blk.conv2, dy2, z1.shape[1:3])
                dy1, _ = bn_bwd(blk.bn1, g_z1, y1, bn1, relu=True)
                wgrad(blk.conv1, x, dy1)
                if blk.downsample is not None:
                    dyd, _ = bn_bwd(blk.downsample[1], gm, yd, bnd)
                    wgrad(blk.downsample[0], x, dyd)
                    g_in = dgrad(blk.downsample[0], dyd, x.shape[1:3])
                else:
                    g_in = gm
                dgrad(blk.conv1, dy1, x.shape[1:3], out=g_in, accumulate=True)
                g_cur = g_in
            else:
                _, p0, y, bn, pidx = item
                m1 = self.bb["bn1"]
                grads[m1.weight], grads[m1.bias] = self._galloc(m1.weight), self._galloc(m1.bias)
                dy = ops.bn_pool_bwd(g_cur, pidx, y, bn, m1.weight, dgamma=grads[m1.weight], dbeta=grads[m1.bias],
                                     acc_param=False, training=training)
                dwst = torch.empty((64, 7, 8, 4), device=dy.device, dtype=dy.dtype)
                ops.stem_wgrad(p0, dy, dwst, False)
                w1 = self.bb["conv1"].weight
                grads[w1] = ops.unpack_stem_weight(dwst, w1, out=self._galloc(w1))
        return grads


class _DeepLabFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine: DeepLabEngine, img, training, supcon, grad_enabled, lazy_seg, *params):
        need_grad = grad_enabled and any(p.requires_grad for p in params)
        ctx.set_materialize_grads(False)
        seg, before, ff, ff0, saved = engine.forward(img, training, supcon, need_grad, lazy_seg)
        ctx.engine, ctx.saved, ctx.params, ctx.lazy_seg = engine, saved, params, lazy_seg
        if lazy_seg:
            ctx.mark_non_differentiable(seg)
        return seg, before, ff, ff0

    @staticmethod
    def backward(ctx, g_seg, g_before, g_ff, g_ff0):
        if ctx.saved is None:
            raise RuntimeError("DeepLab backward without a recorded forward")
        g_seg = g_seg.contiguous() if (g_seg is not None and not ctx.lazy_seg) else None
        g_ff = g_ff.contiguous() if g_ff is not None else None
        grads = ctx.engine.backward(ctx.saved, g_seg, g_before, g_ff, g_ff0)
        flat = ctx.engine.flat
        ctx.saved = None
        res = []
        for p in ctx.params:
            gp = grads.get(p)
            if gp is not None and flat is not None and flat.grad_view.get(p) is gp:
                if p.grad is None:
                    p.grad = gp
                elif p.grad.data_ptr() != gp.data_ptr():
                    p.grad.add_(gp)
                res.append(None)
            else:
                res.append(gp)
        return (None, None, None, None, None, None) + tuple(res)


class DeepLabV3(nn.Module):
    """network/utils.py:159-194 (_SimpleSegmentationModel) with the HIP engine."""

    def __init__(self, backbone: nn.ModuleDict, classifier: DeepLabHeadV3Plus, num_classes: int,
                 lazy_fine_feat0: bool = False):
        super().__init__()
        self.backbone = backbone
        self.classifier = classifier
        self.num_classes = num_classes
        # False: the reference's 4-tuple (fine_feat0 = the upsampled [B,2048,h,w] tensor).  True: fine_feat0 is a
        # losses.LazyUpsampled handle that PixelContrastLoss samples row by row (same values, no 4.3 GB tensor).
        self.lazy_fine_feat0 = lazy_fine_feat0
        self._engine = None

    def _get_engine(self):
        if self._engine is None:
            object.__setattr__(self, "_engine", DeepLabEngine(self.backbone, self.classifier, self.num_classes))
        self._engine.lazy_ff0 = bool(self.lazy_fine_feat0)
        return self._engine

    def flatten_parameters(self):
        flat = FlatBuffers([list(self.parameters())])       # utils/init_trainer.py:163-168: one ADAM group
        flat.adopt_counters(self.modules())
        self._get_engine().flat = flat
        return flat

    def forward(self, left_img, return_supcon_feature=False):
        for t in (left_img if isinstance(left_img, (list, tuple)) else [left_img]):
            ops.require_device(t, "left_img")
        lazy = bool(self.training and torch.is_grad_enabled() and getattr(self, "lazy_pred_segmap", True))
        seg, before, ff, ff0 = _DeepLabFn.apply(self._get_engine(), left_img, self.training, bool(return_supcon_feature),
                                                torch.is_grad_enabled(), lazy, *list(self.parameters()))
        before_nchw = before[..., :self.num_classes].permute(0, 3, 1, 2)
        if lazy:
            from .losses import LazyLogits
            first = left_img[0] if isinstance(left_img, (list, tuple)) else left_img
            seg = LazyLogits(before, self.num_classes, first.shape[-2:])
        if self.lazy_fine_feat0:
            from .losses import LazyUpsampled
            return seg, before_nchw, ff.permute(0, 3, 1, 2), LazyUpsampled(ff0.permute(0, 3, 1, 2), before_nchw.shape[-2:])
        return seg, before_nchw, ff.permute(0, 3, 1, 2), ff0.permute(0, 3, 1, 2)


def _segm_resnet(opts, layers, num_classes, output_stride):
    aspp_dilate = [12, 24, 36] if output_stride == 8 else [6, 12, 18]
    backbone = _resnet_backbone(layers, output_stride)
    m = DeepLabV3(backbone, DeepLabHeadV3Plus(2048, 256, num_classes, aspp_dilate), num_classes,
                  lazy_fine_feat0=bool(getattr(opts, "lazy_fine_feat0", False)))
    m.lazy_pred_segmap = bool(getattr(opts, "lazy_pred_segmap", True))     # see losses.LazyLogits
    return m


def deeplabv3plus_resnet101(opts, num_classes=21, output_stride=8, pretrained_backbone=True):
    """network/modeling.py:212-220.  ``pretrained_backbone`` is accepted for signature parity; ImageNet weights are a
    network fetch in the reference (resnet.py:215) and must be loaded from a local state_dict here."""
    return _segm_resnet(opts, (3, 4, 23, 3), num_classes, output_stride)


def deeplabv3plus_resnet50(opts, num_classes=21, output_stride=8, pretrained_backbone=True):
    return _segm_resnet(opts, (3, 4, 6, 3), num_classes, output_stride)
