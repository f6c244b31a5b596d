"""Pyramid SwiftNet / ResNet-18 with the reference's module tree and state_dict
(network/weathernet.py:14-104, network/backbone/resnet_pyramid.py:55-379,
network/utils.py:35-102), executed by hand-written HIP kernels.

The nn.Module tree below only OWNS parameters and buffers (same names, shapes
and registration order as the reference, so checkpoints load either way);
it is never called.  ``SwiftNetEngine`` runs the forward and an explicit
reverse pass through ``dcs_amd.ops`` and is exposed to autograd as ONE
``torch.autograd.Function`` whose outputs are the reference's model outputs.
"""
from __future__ import annotations

import os
from itertools import chain
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import ops

NUM_FEATURES = 128
LAYERS = (("layer1", 64, 1), ("layer2", 128, 2), ("layer3", 256, 2), ("layer4", 512, 2))
LOGIT_CS = 20      # logits are stored NHWC with a channel stride of 20 (19 classes + one zero pad)


def _cl(conv: nn.Conv2d) -> nn.Conv2d:
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    return conv


def convkxk(inp, out, stride=1, k=3):
    return _cl(nn.Conv2d(inp, out, kernel_size=k, stride=stride, padding=k // 2, bias=False))


class BasicBlock(nn.Module):
    """Parameter container for resnet_pyramid.py:55-69."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = convkxk(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = convkxk(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride


class _BNReluConv(nn.Sequential):
    """Parameter container for network/utils.py:35-49."""

    def __init__(self, num_maps_in, num_maps_out, k=3, bias=False):
        super().__init__()
        self.add_module("norm", nn.BatchNorm2d(num_maps_in, momentum=0.1))
        self.add_module("relu", nn.ReLU(inplace=True))
        self.add_module("conv", _cl(nn.Conv2d(num_maps_in, num_maps_out, kernel_size=k, padding=k // 2, bias=bias)))


class _UpsampleBlend(nn.Module):
    def __init__(self, num_features, k=3):
        super().__init__()
        self.blend_conv = _BNReluConv(num_features, num_features, k=k)


class ResNetPyramid(nn.Module):
    """Parameter container for resnet_pyramid.py:130-254 (registration order kept)."""

    def __init__(self, layers=(2, 2, 2, 2), num_features=NUM_FEATURES, mean=(73.15, 82.90, 72.3),
                 std=(47.67, 48.49, 47.73)):
        super().__init__()
        self.inplanes = 64
        self.conv1 = _cl(nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False))
        self.register_buffer("img_mean", torch.tensor(mean).view(1, -1, 1, 1))
        self.register_buffer("img_std", torch.tensor(std).view(1, -1, 1, 1))
        self.num_features = num_features
        self.bn1_0 = nn.BatchNorm2d(64)
        self.bn1_1 = nn.BatchNorm2d(64)
        self.bn1_2 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0])
        self.upsample_bottlenecks1 = convkxk(self.inplanes, num_features, k=1)
        self.layer2 = self._make_layer(128, layers[1], stride=2)
        self.upsample_bottlenecks2 = convkxk(self.inplanes, num_features, k=1)
        self.layer3 = self._make_layer(256, layers[2], stride=2)
        self.upsample_bottlenecks3 = convkxk(self.inplanes, num_features, k=1)
        self.layer4 = self._make_layer(512, layers[3], stride=2)
        self.upsample_bottlenecks4 = convkxk(self.inplanes, num_features, k=1)
        self.fine_tune = [self.conv1, self.maxpool, self.layer1, self.layer2, self.layer3, self.layer4,
                          self.bn1_0, self.bn1_1, self.bn1_2]
        for i in range(1, 6):
            setattr(self, f"upsample_blends{i}", _UpsampleBlend(num_features, k=3))
        self.random_init = [self.upsample_bottlenecks1, self.upsample_bottlenecks2, self.upsample_bottlenecks3,
                            self.upsample_bottlenecks4] + [getattr(self, f"upsample_blends{i}") for i in range(1, 6)]
        self.features = num_features
        for m in self.modules():                      # resnet_pyramid.py:249-254
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(
                _cl(nn.Conv2d(self.inplanes, planes, kernel_size=1, stride=stride, bias=False)),
                nn.BatchNorm2d(planes))
        layers = [BasicBlock(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        for _ in range(1, blocks):
            layers.append(BasicBlock(self.inplanes, planes))
        return nn.Sequential(*layers)

    def random_init_params(self):
        return chain(*[f.parameters() for f in self.random_init])

    def fine_tune_params(self):
        return chain(*[f.parameters() for f in self.fine_tune])

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        """ImageNet ResNet checkpoints carry one ``bn1.*``; fan it out to the three per-level stem
        BNs like resnet_pyramid.py:381-393."""
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)
        if any(k.startswith(prefix + "bn1.") for k in state_dict):
            for bn in (self.bn1_0, self.bn1_1, self.bn1_2):
                bn._load_from_state_dict(state_dict, prefix + "bn1.", local_metadata, False, [], [], error_msgs)


# --------------------------------------------------------------------------- #
class FlatBuffers:
    """Parameters of each optimizer group re-homed as views of ONE flat fp32 buffer (element order = each
    parameter's own storage order, i.e. KRSC for conv weights), with a same-shaped flat gradient buffer.
    The backward pass writes gradients straight into the gradient views; Adam is then one kernel launch per
    group and the data-parallel all-reduce runs on the flat gradient buffer without a copy."""

    def __init__(self, groups: List[List[nn.Parameter]]):
        self.groups = []
        self.grad_view: Dict[nn.Parameter, torch.Tensor] = {}
        for ps in groups:
            ps = list(ps)
            if not ps:
                continue
            n = sum(p.numel() for p in ps)
            flat_p = torch.empty(n, device=ps[0].device, dtype=ps[0].dtype)
            flat_g = torch.zeros(n, device=ps[0].device, dtype=ps[0].dtype)
            off = 0
            for p in ps:
                v = torch.as_strided(flat_p, p.size(), p.stride(), off)
                v.copy_(p.data)
                p.data = v
                self.grad_view[p] = torch.as_strided(flat_g, p.size(), p.stride(), off)
                off += p.numel()
            self.groups.append(dict(params=ps, flat_p=flat_p, flat_g=flat_g))

    def adopt_counters(self, modules):
        """Re-home every BatchNorm ``num_batches_tracked`` as a 0-dim view of ONE int64 buffer, so a step's bookkeeping
        is a single add instead of one tiny launch per BatchNorm call (~114 per C3 step)."""
        bns = [m for m in modules if isinstance(m, nn.BatchNorm2d) and m.num_batches_tracked is not None]
        if not bns:
            return
        self.nbt_flat = torch.stack([m.num_batches_tracked.reshape(()) for m in bns]).contiguous()
        self.nbt_index = {}
        for i, m in enumerate(bns):
            m._buffers["num_batches_tracked"] = self.nbt_flat[i]
            self.nbt_index[m] = i
        self._nbt_delta = {}

    def bump_counters(self, mods) -> bool:
        """num_batches_tracked += (number of occurrences) for the given modules; False if a module is not adopted
        (e.g. the model was moved / cast after flattening: its buffers are new tensors)."""
        idx = getattr(self, "nbt_index", None)
        if idx is None or any(m not in idx or m.num_batches_tracked.data_ptr() !=
                              self.nbt_flat[idx[m]].data_ptr() for m in set(mods)):
            return False
        key = tuple(idx[m] for m in mods)
        d = self._nbt_delta.get(key)
        if d is None:
            d = torch.zeros_like(self.nbt_flat)
            for i in key:
                d[i] += 1
            self._nbt_delta[key] = d
        self.nbt_flat.add_(d)
        return True

    def aliased(self, ps) -> bool:
        return all(p.grad is not None and p in self.grad_view and p.grad.data_ptr() == self.grad_view[p].data_ptr()
                   for p in ps)


class _Saved:
    __slots__ = ("tape", "shapes", "training", "B", "Bm", "ff_shape")


class SwiftNetEngine:
    """Forward + explicit backward of the pyramid SwiftNet on HIP kernels."""

    def __init__(self, fe: ResNetPyramid, seg: Optional[_BNReluConv], num_classes: int):
        self.fe, self.seg, self.num_classes = fe, seg, num_classes
        self._mean = self._std = None
        self.flat: Optional[FlatBuffers] = None

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

    # ---- helpers ---------------------------------------------------------
    def _bn(self, x, m: nn.BatchNorm2d, training, rows=None, sums=None):
        """BatchNorm record [scale, shift, mean, invstd] of x; ``sums`` = statistics already reduced by the
        producing convolution's epilogue (otherwise a reduction pass over x)."""
        Cc = x.shape[-1]
        rows = rows if rows is not None else x.numel() // Cc
        if training:
            if sums is None:
                sums = ops.colsum(x.reshape(-1, Cc)[:rows], moments=True)
            bn = ops.bn_finalize(sums, m.weight, m.bias, m.running_mean, m.running_var, rows, True,
                                 momentum=m.momentum)
            self._nbt.append(m)
            return bn
        return ops.bn_finalize(None, m.weight, m.bias, m.running_mean, m.running_var, rows, False)

    def _flush_nbt(self):
        if self._nbt and not (self.flat is not None and self.flat.bump_counters(self._nbt)):
            for m in self._nbt:
                m.num_batches_tracked += 1      # scalar bookkeeping (not on the data path)
        self._nbt = []

    # ---- forward ---------------------------------------------------------
    def forward(self, img, training: bool, supcon: bool, need_grad: bool, lazy_seg: bool = False):
        """lazy_seg: do not materialise the full-resolution logits (the caller wraps ``before`` in losses.LazyLogits)."""
        fe = self.fe
        self._mean = fe.img_mean.reshape(3).contiguous()
        self._std = fe.img_std.reshape(3).contiguous()
        self._nbt = []
        ops.new_step(training)         # split-weight images of the previous step are stale (optimizer)
        tape: List[tuple] = [] if need_grad else None
        parts = list(img) if isinstance(img, (list, tuple)) else [img]
        parts = [t if t.is_floating_point() else t.float() for t in parts]
        Bm = sum(t.shape[0] for t in parts)
        H, W = parts[0].shape[2:]
        pyr = ops.normalize_pyramid(parts if len(parts) > 1 else parts[0], self._mean, self._std)
        wst = ops.pack_stem_weight(fe.conv1.weight)
        # The levels run in LOCKSTEP (stem, then block by block): inside ops.level_batch the convolutions the levels have in
        # common go out as one launch per layer.  Each level keeps its own tape (same item order as a level-by-level run).
        nl = len(pyr)
        ltapes: List[list] = [[] for _ in range(nl)] if need_grad else None
        skips: List[List[tuple]] = [[] for _ in range(6)]
        xs: List[torch.Tensor] = [None] * nl
        with ops.level_batch() as lb:
            for idx, p in enumerate(pyr):
                lb.level(idx)
                bnm = getattr(fe, f"bn1_{idx}")
                y, st = ops.stem_conv(p, wst, want_stats=True) if training else (ops.stem_conv(p, wst), None)
                bn = self._bn(y, bnm, training, sums=st)
                xs[idx], pidx = ops.bn_relu_maxpool(y, bn)
                if need_grad:
                    ltapes[idx].append(("stem", idx, p, y, bn, pidx, bnm))
            lb.flush()
            for li, (lname, planes, stride) in enumerate(LAYERS):
                for blk in getattr(fe, lname):
                    for idx in range(nl):
                        lb.level(idx)
                        xs[idx] = self._block_fwd(xs[idx], blk, training, ltapes[idx] if need_grad else None)
                    lb.flush()
                bott = getattr(fe, f"upsample_bottlenecks{li + 1}")
                for idx in range(nl):
                    lb.level(idx)
                    s = ops.conv_fwd(xs[idx], bott.weight, 1, 0)
                    if need_grad:
                        ltapes[idx].append(("skip", idx + li, xs[idx], bott))
                    skips[idx + li].append((idx, s))
                lb.flush()
        skips = [[t for _, t in sorted(sk, key=lambda e: e[0])] for sk in skips]       # level order, as level-by-level
        if need_grad:
            tape.append(("levels", ltapes))
        skips = skips[::-1]
        x = skips[0][0]
        B = Bm // 2 if supcon else Bm
        st_head = None
        for i in range(1, 6):
            blend = getattr(fe, f"upsample_blends{i}").blend_conv
            sk = skips[i]
            OH, OW = sk[0].shape[1:3]
            if training:                   # the BatchNorm statistics of t ride the kernel that writes it
                t, st = ops.upsample_add(x, sk, OH, OW, want_stats=True)
            else:
                t, st = ops.upsample_add(x, sk, OH, OW), None
            bn = self._bn(t, blend.norm, training, sums=st)
            z, pro = self._activated(t, bn)
            if i == 5 and self.seg is not None and training:
                # the segmentation head's BatchNorm sees the first crop only (weathernet.py:78-82): its statistics are a
                # prefix of the per-tile sums of the convolution that produces fine_feat
                xn, st_head = ops.conv_fwd(z, blend.conv.weight, 1, 1, pro=pro, want_stats=True, stats_images=B)
            else:
                xn = ops.conv_fwd(z, blend.conv.weight, 1, 1, pro=pro)
            if need_grad:
                tape.append(("blend", i, x.shape[1:3], t, bn, z, blend))
            x = xn
        fine_feat = x
        seg = before = None
        if self.seg is not None:
            h, w = fine_feat.shape[1:3]
            ff0 = fine_feat[:B]
            bnh = self._bn(ff0, self.seg.norm, training, rows=B * h * w, sums=st_head)
            zh, pro = self._activated(ff0, bnh)
            before = ops.conv_fwd(zh, self.seg.conv.weight, 1, 0, bias=self.seg.conv.bias, dst_cs=LOGIT_CS, pro=pro)
            seg = None if lazy_seg else ops.upsample_to_nchw(before, self.num_classes, H, W)
            if need_grad:
                tape.append(("head", ff0, bnh, zh, (H, W)))
        self._flush_nbt()
        saved = None
        if need_grad:
            saved = _Saved()
            saved.tape, saved.training, saved.B, saved.Bm = tape, training, B, Bm
            saved.ff_shape = tuple(fine_feat.shape)
        return seg, before, fine_feat, saved

    @staticmethod
    def _activated(y, bn):
        """relu(BatchNorm(y)) as the source operand of the next convolution: (y, bn) when the convolution kernels apply
        it as their prologue (the activated tensor is then never materialised -- the backward's weight gradient takes
        the same pair), else (the materialised tensor, None)."""
        if y.is_contiguous() and ops.pro_ok(y.shape[-1]):
            return y, bn
        return ops.bn_act(y, bn, relu=True), None

    def _block_fwd(self, x, blk: BasicBlock, training, tape):
        s = blk.stride

        def conv(inp, w, stride, pad, pro=None):   # conv + (in training) the BN batch statistics from its epilogue
            if training:
                return ops.conv_fwd(inp, w, stride, pad, want_stats=True, pro=pro)
            return ops.conv_fwd(inp, w, stride, pad, pro=pro), None

        y1, st1 = conv(x, blk.conv1.weight, s, 1)
        bn1 = self._bn(y1, blk.bn1, training, sums=st1)
        z1, pro1 = self._activated(y1, bn1)
        y2, st2 = conv(z1, blk.conv2.weight, 1, 1, pro1)
        bn2 = self._bn(y2, blk.bn2, training, sums=st2)
        yd = bnd = None
        if blk.downsample is not None:
            yd, std_ = conv(x, blk.downsample[0].weight, s, 0)
            bnd = self._bn(yd, blk.downsample[1], training, sums=std_)
            out = ops.bn_act(y2, bn2, r=yd, bn2=bnd, relu=True)
        else:
            out = ops.bn_act(y2, bn2, r=x, relu=True)
        if tape is not None:
            tape.append(("block", blk, x, y1, bn1, z1, y2, bn2, yd, bnd, out))
        return out

    # ---- backward --------------------------------------------------------
    def backward(self, saved: _Saved, g_seg, g_before, g_ff) -> Dict[nn.Parameter, torch.Tensor]:
        """Returns {parameter: gradient}.  g_seg NCHW [B,C,H,W]; g_before NHWC [B,h,w,LOGIT_CS] or None;
        g_ff NHWC [Bm,h,w,128] or None (external gradient of fine_feat; consumed/modified)."""
        grads: Dict[nn.Parameter, torch.Tensor] = {}
        packed: Dict[nn.Parameter, torch.Tensor] = {}
        training = saved.training

        def wgrad(conv, x, dy, stride, pad, pro=None):
            w = conv.weight
            acc = w in grads
            if not acc:
                grads[w] = self._galloc(w)
            ops.conv_wgrad(x, dy, grads[w], stride, pad, acc, pro=pro)

        def pro_of(z, y, bn):
            """The forward kept y itself where the consumer applied BatchNorm + ReLU as its prologue (_activated)."""
            return bn if z is y else None

        def wp(conv):
            w = conv.weight
            if w not in packed:
                packed[w] = ops.pack_dgrad_weight(w)
            return packed[w]

        def bn_bwd(m: nn.BatchNorm2d, g, y, bn, **kw):
            acc = m.weight in grads
            if not acc:
                grads[m.weight] = self._galloc(m.weight)
                grads[m.bias] = self._galloc(m.bias)
            return ops.bn_bwd(g, y, bn, m.weight, dgamma=grads[m.weight], dbeta=grads[m.bias], acc_param=acc,
                              training=training, **kw)

        tape = saved.tape
        pos = len(tape) - 1
        g_x = None                     # gradient of the decoder tensor flowing down
        g_skip: Dict[int, torch.Tensor] = {}
        # ---- head ----
        if tape[pos][0] == "head":
            _, ff0, bnh, zh, (H, W) = tape[pos]
            pos -= 1
            B, h, w, _ = ff0.shape
            gb = None
            if g_seg is not None:
                gb = ops.upsample_to_nchw_bwd(g_seg, h, w, LOGIT_CS)
            if g_before is not None:
                if gb is None:
                    gb = g_before.contiguous()
                else:
                    ops.axpy(gb, g_before.contiguous(), 1.0)
            if gb is not None:
                seg = self.seg
                wgrad(seg.conv, zh, gb, 1, 0, pro_of(zh, ff0, bnh))
                bsum = ops.colsum(gb.reshape(-1, LOGIT_CS))
                grads[seg.conv.bias] = self._galloc(seg.conv.bias)
                grads[seg.conv.bias].copy_(bsum[0, 0, :self.num_classes])
                wpad = torch.zeros((NUM_FEATURES, 1, 1, LOGIT_CS), device=gb.device, dtype=gb.dtype)
                wpad[..., :self.num_classes] = wp(seg.conv)
                # the data gradient's epilogue also reduces the sums of the BatchNorm backward that consumes it
                ff0c = ff0 if ff0.is_contiguous() else None
                if ff0c is not None:
                    g_zh, hs = ops.conv_dgrad(gb, wpad, (h, w), 1, 0, bnb=(ff0c, None, bnh, True))
                else:
                    g_zh, hs = ops.conv_dgrad(gb, wpad, (h, w), 1, 0), None
                if g_ff is None and saved.Bm == B:
                    g_ff, _ = bn_bwd(seg.norm, g_zh, ff0, bnh, relu=True, sums=hs)
                else:
                    if g_ff is None:
                        g_ff = torch.zeros((saved.Bm, h, w, NUM_FEATURES), device=gb.device, dtype=gb.dtype)
                    bn_bwd(seg.norm, g_zh, ff0, bnh, relu=True, dy_out=g_ff[:B], acc_dy=True, sums=hs)
        if g_ff is None:
            raise RuntimeError("backward called without any gradient")
        g_x = g_ff
        # ---- decoder ----
        while pos >= 0 and tape[pos][0] == "blend":
            _, i, in_hw, t, bn, z, blend = tape[pos]
            pos -= 1
            ops.tag_max(g_x)        # no BatchNorm backward produced this gradient: one pass for its maximum (fp16 two-piece kernels)
            wgrad(blend.conv, z, g_x, 1, 1, pro_of(z, t, bn))
            g_z, bs = ops.conv_dgrad(g_x, wp(blend.conv), t.shape[1:3], 1, 1, bnb=(t, None, bn, True))
            g_t, _ = bn_bwd(blend.norm, g_z, t, bn, relu=True, sums=bs)
            g_skip[5 - i] = g_t                      # skips[idx + li] with idx + li = 5 - i
            g_x = ops.upsample_bwd(g_t, in_hw[0], in_hw[1])
        g_skip[5] = g_x                              # coarsest map: level 2, layer4
        # ---- encoder: the levels in lockstep (reverse creation order inside every step, like a level-by-level walk of
        # the tape: the shared weights and BatchNorm parameters accumulate their gradients in the same order) ----
        dwst = None
        if pos >= 0:
            assert tape[pos][0] == "levels" and pos == 0
            ltapes = tape[pos][1]
            nl = len(ltapes)
            steps = len(ltapes[0])
            assert all(len(t) == steps for t in ltapes)
            g_curs = [None] * nl
            sums_l = [None] * nl       # BatchNorm-backward sums of g_cur for the block that consumes it next (or None)

            def consumer_bnb(lt, at):
                """The next tape item of the level, if it is a block: g_cur written now is final and goes into that
                block's bn2 backward (mask = sign of the block output): its two sums can ride the epilogue of the kernel
                that writes g_cur last."""
                if at >= 0 and lt[at][0] == "block":
                    _, _, _, _, _, _, y2n, bn2n, _, _, outn = lt[at]
                    return (y2n, outn, bn2n, False)
                return None

            with ops.level_batch() as lb:
                for at in range(steps - 1, -1, -1):
                    for idx in range(nl - 1, -1, -1):
                        lb.level(idx)
                        lt = ltapes[idx]
                        item = lt[at]
                        kind = item[0]
                        g_cur, cur_sums = g_curs[idx], sums_l[idx]
                        if kind == "skip":
                            _, lvl, x, bott = item
                            gs = g_skip[lvl]
                            wgrad(bott, x, gs, 1, 0)
                            # the skip projection's data gradient is the LAST writer of the gradient of this layer's output
                            cb = consumer_bnb(lt, at - 1)
                            if g_cur is None:
                                r = ops.conv_dgrad(gs, wp(bott), x.shape[1:3], 1, 0, bnb=cb)
                            else:
                                r = ops.conv_dgrad(gs, wp(bott), x.shape[1:3], 1, 0, out=g_cur, accumulate=True, bnb=cb)
                            g_cur, cur_sums = r if cb is not None else (r, None)
                        elif kind == "block":
                            _, blk, x, y1, bn1, z1, y2, bn2, yd, bnd, out = item
                            s = blk.stride
                            dy2, gm = bn_bwd(blk.bn2, g_cur, y2, bn2, masksrc=out, want_gm=True, sums=cur_sums)
                            cur_sums = None
                            if training:       # activation-checkpoint recompute side effect
                                ops.bn_ema_again(bn2, blk.bn2.running_mean, blk.bn2.running_var,
                                                 y2.numel() // y2.shape[-1], momentum=blk.bn2.momentum)
                                self._nbt.append(blk.bn2)
                            wgrad(blk.conv2, z1, dy2, 1, 1, pro_of(z1, y1, bn1))
                            g_z1, s1 = ops.conv_dgrad(dy2, wp(blk.conv2), z1.shape[1:3], 1, 1, bnb=(y1, None, bn1, True))
                            dy1, _ = bn_bwd(blk.bn1, g_z1, y1, bn1, relu=True, sums=s1)
                            if training:
                                ops.bn_ema_again(bn1, blk.bn1.running_mean, blk.bn1.running_var,
                                                 y1.numel() // y1.shape[-1], momentum=blk.bn1.momentum)
                                self._nbt.append(blk.bn1)
                            wgrad(blk.conv1, x, dy1, s, 1)
                            # conv1's data gradient is the last writer of this block's input gradient unless a skip
                            # projection follows (first block of a layer: the next tape item is then "skip", not "block")
                            cb = consumer_bnb(lt, at - 1)
                            if blk.downsample is not None:
                                dyd, _ = bn_bwd(blk.downsample[1], gm, yd, bnd)
                                wgrad(blk.downsample[0], x, dyd, s, 0)
                            if blk.downsample is not None and cb is None:
                                # conv1 (3x3) writes every pixel of the input gradient, the 1x1 / stride-s projection
                                # then adds its one parity class: no zero fill of the other three, a quarter of the
                                # accumulate reads (the sum of the two terms is the same either way round)
                                g_in = ops.conv_dgrad(dy1, wp(blk.conv1), x.shape[1:3], s, 1)
                                g_cur = ops.conv_dgrad(dyd, wp(blk.downsample[0]), x.shape[1:3], s, 0, out=g_in,
                                                       accumulate=True)
                                cur_sums = None
                            else:
                                if blk.downsample is not None:
                                    g_in = ops.conv_dgrad(dyd, wp(blk.downsample[0]), x.shape[1:3], s, 0)
                                else:
                                    g_in = gm
                                r = ops.conv_dgrad(dy1, wp(blk.conv1), x.shape[1:3], s, 1, out=g_in, accumulate=True, bnb=cb)
                                g_cur, cur_sums = r if cb is not None else (r, None)
                        elif kind == "stem":
                            _, _, p, y, bn, pidx, bnm = item
                            acc = bnm.weight in grads
                            if not acc:
                                grads[bnm.weight] = self._galloc(bnm.weight)
                                grads[bnm.bias] = self._galloc(bnm.bias)
                            dy = ops.bn_pool_bwd(g_cur, pidx, y, bn, bnm.weight, dgamma=grads[bnm.weight],
                                                 dbeta=grads[bnm.bias], acc_param=acc, training=training)
                            if dwst is None:
                                dwst = torch.empty((64, 7, 8, 4), device=dy.device, dtype=dy.dtype)
                                ops.stem_wgrad(p, dy, dwst, False)
                            else:
                                ops.stem_wgrad(p, dy, dwst, True)
                            g_cur = None
                        else:  # pragma: no cover
                            raise RuntimeError(kind)
                        g_curs[idx], sums_l[idx] = g_cur, cur_sums
                    lb.flush()
        if dwst is not None:
            grads[self.fe.conv1.weight] = ops.unpack_stem_weight(dwst, self.fe.conv1.weight,
                                                                 out=self._galloc(self.fe.conv1.weight))
        self._flush_nbt()
        return grads


class _SwiftNetFn(torch.autograd.Function):
    """Outputs: seg (NCHW), before (NHWC, channel stride 20), fine_feat (NHWC), fine_feat0 (= fine_feat[:B], a
    view that is its own autograd output: the pixel-contrast gradient then arrives separately instead of going
    through autograd's split-backward (a 2 GB cat + add per step at C3))."""

    rows_output = 3        # index of fine_feat0 among the outputs: the one that takes row gradients (losses._row_sink)

    @staticmethod
    def accept_rows(ctx, rowidx, rows):
        """Gradient of fine_feat0 given as rows [A, C] at pixel rows ``rowidx`` (int32 into [B*h*w]); added in backward.
        ``ctx`` is the autograd node of one forward call (its ``_forward_cls`` is this class)."""
        ctx.row_grads = (getattr(ctx, "row_grads", None) or []) + [(rowidx, rows)]

    @staticmethod
    def forward(ctx, engine: SwiftNetEngine, img, training, supcon, grad_enabled, lazy_seg, *params):
        need_grad = grad_enabled and any(p.requires_grad for p in params)
        ctx.set_materialize_grads(False)
        seg, before, ff, saved = engine.forward(img, training, supcon, need_grad, lazy_seg)
        ctx.engine, ctx.saved, ctx.params = engine, saved, params
        B = ff.shape[0] // 2 if supcon else ff.shape[0]
        ff0 = ff[:B] if supcon else ff.view(ff.shape)
        outs = [t if t is not None else ff.new_zeros(1) for t in (seg, before)] + [ff, ff0]
        ctx.has = [seg is not None, before is not None]
        ctx.mark_non_differentiable(*[o for o, h in zip(outs[:2], ctx.has) if not h])
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_seg, g_before, g_ff, g_ff0):
        if ctx.saved is None:
            raise RuntimeError("SwiftNet backward without a recorded forward")
        # both gradient buffers are consumed in place: they are freshly produced by the loss nodes
        if g_ff is not None:
            g_ff = g_ff.contiguous()
        row_grads = getattr(ctx, "row_grads", None) or []
        ctx.row_grads = None
        if g_ff0 is not None and row_grads and all(st == 0 for st in g_ff0.stride()):
            g_ff0 = None                        # the memory-less zero of losses._PixelContrastFn: its rows came separately
        if g_ff0 is not None:
            g_ff0 = g_ff0.contiguous()
            if g_ff is None:
                if g_ff0.shape[0] == ctx.saved.Bm:
                    g_ff = g_ff0
                else:
                    g_ff = torch.zeros((ctx.saved.Bm,) + tuple(g_ff0.shape[1:]), device=g_ff0.device, dtype=g_ff0.dtype)
                    g_ff[:g_ff0.shape[0]].copy_(g_ff0)
            else:
                ops.axpy(g_ff[:g_ff0.shape[0]], g_ff0, 1.0)
        if row_grads:
            if g_ff is None:
                rows0 = row_grads[0][1]
                g_ff = torch.zeros(ctx.saved.ff_shape, device=rows0.device, dtype=rows0.dtype)
            for rowidx, rows in row_grads:
                ops.scatter_add_rows(rows, rowidx, g_ff)
        if g_seg is not None:
            g_seg = g_seg.contiguous()
        grads = ctx.engine.backward(ctx.saved, g_seg if ctx.has[0] else None,
                                    g_before if (ctx.has[1] and g_before is not None) else None, g_ff)
        flat = ctx.engine.flat
        ctx.saved = None
        res = []
        for p in ctx.params:
            gp = grads.get(p)
            if gp is not None and flat is not None and flat.grad_view.get(p) is gp:
                # gradient already sits in the flat buffer: publish it as .grad without autograd's copy
                if p.grad is None:
                    p.grad = gp
                elif p.grad.data_ptr() != gp.data_ptr():
                    p.grad.add_(gp)
                res.append(None)
            else:
                res.append(gp)
        return (None, None, None, None, None, None) + tuple(res)


class WeatherNet(nn.Module):
    """Drop-in for network.WeatherNet (network/weathernet.py:14-104): same constructor, same
    4-tuple forward, same state_dict; ``backbone`` must be 'resnet18'."""

    def __init__(self, opts, num_downsample=2, num_classes=19, device=None, feature_similarity="correlation",
                 aggregation_type="adaptive", num_scales=3, backbone="resnet34", train_semantic=True):
        super().__init__()
        self.opts = opts
        self.num_downsample = num_downsample
        self.aggregation_type = aggregation_type
        self.num_scales = num_scales
        self.num_classes = num_classes
        self.device = device
        if backbone == "resnet18":
            self.feature_extractor = ResNetPyramid((2, 2, 2, 2))
        elif backbone == "resnet34":
            self.feature_extractor = ResNetPyramid((3, 4, 6, 3))
        else:
            raise NotImplementedError
        # resnet18_pyramid(pretrained=True) (resnet_pyramid.py:404): the reference downloads the ImageNet weights; here
        # the same file is used when it is available LOCALLY (opts.pretrained_backbone_path, $DCS_IMAGENET_DIR or the
        # torch hub cache), otherwise the backbone keeps its random initialisation (no network in this environment).
        from .checkpoint import find_cached_imagenet, load_imagenet_backbone
        path = getattr(opts, "pretrained_backbone_path", None) or find_cached_imagenet(backbone)
        self.pretrained_from = path
        if path:
            load_imagenet_backbone(self.feature_extractor, path)
        self.segmentation = None
        if train_semantic:
            self.segmentation = _BNReluConv(self.feature_extractor.num_features, self.num_classes, k=1, bias=True)
            self.loss_ret_additional = False
            self.img_req_grad = False
            self.upsample_logits = True
            self.multiscale_factors = (.5, .75, 1.5, 2.)
        self._engine = None

    def _get_engine(self):
        if self._engine is None:
            object.__setattr__(self, "_engine", SwiftNetEngine(self.feature_extractor, self.segmentation,
                                                               self.num_classes))
        return self._engine

    def _graphed_eval(self, parts, supcon):
        """Eval-mode forward without autograd as ONE hipGraph launch (opts.eval_graph / DCS_EVAL_GRAPH=1).

        The inference forward of a single 2048x1024 image is ~250 kernels of 5-50 us each: issued one by one the step is
        bounded by the host (Python + launch path, ~15 us per kernel), not by the device.  The whole forward is therefore
        captured once per (input shapes, parameter storage) into a HIP graph -- the library's launches go to torch's
        current stream, so stream capture records them like any torch kernel; level batches are merged at capture time --
        and later calls copy the images into the graph's static input and replay it.  Parameters and BatchNorm running
        statistics are read through their own storage at every replay (an optimizer step in between is seen).  The
        OUTPUTS are the graph's static tensors: they are overwritten by the next call with the same shapes (the validate
        loop, trainer.py:303-402 of the reference, consumes them at once)."""
        key = (tuple(tuple(t.shape) for t in parts), bool(supcon), next(self.parameters()).data_ptr(),
               self.segmentation is not None)
        cache = self.__dict__.setdefault("_eval_graphs", {})
        ent = cache.get(key)
        engine = self._get_engine()
        if ent is None:
            static_in = [t.clone() for t in parts]
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                   # every kernel variant is loaded before the capture
                engine.forward(static_in, False, supcon, False, False)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                outs = engine.forward(static_in, False, supcon, False, False)
            ent = cache[key] = (graph, static_in, outs)
            if len(cache) > 8:                              # a handful of resolutions at most (graphs hold their activations)
                cache.pop(next(iter(cache)))
        graph, static_in, outs = ent
        for d, t in zip(static_in, parts):
            d.copy_(t)
        graph.replay()
        return outs[:3]

    def forward(self, left_img, return_supcon_feature=False):
        """left_img: [Bm,3,H,W] raw 0-255 image batch, or a list of batch parts (e.g. the two crops of the
        ``supcon*`` criteria, equivalent to their torch.cat along dim 0 without the copy)."""
        for t in (left_img if isinstance(left_img, (list, tuple)) else [left_img]):
            ops.require_device(t, "left_img")
        if (not self.training and not torch.is_grad_enabled() and
                (getattr(self.opts, "eval_graph", False) or os.environ.get("DCS_EVAL_GRAPH", "0") == "1")):
            parts = list(left_img) if isinstance(left_img, (list, tuple)) else [left_img]
            parts = [t if t.is_floating_point() else t.float() for t in parts]
            seg, before, ff = self._graphed_eval(parts, bool(return_supcon_feature))
            B = ff.shape[0] // 2 if return_supcon_feature else ff.shape[0]
            fine_feat = ff.permute(0, 3, 1, 2)
            fine_feat0 = ff[:B].permute(0, 3, 1, 2)
            if self.segmentation is None:
                return None, None, fine_feat, fine_feat0
            return seg, before[..., :self.num_classes].permute(0, 3, 1, 2), fine_feat, fine_feat0
        params = [p for p in self.parameters()]
        # training with autograd on: pred_segmap is a LazyLogits handle (losses.py) -- the criteria evaluate it fused,
        # any other use materialises it; opts.lazy_pred_segmap = False restores the eager 2.55 GB tensor
        lazy = bool(self.training and torch.is_grad_enabled() and self.segmentation is not None and
                    getattr(self.opts, "lazy_pred_segmap", True))
        seg, before, ff, ff0 = _SwiftNetFn.apply(self._get_engine(), left_img, self.training,
                                                 bool(return_supcon_feature), torch.is_grad_enabled(), lazy, *params)
        # NHWC buffers exposed with the reference's logical NCHW shapes (channels_last strides, no copy);
        # fine_feat0 is the first half of fine_feat (weathernet.py:78-82) and shares its memory
        fine_feat = ff.permute(0, 3, 1, 2)
        fine_feat0 = ff0.permute(0, 3, 1, 2)
        if self.segmentation is None:
            return None, None, fine_feat, fine_feat0
        pred_segmap_beforeup = before[..., :self.num_classes].permute(0, 3, 1, 2)
        if lazy:
            from .losses import LazyLogits
            first = left_img[0] if isinstance(left_img, (list, tuple)) else left_img
            seg = LazyLogits(before, self.num_classes, first.shape[-2:])
        return seg, pred_segmap_beforeup, fine_feat, fine_feat0

    def random_init_params(self):
        return self.feature_extractor.random_init_params()

    def fine_tune_params(self):
        return self.feature_extractor.fine_tune_params()

    def flatten_parameters(self):
        """Optional (call after ``.to(device)``): re-home the parameters of the two ADAM groups
        (utils/init_trainer.py:169-177) and of the segmentation head in flat buffers, see FlatBuffers."""
        rest = [p for p in (self.segmentation.parameters() if self.segmentation is not None else [])]
        flat = FlatBuffers([list(self.random_init_params()), list(self.fine_tune_params()), rest])
        flat.adopt_counters(self.modules())
        self._get_engine().flat = flat
        return flat


class WeatherClassifier(nn.Module):
    """Drop-in for network.WeatherClassifier (network/classifier.py:6-32): GAP + Linear."""

    def __init__(self, opts, weather_class_num):
        super().__init__()
        self.opts = opts
        num_channels = 2048 if getattr(opts, "deeplab", False) else 128
        self.pool_attention = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(num_channels, weather_class_num)

    def forward(self, x):
        from .losses import global_avg_pool
        pooled = global_avg_pool(x)                  # HIP reduction, [B, C]
        return ops.linear(pooled.detach().contiguous(), self.fc.weight.detach().contiguous(),
                          self.fc.bias.detach().contiguous())
