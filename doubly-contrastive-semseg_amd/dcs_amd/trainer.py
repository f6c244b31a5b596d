"""Train-step harness: the body of the reference's ``Trainer.train`` loop
(trainer.py:62-215) and the pieces of ``InitOpts`` it relies on
(utils/init_trainer.py:97-113 model, :163-177 ADAM groups, :215-223 criteria).

``TrainStep.step(sample)`` takes what ``dataloaders/`` would deliver (dicts with
'left', 'label', 'weather', 'label_distance_weight'; a pair of dicts for the
``supcon*`` criteria) and performs forward, the criterion switch, backward and
the optimizer step on one MI355X.
"""
from __future__ import annotations

import types
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import ops
from .losses import (BoundaryAwareFocalLoss, FocalLoss2, PixelContrastLoss, SemsegCrossEntropy, SupConLoss)
from .model import WeatherClassifier, WeatherNet

CRITERIA = ("supcon_focal", "supcon_simclr_focal", "pixelcontrast_focal", "supcon_pixelcontrast_focal",
            "supcon_simclr_pixelcontrast_focal", "crossentropy", "supcon_crossentropy",
            "supcon_simclr_cross_entropy", "focal", "plain_focal", "none")


def make_opts(**kw):
    """The subset of options.py the hot path reads, with the reference's defaults."""
    d = dict(model="resnet18", deeplab=False, criterion="supcon_pixelcontrast_focal", batch_size=8, lr=4e-4,
             weight_decay=1e-4, optimizer_policy="ADAM", dataset="acdc", weather_num=4, num_classes=19,
             train_semantic=True, with_depth_level_loss=False, no_class_weights=False, no_EDT=False,
             epochs=400, last_lr=1e-6, output_stride=16)
    d.update(kw)
    return types.SimpleNamespace(**d)


class DcsAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (L2 weight decay folded into the gradient) on the fused HIP kernel."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, flat=None):
        # same group keys as torch.optim.Adam so that state_dict() round-trips with the reference's optimizer
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False,
                                      maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                                      decoupled_weight_decay=False))
        self.flat = flat                      # model.FlatBuffers or None
        self._flat_state = {}

    # ---- torch.optim.Adam-compatible state (utils/init_trainer.py:258 restores it, trainer.py:417 saves it) ----
    def state_dict(self):
        """Flat moment buffers are exposed as per-parameter ``exp_avg`` / ``exp_avg_sq`` / ``step`` entries in the
        reference's layout (stand-alone contiguous tensors)."""
        synced = []
        for gi, group in enumerate(self.param_groups):
            st, fg = self._flat_state.get(gi), self._flat_group(group)
            if st is None or fg is None:
                continue
            off = 0
            for p in group["params"]:
                self.state[p] = dict(step=torch.tensor(float(st["step"])),
                                     exp_avg=torch.as_strided(st["m"], p.size(), p.stride(), off),
                                     exp_avg_sq=torch.as_strided(st["v"], p.size(), p.stride(), off))
                synced.append(p)
                off += p.numel()
        sd = super().state_dict()
        for p in synced:                      # the flat buffers stay the only live copy of the moments
            del self.state[p]
        for entry in sd["state"].values():
            for k, v in entry.items():
                if torch.is_tensor(v):
                    entry[k] = v.detach().clone(memory_format=torch.contiguous_format)
            if not torch.is_tensor(entry.get("step")):
                entry["step"] = torch.tensor(float(entry["step"]))
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._flat_state = {}
        for gi, group in enumerate(self.param_groups):
            fg = self._flat_group(group)
            if fg is not None and all("exp_avg" in self.state.get(p, {}) for p in group["params"]):
                m, v = torch.zeros_like(fg["flat_p"]), torch.zeros_like(fg["flat_p"])
                off, steps = 0, set()
                for p in group["params"]:
                    s = self.state[p]
                    torch.as_strided(m, p.size(), p.stride(), off).copy_(s["exp_avg"])
                    torch.as_strided(v, p.size(), p.stride(), off).copy_(s["exp_avg_sq"])
                    steps.add(int(s["step"]))
                    off += p.numel()
                    del self.state[p]
                if len(steps) != 1:
                    raise ValueError("parameters of one ADAM group carry different step counts: %s" % sorted(steps))
                self._flat_state[gi] = dict(step=steps.pop(), m=m, v=v)
                continue
            for p in group["params"]:         # per-parameter path: moments in the parameter's own memory layout
                s = self.state.get(p)
                if s and "exp_avg" in s:
                    for k in ("exp_avg", "exp_avg_sq"):
                        if s[k].stride() != p.stride():
                            s[k] = torch.empty_like(p).copy_(s[k])
                    s["step"] = int(s["step"])

    def _flat_group(self, group):
        if self.flat is None:
            return None
        for fg in self.flat.groups:
            if len(fg["params"]) == len(group["params"]) and all(a is b for a, b in zip(fg["params"], group["params"])):
                return fg
        return None

    @torch.no_grad()
    def step(self, closure=None):
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            fg = self._flat_group(group)
            if fg is not None and self.flat.aliased(group["params"]):
                # every gradient of the group lives in the flat buffer: ONE fused launch for the whole group
                st = self._flat_state.setdefault(gi, dict(step=0, m=torch.zeros_like(fg["flat_p"]),
                                                           v=torch.zeros_like(fg["flat_p"])))
                st["step"] += 1
                ops.adam_step(fg["flat_p"], fg["flat_g"], st["m"], st["v"], float(group["lr"]), b1, b2, group["eps"],
                              float(group["weight_decay"]), st["step"])
                continue
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                g = p.grad
                if g.stride() != p.stride():
                    g = g.contiguous(memory_format=torch.channels_last) if p.dim() == 4 else g.contiguous()
                ops.adam_step(p, g, st["exp_avg"], st["exp_avg_sq"], float(group["lr"]), b1, b2, group["eps"],
                              float(group["weight_decay"]), st["step"])


class TrainStep:
    def __init__(self, opts, class_weight: Optional[torch.Tensor] = None, device="cuda"):
        self.opts = opts
        self.device = torch.device(device)
        opts.weight = class_weight
        if getattr(opts, "deeplab", False):                          # utils/init_trainer.py:99-102
            from . import deeplab
            self.model = getattr(deeplab, opts.model)(opts, num_classes=opts.num_classes,
                                                      output_stride=opts.output_stride).to(self.device)
        else:
            self.model = WeatherNet(opts, num_classes=opts.num_classes, device=self.device, backbone=opts.model,
                                    train_semantic=opts.train_semantic).to(self.device)
        self.weather_clf = WeatherClassifier(opts, weather_class_num=opts.weather_num).to(self.device)
        w = class_weight
        self.criterion = BoundaryAwareFocalLoss(gamma=0.5, num_classes=opts.num_classes, ignore_id=255, weight=w,
                                                device=self.device, opts=opts)
        self.focal_criterion = FocalLoss2(gamma=.5, num_classes=opts.num_classes, ignore_id=255, weight=w,
                                          device=self.device)
        self.supcon_criterion = SupConLoss(temperature=0.07, contrast_mode="all", base_temperature=0.07, weight=w,
                                           device=self.device, opts=opts)
        self.pixelcontrast_criterion = PixelContrastLoss(device=self.device)
        self.ce_criterion = SemsegCrossEntropy(num_classes=opts.num_classes, ignore_id=255)
        fine_tune_factor = 4                                       # utils/init_trainer.py:169-177
        self.flat = self.model.flatten_parameters() if getattr(opts, "flat_params", True) else None
        if getattr(opts, "deeplab", False):                          # utils/init_trainer.py:163-168: one group
            groups = [{"params": list(self.model.parameters()), "lr": opts.lr, "weight_decay": opts.weight_decay}]
        else:
            groups = [
                {"params": list(self.model.random_init_params()), "lr": opts.lr, "weight_decay": opts.weight_decay},
                {"params": list(self.model.fine_tune_params()), "lr": opts.lr / fine_tune_factor,
                 "weight_decay": opts.weight_decay / fine_tune_factor}]
        self.optimizer = DcsAdam(groups, betas=(0.9, 0.99), flat=self.flat)
        owned = {id(p) for g in groups for p in g["params"]}
        self._unowned = [p for p in self.model.parameters() if id(p) not in owned]
        self.num_iter = 0
        self.model.train()

    def step(self, sample, do_optimizer_step=True) -> Dict[str, torch.Tensor]:
        o = self.opts
        crit = o.criterion
        dev = self.device
        dt = getattr(o, "dtype", torch.float32)
        if "supcon" in crit:                                       # trainer.py:66-72: cat of the two crops ...
            sample0, sample1 = sample
            sample = dict(sample0)
            # ... expressed as a list of batch parts: the model normalises both crops into one batch directly
            left = [sample0["left"].to(dev, dtype=dt), sample1["left"].to(dev, dtype=dt)]
        else:
            left = sample["left"].to(dev, dtype=dt)
        self.num_iter += 1
        labels = sample["label"].to(dev, dtype=torch.long)
        gt_weather = sample["weather"].to(dev) if "weather" in sample else None
        supcon_flag = "supcon" in crit
        left_seg, left_seg_beforeup, fine_feat, fine_feat0 = self.model(left, return_supcon_feature=supcon_flag)
        if "pixelcontrast" in crit:
            # start the sampler's counting kernel + D2H copy now so the host wait overlaps the other losses
            self.pixelcontrast_criterion.prefetch(fine_feat0, labels, left_seg_beforeup)
        zero = torch.zeros(1, device=dev)
        out = dict(supcon=zero, simclr=zero, pixel=zero, seg=zero, ce=zero)
        if o.dataset == "acdc" and gt_weather is not None:        # trainer.py:109-114 (logged only)
            out["pred_weather"] = self.weather_clf(fine_feat0)
        if crit == "supcon_focal":
            out["supcon"] = self.supcon_criterion(fine_feat, class_labels=gt_weather, mask=None)
            out["seg"] = self.criterion(left_seg, labels, sample)
            total = out["supcon"] * 1 / o.batch_size + out["seg"] * 1.2
        elif crit == "supcon_simclr_focal":
            out["simclr"] = self.supcon_criterion(fine_feat, class_labels=None, mask=None)
            out["seg"] = self.criterion(left_seg, labels, sample)
            total = out["simclr"] * 1 / o.batch_size + out["seg"] * 1.2
        elif crit == "pixelcontrast_focal":
            # The sampler's host half (the reference's CPU randperm stream, ~6 ms at C3) runs while the device works on
            # everything that does not need the anchors: the counting kernel + D2H copy were queued by prefetch() above
            # (on the labels BEFORE the focal loss rewrites 255 -> 0), so the segmentation loss goes first.
            out["seg"] = self.criterion(left_seg, labels, sample)
            out["pixel"] = self.pixelcontrast_criterion(fine_feat0, labels=labels, predict=left_seg_beforeup)
            total = out["pixel"] * 1 / o.batch_size + out["seg"] * 1.2
        elif crit == "supcon_pixelcontrast_focal":
            out["supcon"] = self.supcon_criterion(fine_feat, class_labels=gt_weather, mask=None)
            out["seg"] = self.criterion(left_seg, labels, sample)          # before the pixel loss: see pixelcontrast_focal
            out["pixel"] = self.pixelcontrast_criterion(fine_feat0, labels=labels, predict=left_seg_beforeup)
            total = 1 / o.batch_size * (out["supcon"] + out["pixel"]) + out["seg"] * 1.2
        elif crit == "supcon_simclr_pixelcontrast_focal":
            out["simclr"] = self.supcon_criterion(fine_feat, class_labels=None, mask=None)
            out["seg"] = self.criterion(left_seg, labels, sample)
            out["pixel"] = self.pixelcontrast_criterion(fine_feat0, labels=labels, predict=left_seg_beforeup)
            total = 1 / o.batch_size * (out["simclr"] + out["pixel"]) + out["seg"] * 1.2
        elif crit == "crossentropy":
            out["ce"] = self.ce_criterion(left_seg, labels)
            total = out["ce"]
        elif crit == "supcon_crossentropy":
            out["supcon"] = self.supcon_criterion(fine_feat, class_labels=gt_weather, mask=None)
            out["ce"] = self.ce_criterion(left_seg, labels)
            total = out["ce"] + out["supcon"]
        elif crit == "supcon_simclr_cross_entropy":                 # trainer.py:193-198 adds the zero supcon_loss
            out["simclr"] = self.supcon_criterion(fine_feat, class_labels=None, mask=None)
            out["ce"] = self.ce_criterion(left_seg, labels)
            total = out["ce"] + out["supcon"]
        else:
            out["seg"] = self.criterion(left_seg, labels, sample)
            total = out["seg"]
        self.optimizer.zero_grad()
        self.supcon_criterion.zero_grad()
        # parameters in no ADAM group (the segmentation head): the reference never resets their .grad
        # (it accumulates unused for the whole run); here they are reset so that the gradient lands in the flat buffer
        # directly and the data-parallel all-reduce never re-reduces an old sum
        for p in self._unowned:
            p.grad = None
        total.backward()
        if do_optimizer_step:
            self.optimizer.step()
        out.update(total=total.detach(), left_seg=left_seg, left_seg_beforeup=left_seg_beforeup,
                   fine_feat=fine_feat, labels=labels)
        return out
