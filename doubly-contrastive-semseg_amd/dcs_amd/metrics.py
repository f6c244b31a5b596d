"""Evaluator with the reference's interface (metrics/stream_metrics.py:136-342) whose confusion matrices are
accumulated on the GPU: ``add_batch_device`` fuses the class-id argmax (trainer.py:349) with the histogram
(:330-342), optionally straight from the low-resolution logits, so ``Trainer.validate`` needs no per-batch
D2H copy of [B,H,W] predictions.  The numpy ``add_batch`` of the reference is kept for host-side callers."""
from __future__ import annotations

import numpy as np
import torch

from . import ops


class Evaluator(object):
    def __init__(self, num_class, weather_num, device=None):
        self.num_class = num_class
        self.weather_num = weather_num
        self.device = device
        self.reset()

    def reset(self):
        c = self.num_class
        self._cm = np.zeros((c,) * 2)
        self._cm_w = {str(w): np.zeros((c,) * 2) for w in range(self.weather_num)}
        self.confusion_matrix_weather = np.zeros((self.weather_num,) * 2)
        self.weather_acc = torch.tensor([])
        self._pending = []                 # (device per-image matrices [N,C,C], weather ids) not yet folded in

    # the reference exposes these as plain attributes; reading them folds in pending device-side counts
    @property
    def confusion_matrix(self):
        self._flush()
        return self._cm

    @property
    def confusion_matrix_sem_weather(self):
        self._flush()
        return self._cm_w

    # ---- reference host path (metrics/stream_metrics.py:330-342) ----
    def _generate_matrix(self, gt_image, pre_image):
        mask = (gt_image >= 0) & (gt_image < self.num_class)
        label = self.num_class * gt_image[mask].astype("int") + pre_image[mask]
        count = np.bincount(label, minlength=self.num_class ** 2)
        return count.reshape(self.num_class, self.num_class)

    def add_batch(self, gt_image, pre_image, gt_weather):
        assert gt_image.shape == pre_image.shape
        self._cm += self._generate_matrix(gt_image, pre_image)
        for i, wea in enumerate(gt_weather):
            self._cm_w[str(int(wea))] += self._generate_matrix(gt_image[i], pre_image[i])

    # ---- device path ----
    def add_batch_device(self, labels, logits, gt_weather=None, lowres=False):
        """labels int64 [N,H,W] on the GPU; logits = model output ``pred_segmap`` [N,C,H,W], or with lowres=True the
        ``pred_segmap_beforeup`` view [N,C,h,w] (upsampled on the fly).  Nothing is synchronised here."""
        N, H, W = labels.shape
        conf = torch.zeros((N, self.num_class, self.num_class), device=labels.device, dtype=torch.int64)
        if lowres:
            v = logits.detach().permute(0, 2, 3, 1)
            cs = v.stride(2)
            if not (v.stride(3) == 1 and v.stride(1) == v.shape[2] * cs and v.stride(0) == v.shape[1] * v.shape[2] * cs):
                v = v.contiguous()
                cs = v.shape[3]
            base = torch.as_strided(v, (N, v.shape[1], v.shape[2], cs), (v.shape[1] * v.shape[2] * cs, v.shape[2] * cs, cs, 1))
            ops.confusion(base, labels.contiguous(), self.num_class, conf, lowres=(H, W))
        else:
            ops.confusion(logits.detach().contiguous(), labels.contiguous(), self.num_class, conf)
        self._pending.append((conf, None if gt_weather is None else gt_weather.detach().view(-1)))

    def _flush(self):
        for conf, wea in self._pending:
            c = conf.cpu().numpy().astype(np.float64)
            self._cm += c.sum(0)
            if wea is not None:
                for i, wi in enumerate(wea.cpu().tolist()):
                    self._cm_w[str(int(wi))] += c[i]
        self._pending = []

    def add_batch_weather(self, gt_weather, weather_pred):
        _, preds = torch.max(weather_pred, dim=1)
        acc = torch.tensor([torch.sum(preds == gt_weather.view(-1)).item() / len(preds)])
        for t, p in zip(gt_weather.view(-1), preds.view(-1)):
            self.confusion_matrix_weather[int(t), int(p)] += 1
        self.weather_acc = torch.cat((self.weather_acc, acc))

    # ---- scores (metrics/stream_metrics.py:162-245, :321-328), without the reference's console/file prints ----
    def Pixel_Accuracy(self):
        return np.diag(self.confusion_matrix).sum() / self.confusion_matrix.sum()

    def Pixel_Accuracy_Class(self):
        with np.errstate(divide="ignore", invalid="ignore"):
            acc = np.diag(self.confusion_matrix) / self.confusion_matrix.sum(axis=1)
        return np.nanmean(acc)

    def _iou(self, cf):
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.diag(cf) / (np.sum(cf, axis=1) + np.sum(cf, axis=0) - np.diag(cf))

    def Mean_Intersection_over_Union(self, save_filename=None):
        iou = self._iou(self.confusion_matrix)
        if save_filename is not None:
            with open(save_filename, "a") as f:
                f.write("-----------IoU of each class-----------\n")
                for i, v in enumerate(iou):
                    f.write("class %2d      : %.6f\n" % (i, v * 100.0))
        return np.nanmean(iou)

    def Mean_Intersection_over_Union_each_weather(self, save_filename=None):
        return {str(w): np.nanmean(self._iou(self.confusion_matrix_sem_weather[str(w)]) * 100.0)
                for w in range(self.weather_num)}

    def Frequency_Weighted_Intersection_over_Union(self):
        cf = self.confusion_matrix
        freq = np.sum(cf, axis=1) / np.sum(cf)
        iu = self._iou(cf)
        return (freq[freq > 0] * iu[freq > 0]).sum()
