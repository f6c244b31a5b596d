#!/usr/bin/env python3
"""Golden vectors for the validate-side row from the reference's own Evaluator
(metrics/stream_metrics.py is importable here).  Stores the synthetic pred/target pair's expected confusion
matrix and scores."""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
from metrics.stream_metrics import Evaluator  # noqa: E402

g = np.random.default_rng(91)
N, H, W, C = 3, 40, 56, 19
gt = g.integers(0, C, size=(N, H, W)).astype(np.int64)
gt[:, :4, :] = 255
gt[1, 10:14, :] = 19
pred = np.where(g.random((N, H, W)) < 0.6, gt.clip(0, C - 1), g.integers(0, C, size=(N, H, W))).astype(np.int64)
wea = np.array([0, 2, 2])
ev = Evaluator(C, 4)
ev.add_batch(gt, pred, wea)
with contextlib.redirect_stdout(io.StringIO()):
    miou = ev.Mean_Intersection_over_Union("/tmp/_dcs_val.txt")
    acc_cls = ev.Pixel_Accuracy_Class()
    per_w = ev.Mean_Intersection_over_Union_each_weather("/tmp/_dcs_val.txt")
np.savez_compressed(os.path.join(HERE, "metrics_evaluator.npz"), gt=gt.astype(np.int16), pred=pred.astype(np.uint8), weather=wea,
                    confusion=ev.confusion_matrix, conf_w0=ev.confusion_matrix_sem_weather["0"],
                    conf_w2=ev.confusion_matrix_sem_weather["2"], miou=miou, acc=ev.Pixel_Accuracy(), acc_cls=acc_cls,
                    fwiou=ev.Frequency_Weighted_Intersection_over_Union(), miou_w0=per_w["0"], miou_w2=per_w["2"])
print("metrics golden", miou, ev.Pixel_Accuracy())
