#!/usr/bin/env python3
"""Validate-side throughput (Trainer.validate, trainer.py:303-402): eval-mode forward at full resolution + fused
argmax / confusion-matrix kernel straight from the low-resolution logits, launch by launch and as one hipGraph launch
per forward (DCS_EVAL_GRAPH=1, model._graphed_eval).  usage: eval_bench.py [batch ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "doubly-contrastive-semseg_amd")):
    sys.path.insert(0, p)
import torch
from dcs_amd.metrics import Evaluator
from dcs_amd.model import WeatherNet
from dcs_amd.trainer import make_opts
from oracle import swiftnet_oracle as O

dev = torch.device("cuda", 0)
model = WeatherNet(make_opts(), num_classes=19, device=dev, backbone="resnet18", train_semantic=True).to(dev).eval()
for b, graph in [(int(a), g) for a in (sys.argv[1:] or [1, 4, 16]) for g in ("0", "1")]:
    os.environ["DCS_EVAL_GRAPH"] = graph
    img, labels, _, weather, _ = O.synthetic_batch(b, 1024, 2048, seed=b, cell=64)
    img, labels, weather = img.to(dev), labels.to(dev), weather.to(dev)
    ev = Evaluator(19, 4)
    with torch.no_grad():
        for _ in range(3):
            seg, before, ff, ff0 = model(img)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            seg, before, ff, ff0 = model(img)
            ev.add_batch_device(labels, before, weather, lowres=True)
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"eval batch {b} ({'one hipGraph launch' if graph == '1' else 'launch by launch'}): {ms:.2f} ms per batch = {b / ms * 1e3:.1f} images/s  (mIoU on random weights {ev.Mean_Intersection_over_Union():.4f})", flush=True)
