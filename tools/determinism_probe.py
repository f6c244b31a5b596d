#!/usr/bin/env python3
"""Two identical train steps -> which gradients differ bitwise?  usage: determinism_probe.py B H W"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "doubly-contrastive-semseg_amd")):
    sys.path.insert(0, p)
import torch
from oracle import swiftnet_oracle as O
from dcs_amd.trainer import TrainStep, make_opts
b, h, w = (int(a) for a in sys.argv[1:4])
dev = torch.device("cuda", 0)
img, labels, ldw, weather, cw = O.synthetic_batch(b, h, w, seed=11, two_crops=True, cell=64)
img, labels, ldw, weather = (t.to(dev) for t in (img, labels, ldw, weather))
runs = []
for _ in range(2):
    torch.manual_seed(5)
    ts = TrainStep(make_opts(criterion="supcon_pixelcontrast_focal", batch_size=b), class_weight=cw.to(dev), device=dev)
    s0 = dict(left=img[:b], label=labels.clone(), weather=weather, label_distance_weight=ldw)
    torch.manual_seed(77)
    out = ts.step((s0, dict(left=img[b:])), do_optimizer_step=False)
    torch.cuda.synchronize()
    runs.append(({k: p.grad.detach().clone() for k, p in ts.model.named_parameters() if p.grad is not None}, float(out["total"])))
    del ts, out
(g0, l0), (g1, l1) = runs
diff = [k for k in g0 if not torch.equal(g0[k], g1[k])]
print(f"B={b} {h}x{w} X3={os.environ.get('DCS_CONV_X3','1')} PRO={os.environ.get('DCS_PROLOGUE','1')} KSPLIT={os.environ.get('DCS_KSPLIT','1')}: "
      f"loss equal {l0 == l1}; {len(diff)} of {len(g0)} gradients differ", diff[-6:])
