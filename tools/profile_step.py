#!/usr/bin/env python3
"""Coarse wall-clock breakdown of one train step (synchronising between phases): where host time goes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "doubly-contrastive-semseg_amd")):
    sys.path.insert(0, p)
import torch
from bench import device_batch
from dcs_amd.trainer import TrainStep, make_opts
from dcs_amd import losses
from oracle import swiftnet_oracle as O

b = int(sys.argv[1]) if len(sys.argv) > 1 else 16
h, w = 1024, 2048
dev = torch.device("cuda", 0)
left0, left1, labels, ldw, weather, cw = device_batch(O, b, h, w, 0, True, dev)
ts = TrainStep(make_opts(criterion="supcon_pixelcontrast_focal", batch_size=b), class_weight=cw, device=dev)

def sync():
    torch.cuda.synchronize(); return time.perf_counter()

orig_plan = losses.plan_anchor_requests
plan_t = [0.0]
def timed_plan(*a, **k):
    t0 = time.perf_counter(); r = orig_plan(*a, **k); plan_t[0] += time.perf_counter() - t0; return r
losses.plan_anchor_requests = timed_plan

for it in range(3):
    lab = labels.clone()
    t0 = sync()
    left = torch.cat([left0, left1], 0)
    t1 = sync()
    seg, before, ff, ff0 = ts.model(left, return_supcon_feature=True)
    t2h = time.perf_counter(); t2 = sync()
    sup = ts.supcon_criterion(ff, class_labels=weather)
    t3 = sync()
    plan_t[0] = 0.0
    pix = ts.pixelcontrast_criterion(ff0, labels=lab, predict=before)
    t4 = sync()
    sl = ts.criterion(seg, lab, {"label_distance_weight": ldw})
    t5 = sync()
    total = (sup + pix) / b + 1.2 * sl
    ts.optimizer.zero_grad()
    total.backward()
    t6h = time.perf_counter(); t6 = sync()
    ts.optimizer.step()
    t7 = sync()
    print(f"it{it}: cat {1e3*(t1-t0):.1f} | fwd {1e3*(t2-t1):.1f} (host enqueue {1e3*(t2h-t1):.1f}) | supcon {1e3*(t3-t2):.1f} | "
          f"pixel {1e3*(t4-t3):.1f} (host plan {1e3*plan_t[0]:.1f}) | seg {1e3*(t5-t4):.1f} | bwd {1e3*(t6-t5):.1f} "
          f"(host enqueue {1e3*(t6h-t5):.1f}) | adam {1e3*(t7-t6):.1f} | total {1e3*(t7-t0):.1f}")
