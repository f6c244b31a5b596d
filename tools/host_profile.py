#!/usr/bin/env python3
"""Host-side profile (cProfile) of the train step at a small configuration, where the step is bounded by Python / launch
overhead rather than by the device.  usage: host_profile.py [batch height width criterion steps]"""
import cProfile, pstats, os, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "doubly-contrastive-semseg_amd")):
    sys.path.insert(0, p)
import torch
import bench
from oracle import swiftnet_oracle as O
from dcs_amd.trainer import TrainStep, make_opts

b, h, w = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (8, 512, 1024)
crit = sys.argv[4] if len(sys.argv) > 4 else "pixelcontrast_focal"
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
dev = torch.device("cuda", 0)
two = "supcon" in crit
left0, left1, labels, ldw, weather, cw = bench.device_batch(O, b, h, w, 0, two, dev)
torch.manual_seed(1)
ts = TrainStep(make_opts(criterion=crit, batch_size=b), class_weight=cw, device=dev)


def one():
    s0 = dict(left=left0, label=labels.clone(), weather=weather, label_distance_weight=ldw)
    return ts.step((s0, dict(left=left1)) if two else s0)


for _ in range(3):
    one()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    one()
torch.cuda.synchronize()
pr.disable()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(28)
print(st.getvalue()[:6000])
