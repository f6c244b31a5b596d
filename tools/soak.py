#!/usr/bin/env python3
"""Soak run: N train steps of the C3 workload on one fixed synthetic batch (loss must fall, everything stays finite, the
allocator footprint must not grow).  usage: soak.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "doubly-contrastive-semseg_amd")):
    sys.path.insert(0, p)
import torch
import bench
from oracle import swiftnet_oracle as O
from dcs_amd.trainer import TrainStep, make_opts

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda", 0)
b = 16
left0, left1, labels, ldw, weather, cw = bench.device_batch(O, b, 1024, 2048, 0, True, dev)
torch.manual_seed(1)
ts = TrainStep(make_opts(criterion="supcon_pixelcontrast_focal", batch_size=b), class_weight=cw, device=dev)
t0 = time.time()
mem = []
for i in range(steps):
    s0 = dict(left=left0, label=labels.clone(), weather=weather, label_distance_weight=ldw)
    out = ts.step((s0, dict(left=left1)))
    if i % 10 == 0 or i == steps - 1:
        torch.cuda.synchronize()
        vals = {k: float(out[k]) for k in ("total", "supcon", "pixel", "seg")}
        mem.append(torch.cuda.memory_reserved() / 1e9)
        assert all(v == v and abs(v) < 1e6 for v in vals.values()), vals
        print(f"step {i:4d}  " + "  ".join(f"{k} {v:9.4f}" for k, v in vals.items()) +
              f"  reserved {mem[-1]:.1f} GB  elapsed {time.time() - t0:.1f} s", flush=True)
assert mem[-1] <= mem[1] * 1.02 + 0.5, mem
print("soak ok")
