"""Host time of the sampler's plan at the C3 shape (16 images x 19 classes over 131072 labelled low-resolution pixels
each): the library (dcs_sampler_plan) against the statement in torch (torch.randperm per kept class)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "doubly-contrastive-semseg_amd"))
from dcs_amd import losses

rng = np.random.default_rng(0)
cnt = torch.zeros((16, 19, 2), dtype=torch.int64)
for b in range(16):
    p = rng.dirichlet(np.ones(19) * 0.5) * 131072
    for c in range(19):
        h = int(p[c] * rng.uniform(0.05, 0.5))
        cnt[b, c, 0] = h
        cnt[b, c, 1] = int(p[c]) - h
for mode in ("lib", "torch"):
    os.environ["DCS_SAMPLER_PLAN"] = mode
    torch.manual_seed(0)
    losses.plan_anchor_requests(cnt, 19, 1024, 2)
    t = time.perf_counter()
    n = 20 if mode == "lib" else 5
    for _ in range(n):
        losses.plan_anchor_requests(cnt, 19, 1024, 2)
    print(f"{mode}: {(time.perf_counter() - t) / n * 1e3:.2f} ms per plan", flush=True)
