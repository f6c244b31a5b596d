#!/usr/bin/env python3
"""GPU probe: the ASPP projection (1x1, 1280 -> 256, five accumulating launches over the virtual concat) of the
deeplab_step_b2_128x256 fixture.  Captures the five sources the kernels saw, recomputes the projection in float64 on the CPU
from those very inputs, and reports the error of the per-channel MEAN of the kernel's output (what BatchNorm's running_mean
sees) for the split-bf16 and the exact-fp32 kernels."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "doubly-contrastive-semseg_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from oracle import deeplab_oracle as D, swiftnet_oracle as O  # noqa: E402
import dcs_amd.ops as ops  # noqa: E402
from test_deeplab_gpu import build  # noqa: E402


def run(x3):
    os.environ["DCS_CONV_X3"] = x3
    b = 2
    img, labels, ldw, weather, cw = O.synthetic_batch(b, 128, 256, seed=51, two_crops=True, cell=32)
    ts = build(b, cw)
    cap = []
    orig = ops.conv_fwd

    def spy(x, w, *a, **k):
        y = orig(x, w, *a, **k)
        if "koff" in k and w.shape[1] == 1280:
            cap.append((x.detach().clone(), w.detach().clone(), k["koff"], y))
        return y
    ops.conv_fwd = spy
    torch.manual_seed(321)
    with torch.no_grad():
        ts.model.train()
        ts.model([img[:b].to("cuda:0"), img[b:].to("cuda:0")], return_supcon_feature=True)
    ops.conv_fwd = orig
    torch.cuda.synchronize()
    yj = cap[-1][3].cpu().double()                                   # [B,h,w,256] after the 5th accumulation
    ref = torch.zeros_like(yj)
    for x, w, koff, _ in cap:
        w2 = w.cpu().double().reshape(256, 1280)[:, koff:koff + x.shape[-1]]
        ref += x.cpu().double() @ w2.t()
    err = yj - ref
    m_got, m_ref = yj.mean(dim=(0, 1, 2)), ref.mean(dim=(0, 1, 2))
    sd = ts.model.state_dict()
    return dict(x3=x3, n_launch=len(cap), max_rel=float(err.abs().max() / ref.abs().max()),
                signed_over_abs=float(err.sum() / err.abs().sum()),
                mean_norm_rel_err=float(abs(m_got.norm() - m_ref.norm()) / m_ref.norm()),
                mean_vec_rel_err=float((m_got - m_ref).norm() / m_ref.norm()),
                mean_abs=float(m_ref.abs().mean()), val_abs=float(ref.abs().mean()),
                rm_norm=float(sd["classifier.aspp.project.1.running_mean"].double().norm())), m_ref


if __name__ == "__main__":
    g64 = np.load(os.path.join(ROOT, "tests", "golden", "deeplab_step_b2_128x256.f64.npz"))
    names = [str(s) for s in g64["rs_names"]] if "rs_names" in g64.files else None
    g = np.load(os.path.join(ROOT, "tests", "golden", "deeplab_step_b2_128x256.npz"))
    i = [str(s) for s in g["rs_names"]].index("classifier.aspp.project.1.running_mean")
    print("reference running_mean norm fp32 / fp64:", float(g["rs_norms"][i]), float(g64["rs_norms"][i]))
    a, ma = run("1")
    b, mb = run("0")
    print(a); print(b)
    print("mean vector of exact projection, x3 inputs vs fp32 inputs:", float((ma - mb).norm() / mb.norm()))
