#!/usr/bin/env python3
"""Signed-error probe of the convolution kernels against float64: mean signed relative error (a rounding BIAS shows up
here long before it shows in max / L2 errors) for the split-bf16 kernel and the exact-fp32 MFMA kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "doubly-contrastive-semseg_amd")):
    sys.path.insert(0, p)
import torch
import torch.nn.functional as F
import dcs_amd.ops as ops

dev = "cuda:0"
torch.manual_seed(0)
for positive in (True, False):
    for (N, H, W, Cin, Cout) in [(8, 64, 128, 64, 64), (4, 64, 128, 512, 128), (4, 64, 128, 2304, 128)]:
        x = torch.randn(N, H, W, Cin)
        w = (torch.randn(Cout, Cin, 3, 3) * 0.05)
        if positive:
            x, w = x.abs(), w.abs()
        w = w.contiguous(memory_format=torch.channels_last)
        ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, 1, 1).permute(0, 2, 3, 1)
        res = {}
        for mode in ("1", "0"):
            os.environ["DCS_CONV_X3"] = mode
            y = ops.conv_fwd(x.to(dev), w.to(dev).contiguous(memory_format=torch.channels_last), 1, 1).cpu().double()
            e = (y - ref)
            res[mode] = (float(e.mean() / ref.abs().mean()), float(e.abs().mean() / ref.abs().mean()),
                         float(e.sum() / e.abs().sum()))
        print(f"positive={positive} Cin={Cin}: x3 mean signed {res['1'][0]:+.3e} mean abs {res['1'][1]:.3e} "
              f"sign balance {res['1'][2]:+.3f} | fp32 mean signed {res['0'][0]:+.3e} mean abs {res['0'][1]:.3e} "
              f"sign balance {res['0'][2]:+.3f}")
