#!/usr/bin/env python3
"""Per-tile BatchNorm-backward partial sums of a data-gradient launch (dcs_conv_gather_x3 / dcs_conv_gather_bnbwd through
ops._gather_launch) against sums recomputed from the written gradient: determinism, unwritten rows, wrong entries.  This is
the probe that isolated the SLP-vectorised epilogue defect described in csrc/Makefile (5 % of the odd-channel entries wrong
and different from run to run on the 128-wide split-bf16 tile; exact with -fno-slp-vectorize)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "doubly-contrastive-semseg_amd")):
    sys.path.insert(0, p)
import torch
import dcs_amd.ops as ops
from dcs_amd.ops import _p, _call, _stream
dev = "cuda:0"
torch.manual_seed(0)
N, H, W, Cin, Cout = 4, 128, 256, 128, 128
y = torch.randn(N, H, W, Cin, device=dev)
w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
dy = torch.randn(N, H, W, Cout, device=dev)
wp = ops.pack_dgrad_weight(w)
bn = ops.bn_finalize(ops.colsum(y.reshape(-1, Cin), moments=True), torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev),
                     torch.zeros(Cin, device=dev), torch.ones(Cin, device=dev), N * H * W, True)
g = ops.geoms_dgrad(N, H, W, Cin, Cout, 3, 3, 1, 1)[0]
G = N * H * W // 128
for x3 in ("1", "0"):
    os.environ["DCS_CONV_X3"] = x3
    for masked in (False, True):
        mask = torch.relu(torch.randn(N, H, W, Cin, device=dev)) if masked else None
        parts = []
        for rep in range(2):
            part = torch.full((G, 2, Cin), 7777.0, device=dev)
            out = torch.empty(N, H, W, Cin, device=dev)
            ops._gather_launch(dy, wp, None, out, g, 0, part, None, (y, mask, bn, not masked))
            torch.cuda.synchronize()
            parts.append(part.clone())
        m = (mask > 0) if masked else ((y * bn[0] + bn[1]) > 0)
        gm = (out * m).reshape(G, 128, Cin)
        ref = gm.double().sum(1)
        e = (parts[0][:, 0].double() - ref).abs()
        bad = (e > 1e-2).nonzero()
        print(f"x3={x3} masked={masked}: part deterministic {torch.equal(parts[0], parts[1])}; unwritten {int((parts[0] == 7777.0).sum())}; "
              f"rows x channels wrong {bad.shape[0]} of {G * Cin}; first {bad[:6].tolist()}; max err {float(e.max()):.3e}")
