#!/usr/bin/env python3
"""conv3x3_x3_kernel (halo in LDS) against the per-tap split-bf16 kernel and float64: run with DCS_X3_HALO=2 / 0."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "doubly-contrastive-semseg_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import torch.nn.functional as F
import dcs_amd.ops as ops
dev = "cuda:0"
torch.manual_seed(0)
out = {}
for (N, H, W, Cin, Cout) in [(2, 16, 64, 64, 64), (1, 8, 32, 64, 64), (2, 8, 96, 128, 128), (1, 4, 32, 256, 128), (2, 12, 64, 128, 256),
                             (3, 24, 32, 48, 64), (2, 16, 64, 64, 48)]:
    x = torch.randn(N, H, W, Cin)
    w = (torch.randn(Cout, Cin, 3, 3) * 0.05).contiguous(memory_format=torch.channels_last)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, 1, 1).permute(0, 2, 3, 1)
    xd, wd = x.to(dev), w.to(dev).contiguous(memory_format=torch.channels_last)
    y, st = ops.conv_fwd(xd, wd, 1, 1, want_stats=True)
    y2, st2 = ops.conv_fwd(xd, wd, 1, 1, want_stats=True)
    e = y.cpu().double() - ref
    gam, bet = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1
    bn = ops.bn_finalize(ops.colsum(xd.reshape(-1, Cin), moments=True), gam, bet, torch.zeros(Cin, device=dev), torch.ones(Cin, device=dev),
                         N * H * W, True)
    z = ops.bn_act(xd, bn, relu=True)
    yp = ops.conv_fwd(xd, wd, 1, 1, pro=bn)
    yz = ops.conv_fwd(z, wd, 1, 1)
    dy = torch.randn(N, H, W, Cout, device=dev)
    wp = ops.pack_dgrad_weight(wd)
    refd = torch.nn.grad.conv2d_input((N, Cin, H, W), w.double(), dy.cpu().permute(0, 3, 1, 2).double(), 1, 1).permute(0, 2, 3, 1)
    base = torch.randn(N, H, W, Cin, device=dev)
    d, sums = ops.conv_dgrad(dy, wp, (H, W), 1, 1, out=base.clone(), accumulate=True, bnb=(xd, None, bn, True))
    ed = (d - base).cpu().double() - refd
    print((N, H, W, Cin, Cout), "fwd max-rel %.2e bal %+.3f det %s stats-eq %s | pro==materialised %s | dgrad max-rel %.2e sums %s"
          % (float(e.abs().max() / ref.abs().max()), float(e.sum() / e.abs().sum()), torch.equal(y, y2), torch.equal(st, st2),
             torch.equal(yp, yz), float(ed.abs().max() / refd.abs().max()), None if sums is None else [round(float(v), 3) for v in sums[0, :2]]))
    out[(N, H, W, Cin, Cout)] = (y.cpu(), d.cpu(), None if sums is None else sums.cpu())
torch.save(out, os.environ.get("OUT", "/tmp/x3_halo.pt"))
