#!/usr/bin/env python3
"""Large-shape check of the split-bf16 convolution: run-to-run bitwise determinism and agreement with the fp32 kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "doubly-contrastive-semseg_amd")):
    sys.path.insert(0, p)
import torch
import dcs_amd.ops as ops
dev = "cuda:0"
torch.manual_seed(0)
SH = [(16, 256, 512, 64, 64, 3, 1), (16, 256, 512, 128, 128, 3, 1), (16, 128, 256, 128, 128, 3, 1), (16, 64, 128, 256, 256, 3, 1),
      (16, 32, 64, 512, 512, 3, 1), (16, 256, 512, 64, 128, 3, 2), (16, 256, 512, 64, 128, 1, 1), (16, 8, 16, 512, 512, 3, 1)]
for (N, H, W, Cin, Cout, k, s) in SH:
    x = torch.randn(N, H, W, Cin, device=dev)
    w = (torch.randn(Cout, Cin, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    pad = k // 2
    os.environ["DCS_CONV_X3"] = "1"
    a = ops.conv_fwd(x, w, s, pad, want_stats=True)
    b = ops.conv_fwd(x, w, s, pad, want_stats=True)
    os.environ["DCS_CONV_X3"] = "0"
    c = ops.conv_fwd(x, w, s, pad, want_stats=True)
    d = (a[0] - c[0]).abs()
    bad = (d > 1e-3 * c[0].abs().max()).nonzero()
    print((N, H, W, Cin, Cout, k, s), "fwd deterministic", torch.equal(a[0], b[0]), torch.equal(a[1], b[1]),
          "max|x3-f32| %.3e (scale %.2f)" % (float(d.max()), float(c[0].abs().max())), "bad", bad.shape[0],
          bad[:3].tolist() if bad.shape[0] else "")
    dy = torch.randn_like(c[0])
    wp = ops.pack_dgrad_weight(w)
    os.environ["DCS_CONV_X3"] = "1"
    a = ops.conv_dgrad(dy, wp, (H, W), s, pad)
    b = ops.conv_dgrad(dy, wp, (H, W), s, pad)
    os.environ["DCS_CONV_X3"] = "0"
    c = ops.conv_dgrad(dy, wp, (H, W), s, pad)
    d = (a - c).abs()
    bad = (d > 1e-3 * c.abs().max()).nonzero()
    print("      dgrad deterministic", torch.equal(a, b), "max|x3-f32| %.3e (scale %.2f)" % (float(d.max()), float(c.abs().max())),
          "bad", bad.shape[0], bad[:3].tolist() if bad.shape[0] else "")

print("--- data gradient with BatchNorm-backward sums")
os.environ["DCS_CONV_X3"] = "1"
for (N, H, W, Cin, Cout) in [(4, 128, 256, 128, 128), (4, 64, 128, 64, 64)]:
    y = torch.randn(N, H, W, Cin, device=dev)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(N, H, W, Cout, device=dev)
    wp = ops.pack_dgrad_weight(w)
    bn = ops.bn_finalize(ops.colsum(y.reshape(-1, Cin), moments=True), torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev),
                         torch.zeros(Cin, device=dev), torch.ones(Cin, device=dev), N * H * W, True)
    outs = []
    for rep in range(3):
        junk = torch.randn(1 << 20, device=dev)            # perturb the allocator between runs
        o, s = ops.conv_dgrad(dy, wp, (H, W), 1, 1, bnb=(y, None, bn, True))
        outs.append((o.clone(), s.clone()))
        del junk
    print((N, H, W, Cin, Cout), "out equal", [torch.equal(outs[0][0], o[0]) for o in outs[1:]], "sums equal",
          [torch.equal(outs[0][1], o[1]) for o in outs[1:]], "max diff", float((outs[0][0] - outs[1][0]).abs().max()))
    plain = ops.conv_dgrad(dy, wp, (H, W), 1, 1)
    print("   vs plain data gradient equal:", torch.equal(plain, outs[0][0]), float((plain - outs[0][0]).abs().max()))
