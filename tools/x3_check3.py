#!/usr/bin/env python3
"""Bisect helper for the SLP epilogue defect: the per-tile BatchNorm-backward partial sums of one data-gradient launch
(conv_gather_x3_kernel<128,128>), twice, saved to gpurun_out/slp_parts_<tag>.pt; with `compare <tagA> <tagB>` prints where
two builds / two runs differ (tile row, which sum, channel, magnitude)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "doubly-contrastive-semseg_amd")):
    sys.path.insert(0, p)
import torch
OUT = os.path.join(ROOT, "gpurun_out")
if sys.argv[1] == "compare":
    a, b = (torch.load(os.path.join(OUT, f"slp_parts_{t}.pt")) for t in sys.argv[2:4])
    for name, x, y in (("run0 vs run1 of " + sys.argv[2], a[0], a[1]), (sys.argv[2] + " vs " + sys.argv[3], a[0], b[0])):
        d = (x != y)
        idx = d.nonzero()
        print(f"{name}: {int(d.sum())} of {d.numel()} entries differ; by sum index {[int(d[:, s].sum()) for s in range(2)]}; "
              f"by channel parity {[int(d[:, :, c::2].sum()) for c in range(2)]}; by channel mod 4 {[int(d[:, :, c::4].sum()) for c in range(4)]}")
        if idx.numel():
            rows = idx[:, 0].unique()
            print("   tile rows affected:", rows.numel(), "first", rows[:12].tolist(), " channels first", idx[:12, 2].tolist())
            mag = (x.double() - y.double()).abs()[d]
            print("   |diff| min/median/max", float(mag.min()), float(mag.median()), float(mag.max()), " |value| median", float(x.abs().median()))
    sys.exit(0)
tag = sys.argv[1]
import dcs_amd.ops as ops
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
parts = []
for rep in range(2):
    part = torch.full((G, 2, Cin), 7777.0, device=dev)
    out = torch.empty(N, H, W, Cin, device=dev)
    ops._gather_launch(dy, wp, None, out, g, 0, part, None, (y, None, bn, True))
    torch.cuda.synchronize()
    parts.append(part.cpu())
if len(sys.argv) > 2 and sys.argv[2] == "explain":
    # which single-row term explains a wrong entry?  gm[r, c] = out * relu mask, rows r of the 128-pixel tile
    m = ((y * bn[0] + bn[1]) > 0)
    gm = (out * m).reshape(G, 128, Cin).double().cpu()
    ref = gm.sum(1)
    got = parts[0][:, 0].double()
    bad = ((got - ref).abs() > 1e-3).nonzero()
    print("wrong entries", bad.shape[0])
    import collections
    hist = collections.Counter()
    for (gi, c) in bad[:400].tolist():
        d = float(got[gi, c] - ref[gi, c])
        col, nb = gm[gi, :, c], gm[gi, :, c - 1]
        cands = {"-gm[r,c] (row missing)": -col, "+gm[r,c] (row twice)": col, "gm[r,c-1]-gm[r,c] (even lane's value used)": nb - col,
                 "+gm[r,c-1]": nb}
        best = min(((float((v - d).abs().min()), k, int((v - d).abs().argmin())) for k, v in cands.items()))
        hist[(best[1] if best[0] < 1e-4 else "unexplained", best[2] if best[0] < 1e-4 else -1)] += 1
    for k, v in sorted(hist.items(), key=lambda kv: -kv[1])[:20]:
        print("  ", v, k)
os.makedirs(OUT, exist_ok=True)
torch.save(parts, os.path.join(OUT, f"slp_parts_{tag}.pt"))
print(tag, "saved; deterministic", torch.equal(parts[0], parts[1]))
