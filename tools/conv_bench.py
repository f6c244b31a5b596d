#!/usr/bin/env python3
"""Micro-benchmark of the conv kernels on the C3 layer shapes (HIP events, random data).
usage: conv_bench.py [fwd|dgrad|wgrad|all] [reps] [shape-filter]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "doubly-contrastive-semseg_amd")):
    sys.path.insert(0, p)
import torch
import dcs_amd.ops as ops

which = sys.argv[1] if len(sys.argv) > 1 else "all"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
filt = sys.argv[3] if len(sys.argv) > 3 else ""
dev = "cuda:0"
# (name, N, H, W, Cin, Cout, k, stride)
SHAPES = [
    ("l1_64_256x512", 32, 256, 512, 64, 64, 3, 1),
    ("dec_128_256x512", 32, 256, 512, 128, 128, 3, 1),
    ("l2_128_128x256", 32, 128, 256, 128, 128, 3, 1),
    ("l3_256_64x128", 32, 64, 128, 256, 256, 3, 1),
    ("l4_512_32x64", 32, 32, 64, 512, 512, 3, 1),
    ("l2s2_64_128", 32, 256, 512, 64, 128, 3, 2),
    ("skip_64_128_1x1", 32, 256, 512, 64, 128, 1, 1),
    ("sw_128_128_1x1", 32, 128, 256, 128, 128, 1, 1),
    ("dl_256_1024_1x1", 8, 64, 128, 256, 1024, 1, 1),
    ("dl_1024_256_1x1", 8, 64, 128, 1024, 256, 1, 1),
    ("dl_64_256_1x1", 8, 256, 512, 64, 256, 1, 1),
    ("dl_128_512_1x1", 8, 128, 256, 128, 512, 1, 1),
]


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, N, H, W, Cin, Cout, k, s in SHAPES:
    if filt and not any(f in name for f in filt.split(',')):
        continue
    x = torch.randn(N, H, W, Cin, device=dev)
    w = (torch.randn(Cout, Cin, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    pad = k // 2
    OH, OW = ops.out_size(H, k, s, pad), ops.out_size(W, k, s, pad)
    flops = 2.0 * N * OH * OW * k * k * Cin * Cout
    dy = torch.randn(N, OH, OW, Cout, device=dev)
    ops.new_step(True)
    if ops.x2h_on():
        ops.tag_max(dy)          # the fp16 two-piece data / weight gradient scales by the tensor's maximum
    wp = ops.pack_dgrad_weight(w)
    dw = torch.empty_like(w)
    out = []
    if which in ("fwd", "all"):
        ms = timeit(lambda: ops.conv_fwd(x, w, s, pad)); out.append(f"fwd {ms:7.3f} ms {flops/ms/1e9:6.1f} TF")
    if which in ("fwdstats", "all"):
        ms = timeit(lambda: ops.conv_fwd(x, w, s, pad, want_stats=True)); out.append(f"fwd+stats {ms:7.3f} ms {flops/ms/1e9:6.1f} TF")
    if which in ("dgrad", "all"):
        ms = timeit(lambda: ops.conv_dgrad(dy, wp, (H, W), s, pad)); out.append(f"dgrad {ms:7.3f} ms {flops/ms/1e9:6.1f} TF")
    if which in ("wgrad", "all"):
        ms = timeit(lambda: ops.conv_wgrad(x, dy, dw, s, pad, False)); out.append(f"wgrad {ms:7.3f} ms {flops/ms/1e9:6.1f} TF")
    print(f"{name:18s} " + " | ".join(out), flush=True)
