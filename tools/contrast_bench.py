#!/usr/bin/env python3
"""Timing of the fused similarity / InfoNCE loss (dcs_contrast_fused) at the per-rank and the gathered global size:
eager (host launch overhead included) and as a HIP-graph replay (device time only).
usage: contrast_bench.py [--sizes 608,4864] [--reps 50]   (run under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "doubly-contrastive-semseg_amd")):
    sys.path.insert(0, p)
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="608,4864")
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--mode", type=int, default=0)
    args = ap.parse_args()
    import dcs_amd.ops as ops
    dev = torch.device("cuda", 0)
    for A in [int(v) for v in args.sizes.split(",")]:
        gen = torch.Generator(device="cpu").manual_seed(A)
        X = torch.nn.functional.normalize(torch.randn(A, 128, generator=gen), dim=1).to(dev)
        y = torch.randint(0, 19, (A,), generator=gen).float().to(dev)
        for _ in range(3):
            ops.contrast_fwd_bwd(X, y, args.mode, 0.07)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            ops.contrast_fwd_bwd(X, y, args.mode, 0.07)
        e1.record()
        torch.cuda.synchronize()
        eager = e0.elapsed_time(e1) / args.reps * 1e3
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            ops.contrast_fwd_bwd(X, y, args.mode, 0.07)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                loss, dX = ops.contrast_fwd_bwd(X, y, args.mode, 0.07)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        graph = e0.elapsed_time(e1) / args.reps * 1e3
        fl = 6.0 * A * A * 128
        print(f"A={A:5d} mode {args.mode}: eager {eager:8.1f} us ({fl / eager / 1e6:6.1f} TF)   graph replay {graph:8.1f} us "
              f"({fl / graph / 1e6:6.1f} TF)", flush=True)


if __name__ == "__main__":
    main()
