#!/usr/bin/env python3
"""Micro-benchmark of the HBM-bound elementwise / reduction kernels at the tensor sizes of one C3 train step
(SwiftNet-RN18, 32 crops of 1024x2048): achieved TB/s of algorithmic traffic (every tensor read or written once).

usage: elem_bench.py [--reps 10] [--out profiles/xxx.json]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "doubly-contrastive-semseg_amd")):
    sys.path.insert(0, p)
import torch


def timed(fn, reps):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import dcs_amd.ops as ops
    dev = torch.device("cuda", 0)
    rows = []
    for (N, H, W, C) in [(32, 256, 512, 64), (32, 128, 256, 128), (32, 64, 128, 256), (32, 256, 512, 128)]:
        n = N * H * W * C
        gb = n * 4 / 1e9
        y = torch.randn((N, H, W, C), device=dev)
        g = torch.randn_like(y)
        out = torch.relu(torch.randn_like(y))
        gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        bn = ops.bn_finalize(ops.colsum(y.reshape(-1, C), moments=True), gamma, beta, rm, rv, n // C, True)
        dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
        cases = {
            "colsum (1R)": (lambda: ops.colsum(y.reshape(-1, C), moments=True), 1),
            "bn_act (1R+1W)": (lambda: ops.bn_act(y, bn, relu=True), 2),
            "bn_act residual (2R+1W)": (lambda: ops.bn_act(y, bn, r=g, relu=True), 3),
            "bn_bwd relu (4R+1W)": (lambda: ops.bn_bwd(g, y, bn, gamma, relu=True, dgamma=dg, dbeta=db), 5),
            "bn_bwd masksrc+gm (6R+2W)": (lambda: ops.bn_bwd(g, y, bn, gamma, masksrc=out, want_gm=True, dgamma=dg, dbeta=db), 8),
            "axpy (2R+1W)": (lambda: ops.axpy(g, y, 0.5), 3),
        }
        for name, (fn, passes) in cases.items():
            ms = timed(fn, args.reps)
            rows.append(dict(shape=[N, H, W, C], op=name, ms=ms, tb_per_s=passes * gb / ms))
            print(f"{str((N, H, W, C)):24s} {name:28s} {ms:7.3f} ms  {passes * gb / ms:5.2f} TB/s", flush=True)
        del y, g, out
    # fused logits-upsample + loss (C3: 16 labelled images, logits 256x512x20 -> 1024x2048)
    N, ih, iw = 16, 256, 512
    lr = torch.randn((N, ih, iw, 20), device=dev)
    tgt = torch.randint(0, 19, (N, 4 * ih, 4 * iw), device=dev)
    tgt[:, :8] = 255
    ldw = torch.rand((N, 4 * ih, 4 * iw), device=dev)
    ldw[tgt == 255] = 0
    cw = torch.rand(19, device=dev) + 0.5
    alg = (lr.numel() * 4 * 2 + tgt.numel() * 8 + ldw.numel() * 4) / 1e9
    ms = timed(lambda: ops.seg_loss_fused(lr, 19, tgt.clone() if False else tgt, ldw, cw, "full"), args.reps)
    print(f"{'seg_loss_fused C3':24s} {'(low-res logits+labels+ldw)':28s} {ms:7.3f} ms  {alg / ms:5.2f} TB/s", flush=True)
    rows.append(dict(shape=[N, ih, iw, 20], op="seg_loss_fused", ms=ms, tb_per_s=alg / ms))
    up = ops.upsample_to_nchw(lr, 19, 4 * ih, 4 * iw)
    ms_u = timed(lambda: ops.upsample_to_nchw(lr, 19, 4 * ih, 4 * iw), args.reps)
    ms_l = timed(lambda: ops.seg_loss(up, tgt, ldw, cw, "full"), args.reps)
    g = torch.empty_like(up)
    ms_b = timed(lambda: ops.upsample_to_nchw_bwd(g, ih, iw, 20), args.reps)
    print(f"{'unfused chain':24s} upsample {ms_u:.3f} + loss {ms_l:.3f} + fold {ms_b:.3f} = {ms_u + ms_l + ms_b:.3f} ms (+ scale)", flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
