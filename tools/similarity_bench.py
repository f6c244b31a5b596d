#!/usr/bin/env python3
"""The embedding-similarity kernels of the pixel-contrastive loss in isolation (S = X X^T, dX = (G + G^T) X) for
A anchors of dimension 128; run under `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`
for the MFMA utilisation of the similarity matmul.  usage: similarity_bench.py [A ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "doubly-contrastive-semseg_amd")):
    sys.path.insert(0, p)
import torch
import dcs_amd.ops as ops

for A in [int(a) for a in sys.argv[1:]] or [608, 4864]:
    g = torch.Generator().manual_seed(A)
    X = torch.nn.functional.normalize(torch.randn(A, 128, generator=g), dim=1).cuda()
    y = torch.randint(0, 19, (A,), generator=g).float().cuda()
    for _ in range(3):
        ops.linear(X, X)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.linear(X, X)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"A={A}: S = X X^T {us:8.1f} us  {2.0 * A * A * 128 / us / 1e6:6.1f} TFLOP/s", flush=True)
    for _ in range(3):
        ops.contrast_fwd_bwd(X, y, 0, 0.07)
    torch.cuda.synchronize()
