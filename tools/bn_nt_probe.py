"""A/B of streaming (non-temporal) accesses in bn_bwd_apply on the BatchNorm shapes of C3 (library option bn_nt = 0 / 1)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "doubly-contrastive-semseg_amd"))
from dcs_amd import lib, ops
dev = "cuda:0"
for (N, H, W, C) in [(32, 256, 512, 64), (32, 128, 256, 128), (32, 64, 128, 256), (32, 256, 512, 128)]:
    g = torch.randn(N, H, W, C, device=dev); y = torch.randn(N, H, W, C, device=dev); out = torch.randn(N, H, W, C, device=dev)
    bn = torch.stack([torch.ones(C), torch.zeros(C), torch.zeros(C), torch.ones(C)]).to(dev).contiguous()
    gamma = torch.ones(C, device=dev); sums = torch.zeros(2, C, device=dev)
    for mode in ("0", "1", "0", "1"):
        lib.set_option("bn_nt", int(mode))
        for variant in ("bn2", "bn1"):
            kw = dict(masksrc=out, want_gm=True) if variant == "bn2" else dict(relu=True)
            for _ in range(3):
                ops.bn_bwd(g, y, bn, gamma, sums=sums, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.bn_bwd(g, y, bn, gamma, sums=sums, **kw)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            passes = 5 if variant == "bn2" else 3
            print(f"{N}x{H}x{W}x{C} {variant} NT={mode}: {ms*1e3:8.1f} us  {passes * g.numel() * 4 / ms / 1e9:6.2f} TB/s", flush=True)
