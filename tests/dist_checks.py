"""Assertions shared by the CPU (emulated kernels) and GPU (real kernels) two-rank data-parallel tests."""
import os
import socket
import subprocess
import sys

import torch

from oracle import swiftnet_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def run_workers(out, device):
    port = free_port()
    env = dict(os.environ, DCS_DIST_DEVICE=device, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), str(r), "2", str(port), str(out)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(2)]
    logs = [p.communicate(timeout=900)[0].decode() for p in procs]
    for p, l in zip(procs, logs):
        assert p.returncode == 0, l[-3000:]
    return [torch.load(os.path.join(out, f"rank{r}.pt"), weights_only=True) for r in range(2)]


def single_process(criterion, seed, device, eval_bn=True, deeplab=False):
    """The same global batch in ONE process (no row gather, no all-reduce)."""
    sys.path.insert(0, HERE)
    from dist_worker import build, shard_sample
    B, h, w = (2, 96, 160) if deeplab else (2, 128, 256)
    batch = O.synthetic_batch(B, h, w, seed=seed, two_crops=True, cell=16 if deeplab else 32)
    ts = build(criterion, B, batch[4], device=device, deeplab=deeplab)
    if eval_bn:
        ts.model.eval()
    out = ts.step(shard_sample(batch, 0, B, True, B))
    return ts, out


def check_equals_single_process(r0, r1, prefix, ts, out, loss_rtol=1e-5, grad_rtol=2e-3):
    want = float(out["total"])
    assert abs(float(r0[prefix + "_total"]) - want) <= loss_rtol * abs(want), (float(r0[prefix + "_total"]), want)
    assert float(r0[prefix + "_total"]) == float(r1[prefix + "_total"])
    for k, p in ts.model.named_parameters():
        if p.grad is None:
            assert k not in r0[prefix + "_grads"]
            continue
        ref = p.grad.detach().cpu()
        for r in (r0, r1):
            err = float((r[prefix + "_grads"][k] - ref).norm() / ref.norm().clamp_min(1e-20))
            assert err < grad_rtol, (k, err)
        assert torch.equal(r0[prefix + "_grads"][k], r1[prefix + "_grads"][k]), k       # all-reduce leaves identical bits


def check_training_mode(r0, r1, prefix="B", images_per_rank=1):
    P = prefix + "_"
    for k in ("total", "supcon", "pixel", "seg"):
        assert float(r0[P + k]) == float(r1[P + k]), k
    assert torch.equal(r0[P + "param_checksum"], r1[P + "param_checksum"])
    X, y = r0[P + "pixel_rows"], r0[P + "pixel_labels"]
    assert torch.equal(X, r1[P + "pixel_rows"])
    assert X.shape[0] == int(r0[P + "local_anchor_count"]) + int(r1[P + "local_anchor_count"])
    assert int(r0[P + "local_anchor_count"]) > 0 and int(r1[P + "local_anchor_count"]) > 0
    # fixed-shape gather: 2 ranks x cap rows (cap = max_views * 19 classes * 1 image per rank), padding rows labelled -1
    assert int(r0[P + "gathered_rows"]) == 2 * 38 * images_per_rank
    want = O.pixel_contrastive(X.unsqueeze(1), y)
    assert abs(float(want) - float(r0[P + "pixel"])) < 1e-5 * abs(float(want))


def check_empty_rank(r0, r1):
    """Scenario D: rank 1 sampled nothing."""
    assert int(r1["D_local_anchor_count"]) == 0 and int(r0["D_local_anchor_count"]) > 0
    for k in ("D_pixel", "D_seg", "D_total"):
        assert float(r0[k]) == float(r1[k]), k
    assert torch.equal(r0["D_param_checksum"], r1["D_param_checksum"])
    X, y = r0["D_pixel_rows"], r0["D_pixel_labels"]
    assert X.shape[0] == int(r0["D_local_anchor_count"]) and torch.equal(X, r1["D_pixel_rows"])
    want = O.pixel_contrastive(X.unsqueeze(1), y)
    assert abs(float(want) - float(r0["D_pixel"])) < 1e-5 * abs(float(want))
