"""CPU: the data-parallel exchange (dcs_amd/dist.py) with world_size 2 over gloo, kernels emulated.
A: with eval-mode BatchNorm the model is per-image independent, so the DP step (sharded batch, all-gathered
   SupCon rows, globally normalised seg loss, summed gradients) must reproduce the single-process step.
B: training mode with per-rank anchor sampling: ranks agree on the global losses and end with identical
   parameters; the global pixel loss equals the oracle's loss on the union of both ranks' anchors."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import emu_ops
from oracle import swiftnet_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.fixture(scope="module")
def dp_results(tmp_path_factory):
    out = tmp_path_factory.mktemp("dp")
    port = free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), str(r), "2", str(port), str(out)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    logs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, l in zip(procs, logs):
        assert p.returncode == 0, l[-3000:]
    return [torch.load(os.path.join(out, f"rank{r}.pt"), weights_only=True) for r in range(2)]


def test_dp_equals_single_process_with_eval_batchnorm(dp_results, monkeypatch):
    emu_ops.install(monkeypatch)
    sys.path.insert(0, HERE)
    from dist_worker import build, shard_sample
    B, h, w = 2, 128, 256
    batch = O.synthetic_batch(B, h, w, seed=41, two_crops=True, cell=32)
    ts = build("supcon_focal", B, batch[4])
    ts.model.eval()
    out = ts.step(shard_sample(batch, 0, B, True, B))
    r0, r1 = dp_results
    assert abs(float(r0["A_total"]) - float(out["total"])) < 1e-4 * abs(float(out["total"]))
    assert float(r0["A_total"]) == float(r1["A_total"])
    for k, p in ts.model.named_parameters():
        if p.grad is None:
            assert k not in r0["A_grads"]
            continue
        ref = p.grad.detach()
        for r in (r0, r1):
            err = float((r["A_grads"][k] - ref).norm() / ref.norm().clamp_min(1e-20))
            assert err < 2e-3, (k, err)
        assert torch.equal(r0["A_grads"][k], r1["A_grads"][k]), k       # all-reduce leaves identical bits
        assert torch.equal(r0["A_params"][k], r1["A_params"][k]), k


def test_dp_training_mode_global_losses_and_sync(dp_results):
    r0, r1 = dp_results
    for k in ("B_total", "B_supcon", "B_pixel", "B_seg"):
        assert float(r0[k]) == float(r1[k]), k
    assert torch.equal(r0["B_param_checksum"], r1["B_param_checksum"])
    X, y = r0["B_pixel_rows"], r0["B_pixel_labels"]
    assert torch.equal(X, r1["B_pixel_rows"])
    assert X.shape[0] == int(r0["B_local_anchor_count"]) + int(r1["B_local_anchor_count"])
    want = O.pixel_contrastive(X.unsqueeze(1), y)
    assert abs(float(want) - float(r0["B_pixel"])) < 1e-5 * abs(float(want))
