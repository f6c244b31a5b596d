"""CPU: the data-parallel exchange (dcs_amd/dist.py) with world_size 2 over gloo, kernels emulated.
A: with eval-mode BatchNorm the model is per-image independent, so the DP step (sharded batch, all-gathered
   SupCon rows, globally normalised seg loss, summed gradients) must reproduce the single-process step.
B: training mode with per-rank anchor sampling: ranks agree on the global losses and end with identical
   parameters; the global pixel loss equals the oracle's loss on the union of both ranks' anchors.
C: SimCLR image contrast (class_labels=None): instance ids stay unique across ranks -- DP == single process.
D: one rank's shard is all "ignore": it joins the collectives with padding only (no hang), zero pixel gradient.
E/F: BASELINE config 5's model (DeepLabV3+ / ResNet-101) under the same wrapper: E as A, F as B with the lazy 2048-channel
   fine_feat0 (rows interpolated on demand before the gather)."""
import pytest
import torch

import emu_ops
from dist_checks import (check_empty_rank, check_equals_single_process, check_training_mode, run_workers,
                         single_process)


@pytest.fixture(scope="module")
def dp_results(tmp_path_factory):
    return run_workers(tmp_path_factory.mktemp("dp"), "cpu")


def test_dp_equals_single_process_with_eval_batchnorm(dp_results, monkeypatch):
    emu_ops.install(monkeypatch)
    ts, out = single_process("supcon_focal", 41, "cpu")
    check_equals_single_process(dp_results[0], dp_results[1], "A", ts, out, loss_rtol=1e-4)
    for k, p in ts.model.named_parameters():
        assert torch.equal(dp_results[0]["A_params"][k], dp_results[1]["A_params"][k]), k


def test_dp_training_mode_global_losses_and_sync(dp_results):
    check_training_mode(*dp_results)


def test_dp_simclr_instance_labels_are_unique_across_ranks(dp_results, monkeypatch):
    emu_ops.install(monkeypatch)
    r0, r1 = dp_results
    lab = r0["C_labels"]
    assert lab.numel() == 4 and len(set(lab.tolist())) == 2 and torch.equal(lab, r1["C_labels"])     # 2 images x 2 views
    ts, out = single_process("supcon_simclr_focal", 45, "cpu")
    assert abs(float(r0["C_simclr"]) - float(out["simclr"])) <= 1e-5 * abs(float(out["simclr"]))
    check_equals_single_process(r0, r1, "C", ts, out, loss_rtol=1e-4)
    # every rank back-propagates only its own rows: the projection-head gradients of the ranks sum to the global one
    summed = [a + b for a, b in zip(r0["C_proj_grads"], r1["C_proj_grads"])]
    for s, p in zip(summed, ts.supcon_criterion.projection.parameters()):
        ref = p.grad.detach()
        assert float((s - ref).norm() / ref.norm().clamp_min(1e-20)) < 2e-3


def test_dp_rank_without_anchors_joins_the_collectives(dp_results):
    check_empty_rank(*dp_results)


def test_deeplab_dp_equals_single_process_with_eval_batchnorm(dp_results, monkeypatch):
    emu_ops.install(monkeypatch)
    ts, out = single_process("supcon_focal", 49, "cpu", deeplab=True)
    check_equals_single_process(dp_results[0], dp_results[1], "E", ts, out, loss_rtol=1e-4)


def test_deeplab_dp_training_mode_global_losses_and_sync(dp_results):
    check_training_mode(*dp_results, prefix="F", images_per_rank=2)
    assert dp_results[0]["F_pixel_rows"].shape[1] == 2048
