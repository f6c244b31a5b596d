"""GPU: what ONE GPU allows of the multi-rank path.
  * two ranks sharing cuda:0 over GLOO (a single card cannot host two RCCL ranks): the bench rehearsal and the numeric
    DP checks on the real kernels (SwiftNet and DeepLabV3+);
  * the REAL backend (RCCL, "nccl") at world size 1 (tests/rccl_worker.py): librccl loads, every collective of
    dcs_amd/dist.py runs on device memory, DP step == plain step for both models.
No RCCL run with N > 1 exists in this repository: that is the driver's 8-GPU scaling bench."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_rehearsal():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, DCS_DIST_BACKEND="gloo", DCS_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "2",
           "--height", "256", "--width", "512", "--steps", "2", "--warmup", "1"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 4 and rec["scaling"] == "weak"
    assert rec["value"] > 0 and rec["roofline"]["bound"] == "mfma" and "cpu_baseline" not in rec


@pytest.fixture(scope="module")
def dp_gpu_results(tmp_path_factory):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from dist_checks import run_workers
    return run_workers(tmp_path_factory.mktemp("dp_gpu"), "cuda")


def test_two_rank_dp_step_on_real_kernels_equals_single_process(dp_gpu_results):
    """C4's data-parallel step on the REAL kernels (two gloo ranks on one GPU): with eval-mode BatchNorm the DP step on
    two half-batches equals the single-process step on the full batch (loss 1e-5, gradients 2e-3 relative L2), the two
    ranks hold bit-identical gradients and parameters after the all-reduce + Adam."""
    from dist_checks import check_equals_single_process, single_process
    r0, r1 = dp_gpu_results
    ts, out = single_process("supcon_focal", 41, "cuda:0")
    check_equals_single_process(r0, r1, "A", ts, out, loss_rtol=1e-5, grad_rtol=2e-3)
    for k, _ in ts.model.named_parameters():
        assert torch.equal(r0["A_params"][k], r1["A_params"][k]), k
    ts, out = single_process("supcon_simclr_focal", 45, "cuda:0")           # instance ids unique across ranks
    assert abs(float(r0["C_simclr"]) - float(out["simclr"].detach())) <= 1e-5 * abs(float(out["simclr"].detach()))
    check_equals_single_process(r0, r1, "C", ts, out, loss_rtol=1e-5, grad_rtol=2e-3)


def test_two_rank_dp_global_pixel_loss_on_real_kernels(dp_gpu_results):
    """Training mode: per-rank sampling, fixed-shape gather with -1 padding, global denominators: the ranks agree bit for
    bit and the global pixel loss equals the oracle on the union of both ranks' anchors; a rank whose shard holds no
    class joins the collectives with padding only."""
    from dist_checks import check_empty_rank, check_training_mode
    check_training_mode(*dp_gpu_results)
    check_empty_rank(*dp_gpu_results)


def test_two_rank_deeplab_dp_on_real_kernels(dp_gpu_results):
    """BASELINE config 5's model under DataParallelStep on the real kernels (two gloo ranks on one GPU): eval-mode
    BatchNorm => equals the single-process step; training mode with the lazy 2048-channel fine_feat0 => ranks agree bit
    for bit, global pixel loss equals the oracle on the union of the 2048-wide anchor rows."""
    from dist_checks import check_equals_single_process, check_training_mode, single_process
    r0, r1 = dp_gpu_results
    ts, out = single_process("supcon_focal", 49, "cuda:0", deeplab=True)
    check_equals_single_process(r0, r1, "E", ts, out, loss_rtol=1e-5, grad_rtol=2e-3)
    check_training_mode(r0, r1, prefix="F", images_per_rank=2)


def test_rccl_backend_world_size_one():
    """The production backend itself: one rank, backend "nccl" (= RCCL), launched like the driver launches bench.py."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "tests", "rccl_worker.py")]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    line = [l for l in p.stdout.decode().splitlines() if l.startswith("RCCL_WORKER ")][0]
    rec = json.loads(line[len("RCCL_WORKER "):])
    out = os.path.join(ROOT, "gpurun_out", "parity")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "rccl_world1.json"), "w") as f:
        json.dump(rec, f, indent=1)
    assert rec["backend"] == "nccl" and rec["world"] == 1
    assert any("rccl" in l for l in rec["mapped_libraries"]) and any("libdcs_hip" in l for l in rec["mapped_libraries"])
    assert rec["row_gather_ok"] and rec["flat_allreduce_ok"]
    assert abs(rec["seg_reduce"][0] - 2.5) < 1e-6 and rec["seg_reduce"][1] == 1000.0
    for st in rec["steps"]:
        # bitwise the plain step: padding rows of the fixed-shape gather take part in nothing, x * N * (1 / N) of the
        # seg-loss renormalisation and the world-1 sum all-reduce happen to be exact here (measured: 87 / 338 gradient
        # tensors bitwise); kernels are deterministic, so this does not flake
        assert st["same_anchors"], st
        assert st["loss_rel_err"] == 0.0 and st["grads_bitwise"] == st["grad_tensors"] and st["param_rel_l2_max"] == 0.0, st
