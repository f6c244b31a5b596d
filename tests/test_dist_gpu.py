"""GPU: rehearsal of the multi-rank bench path on ONE GPU (two ranks share cuda:0, gloo backend; the real job uses
RCCL with one rank per GPU): device-tensor all-gather of contrastive rows, flat-bucket gradient all-reduce, one JSON
line from rank 0."""
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
