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
