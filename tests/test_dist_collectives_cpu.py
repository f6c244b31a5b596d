"""CPU: the two exchange primitives of the data-parallel step (dcs_amd/dist.py) on FOUR gloo ranks with ragged anchor
counts (including a rank that sampled nothing): the fixed-shape gathered buffer holds every rank's rows at its slot with
label -1 padding behind them (no count exchange, no host synchronisation), and
the globally normalised segmentation loss equals the single-process value."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COUNTS = [5, 0, 17, 3]
CAP = 20


def rows_of(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return torch.randn(COUNTS[rank], 8, generator=g), torch.randint(0, 19, (COUNTS[rank],), generator=g).float()


def seg_of(rank):
    g = torch.Generator().manual_seed(200 + rank)
    n = float(torch.randint(0, 50, (1,), generator=g)) if rank != 1 else 0.0      # rank 1: every pixel ignored
    loss = float(torch.rand(1, generator=g)) if n > 0 else 0.0
    return torch.tensor([loss, n, 1.0 / n if n > 0 else 0.0])


def worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "doubly-contrastive-semseg_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dcs_amd.dist import RowGather, SegLossReduce
    X, y = rows_of(rank)
    buf, start = RowGather()(X, y, CAP)
    red = SegLossReduce()(seg_of(rank))
    q.put((rank, buf, start, red))
    dist.barrier()
    dist.destroy_process_group()


def test_four_rank_ragged_gather_and_loss_reduce():
    world = 4
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, buf, start, red = q.get(timeout=120)
        got[r] = (buf.clone(), start, red.clone())
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    X_all = torch.cat([rows_of(r)[0] for r in range(world)])
    y_all = torch.cat([rows_of(r)[1] for r in range(world)])
    segs = torch.stack([seg_of(r) for r in range(world)])
    n_g = float(segs[:, 1].sum())
    loss_g = float((segs[:, 0] * segs[:, 1]).sum()) / n_g
    for r in range(world):
        buf, start, red = got[r]
        # fixed shape: world x CAP rows of C + 4 floats, rank q's rows at [q * CAP, q * CAP + COUNTS[q]), label -1 beyond
        assert tuple(buf.shape) == (world * CAP, 8 + 4) and start == r * CAP
        valid = buf[:, 8] >= 0
        assert torch.equal(buf[valid][:, :8], X_all) and torch.equal(buf[valid][:, 8], y_all)
        for q_ in range(world):
            blk = buf[q_ * CAP:(q_ + 1) * CAP]
            assert bool((blk[:COUNTS[q_], 8] >= 0).all()) and bool((blk[COUNTS[q_]:, 8] == -1).all())
            assert float(blk[COUNTS[q_]:, :8].abs().max()) == 0.0 if COUNTS[q_] < CAP else True
        assert abs(float(red[0]) - loss_g) < 1e-6 and float(red[1]) == n_g and abs(float(red[2]) - 1.0 / n_g) < 1e-9
