"""Data-parallel train step: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI).

The reference has no multi-process path (only ``nn.DataParallel`` on resume, utils/init_trainer.py:310-313);
SURVEY.md 8(e) defines the MI355X design implemented here:

  * the batch is sharded by image; BatchNorm statistics stay per rank (the semantics of the reference's
    DataParallel replicas);
  * hard anchors are SAMPLED per rank (``n_view`` depends on the local class count, utils/loss.py:290-291), then
    the sampled pixel embeddings [A_r,128] (+labels) and the projected image embeddings [2B_r,128] (+labels)
    are ALL-GATHERED -- one fixed-shape ``all_gather_into_tensor`` each, padding rows labelled -1, no count
    exchange and no host synchronisation -- so that both contrastive denominators run over the global batch.
    Every rank evaluates the full global loss and back-propagates only its own rows -- no collective in the
    backward of the losses;
  * the segmentation loss is normalised by the GLOBAL count of valid pixels (all-reduce of 2 floats);
  * parameter gradients of all ranks are SUMMED with ONE all-reduce over a flat fp32 bucket (48.2 MB for
    ResNet-18: ~0.3-0.6 ms on xGMI against >=250 ms of backward, so no bucketing/overlap is needed);
  * ``1/batch_size`` in the criterion (trainer.py:158) uses the GLOBAL batch size.

With sum-reduction the result equals the gradient of the single global objective
``(supcon_g + pixel_g)/B_g + 1.2 * seg_g`` evaluated with per-rank BatchNorm.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.distributed as dist


class RowGather:
    """Fixed-shape all-gather of [rows, C] embeddings and their labels -- no count exchange, no host synchronisation.

    Every rank contributes ``cap`` rows of C + 4 floats: its A <= cap real rows [x_0 .. x_{C-1}, label, 0, 0, 0] followed
    by padding rows whose label is -1, which the fused contrast kernels skip (they enter no maximum, norm, denominator,
    loss or gradient).  One ``all_gather_into_tensor`` of [cap, C+4] per contrastive loss; ``__call__`` returns the
    gathered buffer [world * cap, C + 4] (rows of rank r at [r * cap, (r + 1) * cap)) and this rank's first row.  The
    caller hands the strided views buf[:, :C] / buf[:, C] straight to ``ops.contrast_fwd_bwd``.
    ``cap`` must be the same on every rank: it is derived from the per-rank batch size (weak scaling)."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def instance_offset(self) -> int:
        """SimCLR labels images by their index (class_labels=None, utils/loss.py:151-152 ``torch.eye``): indices of
        different ranks must not collide in the gathered set, so rank r labels from r * 2**16 (exact in fp32)."""
        return self.rank << 16

    def __call__(self, X: torch.Tensor, y: torch.Tensor, cap: int):
        A, Cc = X.shape
        if A > cap:
            raise ValueError(f"RowGather: {A} rows exceed the per-rank capacity {cap}")
        buf = torch.zeros((cap, Cc + 4), device=X.device, dtype=X.dtype)
        buf[:, Cc] = -1.0
        if A:
            buf[:A, :Cc] = X
            buf[:A, Cc] = y
        out = torch.empty((self.world * cap, Cc + 4), device=X.device, dtype=X.dtype)
        dist.all_gather_into_tensor(out, buf, group=self.group)
        return out, self.rank * cap


class SegLossReduce:
    """out = [loss_r, N_r, 1/N_r] -> [sum_r(loss_r N_r) / N_g, N_g, 1/N_g]."""

    def __init__(self, group=None):
        self.group = group

    def __call__(self, out: torch.Tensor) -> torch.Tensor:
        t = torch.stack([out[0] * out[1], out[1]])
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        n = t[1]
        inv = torch.where(n > 0, 1.0 / n.clamp_min(1.0), torch.zeros_like(n))
        return torch.stack([t[0] * inv, n, inv])


class DataParallelStep:
    """Wraps a ``TrainStep`` (built with ``opts.batch_size`` = GLOBAL batch) for one rank."""

    def __init__(self, ts, rank: int, world: int, group=None):
        self.ts, self.rank, self.world, self.group = ts, rank, world, group
        rg = RowGather(group)
        ts.supcon_criterion.row_gather = rg
        ts.pixelcontrast_criterion.row_gather = rg
        b_local = max(1, int(ts.opts.batch_size) // world)        # per-rank labelled images (weak scaling: equal shards)
        ts.supcon_criterion.gather_cap = 2 * b_local
        pc = ts.pixelcontrast_criterion
        pc.gather_cap = min(pc.max_samples, pc.max_views * int(ts.opts.num_classes) * b_local)
        red = SegLossReduce(group)
        ts.criterion.dist_reduce = red
        ts.ce_criterion.dist_reduce = red
        self.params: List[torch.nn.Parameter] = [p for p in ts.model.parameters()]
        # identical initial parameters / buffers on every rank
        flat = getattr(ts, "flat", None)
        tensors = ([g["flat_p"] for g in flat.groups] if flat is not None else [p.data for p in ts.model.parameters()])
        tensors += [b for b in ts.model.buffers()] + [p.data for p in ts.supcon_criterion.parameters()]
        for t in tensors:
            dist.broadcast(t, src=0, group=group)
        self._flat = None

    def _allreduce_grads(self):
        flat = getattr(self.ts, "flat", None)
        if flat is not None and all(flat.aliased([p for p in g["params"]]) for g in flat.groups
                                    if any(p.grad is not None for p in g["params"])):
            # gradients already live in the flat per-group buffers: all-reduce them in place, no copy
            for g in flat.groups:
                if any(p.grad is not None for p in g["params"]):
                    dist.all_reduce(g["flat_g"], op=dist.ReduceOp.SUM, group=self.group)
            return
        ps = [p for p in self.params if p.grad is not None]
        n = sum(p.numel() for p in ps)
        if self._flat is None or self._flat.numel() != n:
            self._flat = torch.empty(n, device=ps[0].device, dtype=ps[0].dtype)
        flat, o = self._flat, 0
        views = []
        for p in ps:
            k = p.numel()
            # same element order as the parameter's own storage (channels_last for conv weights)
            v = torch.as_strided(flat, p.size(), p.stride(), o)
            v.copy_(p.grad)
            views.append(v)
            o += k
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        for p, v in zip(ps, views):
            p.grad = v

    def step(self, sample) -> Dict[str, torch.Tensor]:
        out = self.ts.step(sample, do_optimizer_step=False)
        self._allreduce_grads()
        self.ts.optimizer.step()
        return out
