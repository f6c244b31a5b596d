"""One rank of a world_size-2 gloo job running the data-parallel step (dcs_amd/dist.py).
CPU (tests/test_dist_cpu.py): kernels emulated (tests/emu_ops.py).  GPU (tests/test_dist_gpu.py, DCS_DIST_DEVICE=cuda):
the REAL kernels, both ranks sharing cuda:0 (gloo moves the device tensors; the production job uses RCCL, one rank per
GPU -- a single GPU cannot host two RCCL ranks)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "doubly-contrastive-semseg_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

DEVICE = os.environ.get("DCS_DIST_DEVICE", "cpu")


class _MP:
    def setattr(self, obj, name, val):
        setattr(obj, name, val)


def build(criterion, global_batch, cw, device=None, deeplab=False, lazy=False):
    """deeplab=True: BASELINE config 5's model (DeepLabV3+ / ResNet-101) on the well-conditioned seeded state."""
    from dcs_amd.trainer import TrainStep, make_opts
    from oracle import swiftnet_oracle as O
    kw = dict(deeplab=True, model="deeplabv3plus_resnet101", lazy_fine_feat0=lazy) if deeplab else {}
    ts = TrainStep(make_opts(criterion=criterion, batch_size=global_batch, **kw), class_weight=cw, device=device or DEVICE)
    if deeplab:
        from oracle import deeplab_oracle as D
        ts.model.load_state_dict(D.make_state(seed=7, residual_gain=0.25), strict=True)
        proj = O.make_proj(seed=9, dim_in=2048)
        # dropout keep mask from the CPU generator, like F.dropout in the reference (each rank seeds its own stream)
        ts.model._get_engine().dropout_noise = lambda shape: torch.empty(shape).bernoulli_(0.9)
    else:
        ts.model.load_state_dict(O.make_state(seed=1), strict=True)
        proj = O.make_proj(seed=2)
    with torch.no_grad():
        p = ts.supcon_criterion.projection
        for dst, src in zip((p[0].weight, p[0].bias, p[2].weight, p[2].bias), proj):
            dst.copy_(src)
    return ts


def shard_sample(batch, lo, hi, two, B):
    img, labels, ldw, weather, cw = batch
    s0 = dict(left=img[lo:hi], label=labels[lo:hi].clone(), weather=weather[lo:hi], label_distance_weight=ldw[lo:hi])
    return (s0, dict(left=img[B + lo:B + hi])) if two else s0


def cpu(t):
    return t.detach().cpu().contiguous().clone()


def main(rank, world, port, outdir):
    if DEVICE == "cpu":
        import emu_ops
        emu_ops.install(_MP())
    else:
        torch.cuda.set_device(0)
    import dcs_amd.ops as ops
    from dcs_amd.dist import DataParallelStep
    from oracle import swiftnet_oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    res = {}
    B, h, w = 2, 128, 256
    # ---- scenario A: eval-mode BatchNorm (per-image independent model) => DP must equal the single-process run
    batch = O.synthetic_batch(B, h, w, seed=41, two_crops=True, cell=32)
    ts = build("supcon_focal", B, batch[4])
    ts.model.eval()
    dp = DataParallelStep(ts, rank, world)
    out = dp.step(shard_sample(batch, rank, rank + 1, True, B))
    res["A_total"] = cpu(out["total"].reshape(()))
    res["A_grads"] = {k: cpu(p.grad) for k, p in ts.model.named_parameters() if p.grad is not None}
    res["A_params"] = {k: cpu(p) for k, p in ts.model.named_parameters()}
    # ---- scenario B: training-mode, doubly-contrastive, per-rank sampling + global denominators
    captured = []
    orig = ops.contrast_fwd_bwd
    def spy(X, y, mode, temperature=0.07, mask=None):
        valid = (y >= 0).cpu()
        captured.append((cpu(X)[valid], cpu(y)[valid], mode, int(X.shape[0])))
        return orig(X, y, mode, temperature, mask=mask)
    ops.contrast_fwd_bwd = spy
    batch = O.synthetic_batch(B, h, w, seed=43, two_crops=True, cell=32)
    ts = build("supcon_pixelcontrast_focal", B, batch[4])
    dp = DataParallelStep(ts, rank, world)
    torch.manual_seed(100 + rank)
    out = dp.step(shard_sample(batch, rank, rank + 1, True, B))
    for k in ("total", "supcon", "pixel", "seg"):
        res["B_" + k] = cpu(out[k].reshape(()))
    pix = [c for c in captured if c[2] == 0][0]
    res["B_pixel_rows"], res["B_pixel_labels"], res["B_gathered_rows"] = pix[0], pix[1], torch.tensor(pix[3])
    res["B_local_anchor_count"] = torch.tensor(len(ts.pixelcontrast_criterion.last_anchors[1]) *
                                               ts.pixelcontrast_criterion.last_anchors[3])
    res["B_param_checksum"] = torch.stack([cpu(p).double().sum() for p in ts.model.parameters()])
    # ---- scenario C: SimCLR image contrast (class_labels=None): instance ids must stay unique across ranks; with
    #      eval-mode BatchNorm the DP loss and gradients equal the single-process run on the union batch
    captured.clear()
    batch = O.synthetic_batch(B, h, w, seed=45, two_crops=True, cell=32)
    ts = build("supcon_simclr_focal", B, batch[4])
    ts.model.eval()
    dp = DataParallelStep(ts, rank, world)
    out = dp.step(shard_sample(batch, rank, rank + 1, True, B))
    res["C_total"], res["C_simclr"] = cpu(out["total"].reshape(())), cpu(out["simclr"].reshape(()))
    res["C_labels"] = [c for c in captured if c[2] == 1][0][1]
    res["C_grads"] = {k: cpu(p.grad) for k, p in ts.model.named_parameters() if p.grad is not None}
    res["C_proj_grads"] = [cpu(p.grad) for p in ts.supcon_criterion.projection.parameters()]
    # ---- scenario D: rank 1's shard is all "ignore": its sampler finds no class, it must still join the collectives
    #      (no hang), report the global pixel loss and get a zero pixel gradient
    captured.clear()
    batch = list(O.synthetic_batch(B, h, w, seed=47, two_crops=False, cell=32))
    batch[1] = batch[1].clone()
    batch[1][1] = 255
    batch[2] = batch[2].clone()
    batch[2][1] = 0.0
    ts = build("pixelcontrast_focal", B, batch[4])
    dp = DataParallelStep(ts, rank, world)
    torch.manual_seed(7 + rank)
    out = dp.step(shard_sample(tuple(batch), rank, rank + 1, False, B))
    res["D_pixel"], res["D_seg"], res["D_total"] = (cpu(out[k].reshape(())) for k in ("pixel", "seg", "total"))
    pix = [c for c in captured if c[2] == 0][0]
    res["D_pixel_rows"], res["D_pixel_labels"] = pix[0], pix[1]
    res["D_local_anchor_count"] = torch.tensor(len(ts.pixelcontrast_criterion.last_anchors[1]) *
                                               ts.pixelcontrast_criterion.last_anchors[3])
    res["D_param_checksum"] = torch.stack([cpu(p).double().sum() for p in ts.model.parameters()])
    # ---- scenario E (BASELINE config 5's model under the same wrapper): DeepLabV3+ / ResNet-101, eval-mode BatchNorm
    #      (dropout off) => the DP step on two half-batches equals the single-process step on the full batch
    captured.clear()
    hd, wd = 96, 160
    batch = O.synthetic_batch(B, hd, wd, seed=49, two_crops=True, cell=16)
    ts = build("supcon_focal", B, batch[4], deeplab=True)
    ts.model.eval()
    dp = DataParallelStep(ts, rank, world)
    out = dp.step(shard_sample(batch, rank, rank + 1, True, B))
    res["E_total"] = cpu(out["total"].reshape(()))
    res["E_grads"] = {k: cpu(p.grad) for k, p in ts.model.named_parameters() if p.grad is not None}
    # ---- scenario F: DeepLab, training mode, doubly-contrastive with the lazy 2048-channel fine_feat0: per-rank sampling,
    #      global denominators over 2048-wide anchor rows, flat-bucket all-reduce of the 58.7 M-parameter gradient
    captured.clear()
    Bf = 4                       # 2 labelled images per rank: the ASPP pooling branch's BatchNorm needs > 1 value per channel
    batch = O.synthetic_batch(Bf, hd, wd, seed=50, two_crops=True, cell=16)
    ts = build("supcon_pixelcontrast_focal", Bf, batch[4], deeplab=True, lazy=True)
    dp = DataParallelStep(ts, rank, world)
    torch.manual_seed(200 + rank)
    out = dp.step(shard_sample(batch, 2 * rank, 2 * rank + 2, True, Bf))
    for k in ("total", "supcon", "pixel", "seg"):
        res["F_" + k] = cpu(out[k].reshape(()))
    pix = [c for c in captured if c[2] == 0][0]
    res["F_pixel_rows"], res["F_pixel_labels"], res["F_gathered_rows"] = pix[0], pix[1], torch.tensor(pix[3])
    res["F_local_anchor_count"] = torch.tensor(len(ts.pixelcontrast_criterion.last_anchors[1]) *
                                               ts.pixelcontrast_criterion.last_anchors[3])
    res["F_param_checksum"] = torch.stack([cpu(p).double().sum() for p in ts.model.parameters()])
    ops.contrast_fwd_bwd = orig
    torch.save(res, os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
