"""One rank of a torch.distributed job on the REAL backend ("nccl" = RCCL on ROCm), launched by torch.distributed.run
(tests/test_dist_gpu.py::test_rccl_*).  A single-GPU box can host ONE RCCL rank, so this is world size 1: it proves that
librccl loads, that init_process_group("nccl", device_id=...) works, and that every collective of dcs_amd/dist.py
(broadcast of the flat parameter buffers, fixed-shape all_gather_into_tensor of the padded row blocks, the 3-float
seg-loss all_reduce, the in-place all_reduce of the flat gradient bucket) runs on device memory through RCCL -- for the
SwiftNet step (BASELINE config 4) and the DeepLabV3+ step (config 5) -- and that the data-parallel step equals the plain
step.  N > 1 over xGMI is only run by the driver's 8-GPU scaling bench."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "doubly-contrastive-semseg_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def one_model(deeplab, rank, world):
    from dist_worker import build, shard_sample
    from dcs_amd.dist import DataParallelStep
    from oracle import swiftnet_oracle as O
    B, h, w, cell = (2, 96, 160, 16) if deeplab else (2, 128, 256, 32)
    crit = "supcon_pixelcontrast_focal"
    batch = O.synthetic_batch(B, h, w, seed=49 if deeplab else 43, two_crops=True, cell=cell)
    runs = []
    for dp_mode in (False, True):
        ts = build(crit, B, batch[4], device="cuda:0", deeplab=deeplab, lazy=deeplab)
        stepper = DataParallelStep(ts, rank, world) if dp_mode else ts
        torch.manual_seed(100)
        out = stepper.step(shard_sample(batch, 0, B, True, B))
        torch.cuda.synchronize()
        runs.append(dict(losses={k: float(out[k].detach()) for k in ("total", "supcon", "pixel", "seg")},
                         params={k: p.detach().clone() for k, p in ts.model.named_parameters()},
                         grads={k: p.grad.detach().clone() for k, p in ts.model.named_parameters() if p.grad is not None},
                         anchors=ts.pixelcontrast_criterion.last_anchors[2].clone()))
        del ts, stepper, out
    a, b = runs
    rec = dict(model="deeplabv3plus_resnet101" if deeplab else "swiftnet_rn18",
               same_anchors=bool(torch.equal(a["anchors"], b["anchors"])),
               losses_plain=a["losses"], losses_dp=b["losses"],
               loss_rel_err=max(abs(a["losses"][k] - b["losses"][k]) / abs(a["losses"][k]) for k in a["losses"]),
               grad_tensors=len(a["grads"]),
               grad_rel_l2_max=max(rel(b["grads"][k], a["grads"][k]) for k in a["grads"]),
               param_rel_l2_max=max(rel(b["params"][k], a["params"][k]) for k in a["params"]),
               grads_bitwise=sum(int(torch.equal(a["grads"][k], b["grads"][k])) for k in a["grads"]))
    return rec


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    rec = dict(backend=dist.get_backend(), world=world)
    # the collectives of dcs_amd/dist.py on device memory, by themselves
    from dcs_amd.dist import RowGather, SegLossReduce
    rg = RowGather()
    X = torch.randn(37, 128, device="cuda")
    y = torch.arange(37, device="cuda", dtype=torch.float32) % 5
    buf, start = rg(X, y, 76)
    rec["row_gather_ok"] = bool(buf.shape == (76 * world, 132) and torch.equal(buf[start:start + 37, :128], X)
                                and torch.equal(buf[start:start + 37, 128], y) and bool((buf[start + 37:start + 76, 128] == -1).all()))
    t = SegLossReduce()(torch.tensor([2.5, 1000.0, 1e-3], device="cuda"))
    rec["seg_reduce"] = [float(v) for v in t]
    flat = torch.arange(12_000_000, device="cuda", dtype=torch.float32)          # 48 MB: the SwiftNet gradient bucket
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    rec["flat_allreduce_ok"] = bool(float(flat[-1]) == 11_999_999.0 * world)
    rec["steps"] = [one_model(False, rank, world), one_model(True, rank, world)]
    with open("/proc/self/maps") as f:
        libs = sorted({l.split()[-1] for l in f if "rccl" in l or "libdcs_hip" in l})
    rec["mapped_libraries"] = libs
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("RCCL_WORKER " + json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
