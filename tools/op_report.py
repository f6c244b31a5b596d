#!/usr/bin/env python3
"""Per-operation roofline of one C3 train step: every public function of dcs_amd.ops is wrapped with a HIP event pair
and an algorithmic byte count (every distinct tensor argument / result counted once: read once or written once), then
`steps` train steps run and the totals are reported per operation as achieved TB/s against the 8 TB/s HBM3E peak
(convolutions additionally as TFLOP/s in bench.py).  Events around every call serialise nothing (same stream) but add
launch overhead, so the step itself runs a few % slower than in bench.py.

usage: op_report.py [--steps 3] [--out profiles/xxx.json] [--batch 16 --height 1024 --width 2048]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "doubly-contrastive-semseg_amd")):
    sys.path.insert(0, p)
import torch

HBM_PEAK_TBS = 8.0
SPARSE = {"gather_rows", "scatter_add_rows", "gather_rows_bilinear", "scatter_rows_bilinear"}


def tensors_of(obj, acc, mask=False):
    if torch.is_tensor(obj):
        if obj.is_cuda:
            m8 = getattr(obj, "_mask8", None) if mask else None
            if m8 is not None and os.environ.get("DCS_MASK8", "1") != "0":      # read through its byte mask (ops.bn_bwd masksrc)
                acc[m8.data_ptr()] = m8.numel()
                return
            acc[obj.data_ptr()] = max(acc.get(obj.data_ptr(), 0), obj.numel() * obj.element_size())
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            tensors_of(o, acc)
    elif isinstance(obj, dict):
        for o in obj.values():
            tensors_of(o, acc)


class OpProfiler:
    """Wraps every public function of dcs_amd.ops with a HIP event pair and an algorithmic byte count.  Level batching
    must be off while it is enabled (DCS_LEVEL_BATCH=0: a recorded launch leaves the library later than the function that
    recorded it returns); the HBM-bound kernels are launched per level either way."""

    SKIP = {"require_device", "out_size", "out_size_d", "krsc", "geom_fwd", "geoms_dgrad", "geom_stem", "geom_stem_fwd",
            "seg_loss_fused_ok", "level_batch", "x3_ok", "x3w_ok"}
    # not HBM-bound by design: matrix-core work (convolutions, linear layers, the similarity loss), weight repacks and
    # the few-row sampler / gather kernels
    NOT_HBM = ("conv_", "linear", "contrast", "split_weight", "pack_", "anchor_", "gather_rows", "scatter_add_rows",
               "scatter_rows", "stem_", "transpose", "adam_step", "sum_scalar", "scale_inplace", "seg_loss_final")

    def __init__(self, ops):
        self.ops, self.records, self.enabled, self._saved, self._depth = ops, [], False, {}, 0
        for name in dir(ops):
            fn = getattr(ops, name)
            if name.startswith("_") or name in self.SKIP or not callable(fn) or isinstance(fn, type) or \
                    getattr(fn, "__module__", "") != ops.__name__:
                continue
            self._saved[name] = fn
            setattr(ops, name, self._wrap(name, fn))

    def _wrap(self, name, fn):
        def inner(*a, **k):
            if not self.enabled or self._depth:          # an op called by another op belongs to the outer one
                return fn(*a, **k)
            acc = {}
            tensors_of(a, acc)
            tensors_of({kk: v for kk, v in k.items() if kk != "masksrc"}, acc)
            tensors_of(k.get("masksrc"), acc, mask=True)    # a ReLU-mask source is read as 1 byte per float4 when it has one
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self._depth += 1
            try:
                out = fn(*a, **k)
            finally:
                self._depth -= 1
            e1.record()
            if name in SPARSE:                       # row gathers / scatters touch only the selected rows
                rows = [t for t in (list(a) + [out]) if torch.is_tensor(t) and t.dim() == 2 and t.dtype == torch.float32]
                nbytes = 2 * min(t.numel() * 4 for t in rows) if rows else 0
            else:
                tensors_of(out, acc)
                nbytes = sum(acc.values())
            self.records.append((name, nbytes, e0, e1))
            return out
        return inner

    def restore(self):
        for name, fn in self._saved.items():
            setattr(self.ops, name, fn)

    def rows(self, steps):
        agg = {}
        for name, nbytes, e0, e1 in self.records:
            a = agg.setdefault(name, [0, 0.0, 0.0])
            a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += nbytes
        rows = []
        for name, (n, ms, nb) in agg.items():
            rows.append({"op": name, "calls_per_step": n / steps, "ms_per_step": ms / steps,
                         "algorithmic_gb_per_step": nb / steps / 1e9,
                         "achieved_tb_per_s": nb / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
                         "frac_of_hbm_peak": nb / (ms * 1e-3) / 1e12 / HBM_PEAK_TBS if ms > 0 else 0.0})
        rows.sort(key=lambda r: -r["ms_per_step"])
        return rows

    def hbm_family(self, steps, top=8):
        """Aggregate of the HBM-bound operations: algorithmic bytes (every distinct tensor argument / result once),
        HIP-event time, achieved TB/s against the 8 TB/s peak."""
        rows = [r for r in self.rows(steps) if not r["op"].startswith(self.NOT_HBM)]
        gb, ms = sum(r["algorithmic_gb_per_step"] for r in rows), sum(r["ms_per_step"] for r in rows)
        return {"bound": "hbm", "algorithmic_gb_per_step": gb, "ms_per_step": ms, "achieved": gb / ms if ms > 0 else 0.0,
                "peak": HBM_PEAK_TBS, "unit": "TB/s", "frac": gb / ms / HBM_PEAK_TBS if ms > 0 else 0.0,
                "ops": [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()} for r in rows[:top]],
                "note": "all dcs_amd.ops functions except matrix-core work (conv_*, linear*, contrast*), weight repacks and "
                        "few-row gathers; measured in this run with HIP events around every call, level batching off, "
                        "in extra steps after the timed region"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    os.environ["DCS_LEVEL_BATCH"] = "0"
    import dcs_amd.ops as ops
    from dcs_amd.trainer import TrainStep, make_opts
    from oracle import swiftnet_oracle as O
    import bench
    dev = torch.device("cuda", 0)
    left0, left1, labels, ldw, weather, cw = bench.device_batch(O, args.batch, args.height, args.width, 0, True, dev)
    torch.manual_seed(1)
    ts = TrainStep(make_opts(criterion="supcon_pixelcontrast_focal", batch_size=args.batch), class_weight=cw, device=dev)
    prof = OpProfiler(ops)

    def one_step():
        s0 = dict(left=left0, label=labels.clone(), weather=weather, label_distance_weight=ldw)
        return ts.step((s0, dict(left=left1)))

    for _ in range(2):
        one_step()
    torch.cuda.synchronize()
    prof.enabled = True
    m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    m0.record()
    for _ in range(args.steps):
        one_step()
    m1.record()
    torch.cuda.synchronize()
    prof.enabled = False
    step_ms = m0.elapsed_time(m1) / args.steps
    rows = prof.rows(args.steps)
    res = {"workload": f"C3 B={args.batch} x 2 crops at {args.width}x{args.height}", "steps": args.steps,
           "step_ms_instrumented": step_ms, "hbm_peak_tb_per_s": HBM_PEAK_TBS, "ops": rows,
           "hbm_family": prof.hbm_family(args.steps),
           "note": "nested ops (e.g. bn_bwd = partial + final + apply, conv_wgrad = kernel + slab reduce) are reported "
                   "at the level of the dcs_amd.ops function; MFMA-bound ops (conv_*, linear*) are far below the HBM "
                   "roof by design; level batching off (launches leave the library inside the function that is timed)"}
    print(f"{'op':28s} {'calls':>6s} {'ms/step':>8s} {'GB/step':>8s} {'TB/s':>6s} {'% HBM':>6s}")
    for r in rows:
        print(f"{r['op']:28s} {r['calls_per_step']:6.1f} {r['ms_per_step']:8.2f} {r['algorithmic_gb_per_step']:8.2f} "
              f"{r['achieved_tb_per_s']:6.2f} {100 * r['frac_of_hbm_peak']:6.1f}")
    print(f"instrumented step: {step_ms:.1f} ms")
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
