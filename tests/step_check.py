"""Shared by the CPU (emulated kernels) and GPU (real kernels) train-step tests: run one
``TrainStep.step`` on the seeded synthetic batch of a golden fixture and compare losses, outputs,
gradients, post-step parameters and BatchNorm buffers with what the reference produced."""
import os

import numpy as np
import torch

from budget import Budget, rel_l2, rel_max
from oracle import swiftnet_oracle as O


def close(a, b, rtol, what=""):
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=rtol * max(np.abs(b).max(), 1e-30), err_msg=what)


def close_l2(a, b, rtol, what=""):
    """Relative L2 error (robust to the few elements an ill-conditioned fixture flips)."""
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    err = np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-30)
    assert err <= rtol, (what, err)


def check_argmax(logits, ref_argmax, rtol):
    """Class ids must equal the reference's except at numerical near-ties: where they differ, our logit of
    the reference's class must be within rtol*max|logit| of our maximum (and such pixels must be rare).
    Used where no float64 anchor exists (oracle-only cases); the golden fixtures use argmax_vs_anchor."""
    lg = logits.detach().cpu()
    am = lg.argmax(1).numpy().astype(np.uint8)
    bad = am != ref_argmax
    frac = float(bad.mean())
    if frac == 0.0:
        return 0
    assert frac < 1e-3, f"argmax mismatch fraction {frac}"
    ref = torch.from_numpy(ref_argmax.astype(np.int64)).unsqueeze(1)
    gap = (lg.max(1, keepdim=True)[0] - lg.gather(1, ref)).squeeze(1).numpy()
    assert gap[bad].max() <= rtol * float(lg.abs().max()), (frac, float(gap[bad].max()))
    return int(bad.sum())


def argmax_vs_anchor(bud, logits, g, g64, stride=1, expect_exact=None):
    """north_star: class-id argmax bit-exact.  Counts the pixels whose class id differs from the reference's float32
    result and records the count.  Every differing pixel must be a pixel the reference itself cannot decide in
    float32: its own best-vs-second logit margin (stored per pixel) must not exceed 2 x K x the reference's own
    float32 logit error (|ref32 - ref64|, absolute).  ``expect_exact``: the count must be 0 (fixtures where it is)."""
    from budget import K
    lg = logits.detach().cpu()[:, :, ::stride, ::stride]
    am = lg.argmax(1).numpy().astype(np.uint8)
    bad = am != g["seg_argmax"]
    n_bad = int(bad.sum())
    n_ref = int((g["seg_argmax"] != g64["seg_argmax"]).sum())
    e32_abs = float(np.abs(g["seg_logits_sub"].astype(np.float64) - g64["seg_logits_sub"]).max())
    worst = float(g["seg_margin"].astype(np.float64)[bad].max()) if n_bad else 0.0
    bud.note("argmax", mismatches_hip_vs_ref32=n_bad, mismatches_ref32_vs_ref64=n_ref, pixels=int(bad.size),
             worst_ref_margin_at_mismatch=worst, ref32_logit_abs_err=e32_abs)
    if expect_exact:
        if n_bad != 0:
            bud.failures.append(f"argmax: {n_bad} of {bad.size} class ids differ from the reference (expected 0)")
    elif n_bad:
        # fp16 storage of the margin: one ulp of slack
        if worst > 2 * K * e32_abs * 1.001 + 1e-3 * worst:
            bud.failures.append(f"argmax: {n_bad} mismatches, one at a pixel the reference decides by {worst:.3e} "
                                f"(> 2K x its fp32 logit error {e32_abs:.3e})")
        if n_bad > max(4 * n_ref, 8):
            bud.failures.append(f"argmax: {n_bad} mismatches vs {n_ref} between the reference's own fp32 and fp64 runs")
    return n_bad


def load_anchor(golden_dir, fname):
    path = os.path.join(golden_dir, fname.replace(".npz", ".f64.npz"))
    return np.load(path, allow_pickle=False) if os.path.exists(path) else None


def run_and_check_step(ts, g, criterion, shape, rng_seed, rtol, device="cpu", grad_rtol=None, g64=None, name=None,
                       argmax_exact=None, argmax_stride=1):
    """rtol: outputs/losses (north_star: 1e-3).  With the float64 anchor ``g64`` of the fixture, gradients, gradient
    norms and BatchNorm buffers are held to K x the reference's own float32 error (tests/budget.py); without it
    (CPU host-logic tests on emulated kernels) to ``grad_rtol``."""
    grad_rtol = grad_rtol or rtol
    bud = Budget(name or ("step_" + criterion))
    img, labels, ldw, weather, cw = O.synthetic_batch(shape["b"], shape["h"], shape["w"], seed=shape["seed"],
                                                      two_crops=shape["two"], cell=shape["cell"])
    ts.criterion.weight = cw
    b = shape["b"]
    s0 = dict(left=img[:b], label=labels.clone(), weather=weather, label_distance_weight=ldw)
    sample = (s0, dict(left=img[b:])) if shape["two"] else s0
    torch.manual_seed(rng_seed)
    out = ts.step(sample)
    close(out["total"].reshape(()), g["total"], rtol, "total")
    for k in ("supcon", "pixel", "seg", "ce"):
        if float(g[k]) != 0.0:
            close(out[k].reshape(()), g[k], rtol, k)
    s0_, s1_, s2_ = [int(v) for v in g["sub_strides"]] if "sub_strides" in g.files else (2, 4, 8)
    sub = dict(before_sub=out["left_seg_beforeup"][:, :, ::s0_, ::s0_], fine_feat_sub=out["fine_feat"][:, :, ::s1_, ::s1_],
               seg_logits_sub=out["left_seg"][:, :, ::s2_, ::s2_])
    for key, t in sub.items():
        close(t, g[key], rtol, key)
    if g64 is not None:
        for k in ("total", "supcon", "pixel", "seg", "ce"):
            if float(g[k]) != 0.0:
                bud.note("loss " + k, err_hip=abs(float(out[k]) - float(g64[k])) / abs(float(g64[k])),
                         err_ref32=abs(float(g[k]) - float(g64[k])) / abs(float(g64[k])))
        for key, t in sub.items():
            bud.note("output " + key, err_hip=rel_max(t, g64[key]), err_ref32=rel_max(g[key], g64[key]))
        argmax_vs_anchor(bud, out["left_seg"], g, g64, argmax_stride, argmax_exact)
    else:
        check_argmax(out["left_seg"][:, :, ::argmax_stride, ::argmax_stride], g["seg_argmax"], rtol)
    if criterion != "crossentropy" and "labels_after" in g.files:
        assert np.array_equal(out["labels"].cpu().numpy().astype(np.int16), g["labels_after"])
    if "anchor_y" in g.files:
        img_i, cls, pix, n_view = ts.pixelcontrast_criterion.last_anchors
        T = len(cls)
        assert np.array_equal(np.asarray(cls, dtype=np.float32), g["anchor_y"])
        if "anchor_pix" in g.files:                      # the very pixels the reference's sampler drew
            assert np.array_equal(np.asarray(img_i), g["anchor_img"])
            assert np.array_equal(pix.cpu().numpy().T.astype(np.int32), g["anchor_pix"])
        ff = out["fine_feat"].detach().cpu()[:b].permute(0, 2, 3, 1).reshape(b, -1, 128)
        pixc = pix.cpu().long()
        mine = torch.stack([ff[torch.tensor(img_i), pixc[v]] for v in range(n_view)], dim=1)   # [T, n_view, C]
        close(mine, g["anchor_x"], rtol, "anchors (same pixels sampled as the reference)")
    names = [str(s) for s in g["grad_names"]]
    params = dict(ts.model.named_parameters())
    def e32_of(key):                                      # reference fp32 error over its execution paths, if stored
        return float(g64["e32::" + key]) if (g64 is not None and "e32::" + key in g64.files) else None

    if g64 is not None:
        n32, n64 = g["grad_norms"], g64["grad_norms"]
        live = n64 > 0
        e32n = np.abs(n32 - n64) / np.maximum(n64, 1e-300)
        if "e32::grad_norms" in g64.files:
            e32n = np.maximum(e32n, g64["e32::grad_norms"])
        # A norm is ONE number: the ratio of two independent rounding-error draws scatters widely (the reference's own
        # 87 norm errors span two decades on every fixture), so every norm is held to K x the reference's WORST
        # relative norm error on this fixture; the stable per-tensor ratios are taken on the full tensors below.
        worst_n = float(e32n[live].max())
        bn_keys = [k for k in g.files if k.startswith("post::") and "running_" in k]
        worst_bn = max(max(rel_max(g[k], g64[k]), e32_of(k) or 0.0) for k in bn_keys)
    for i, (k, n) in enumerate(zip(names, g["grad_norms"])):
        p = params[k]
        if n < 0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        gn = float(p.grad.double().norm())
        if g64 is None:
            assert abs(gn - n) <= grad_rtol * max(n, 1e-6) + 1e-7, (k, gn, n)
        else:
            bud.check("|grad| " + k, gn, n, float(g64["grad_norms"][i]), metric=rel_max, floor=worst_n)
    sd = ts.model.state_dict()
    for key in g.files:
        if key.startswith("grad::"):
            if g64 is None:
                close_l2(params[key[6:]].grad, g[key], grad_rtol, key)
            else:
                bud.family("stored gradient tensors", key, params[key[6:]].grad, g[key], g64[key], e32=e32_of(key))
        if key.startswith("post::"):
            v = sd[key[6:]]
            if "num_batches" in key:
                assert int(v) == int(g[key]), key
            elif "running_" in key:
                if g64 is None:
                    close(v, g[key], grad_rtol, key)
                else:
                    bud.check(key, v, g[key], g64[key], metric=rel_max, floor=worst_bn, e32=e32_of(key))
            else:
                # Adam's first step moves every weight by ~lr*sign(g): elements whose gradient is ~0 (|g| ~ eps)
                # flip with rounding noise, so allow a tiny fraction of elements to differ by up to 2*lr.
                a = v.detach().cpu().double().numpy()
                d = np.abs(a - g[key].astype(np.float64))
                tol = 1e-2 * max(np.abs(g[key]).max(), 1e-30)
                assert (d > tol).mean() < 2e-3 and d.max() <= 2.5 * ts.opts.lr, (key, float(d.max()))
    pr = ts.supcon_criterion.projection
    for i, gp in enumerate([pr[0].weight.grad, pr[0].bias.grad, pr[2].weight.grad, pr[2].bias.grad]):
        if f"proj_grad_{i}" in g.files:
            if g64 is None:
                close_l2(gp, g[f"proj_grad_{i}"], grad_rtol, f"proj{i}")
            else:
                bud.family("projection-head gradients", f"proj_grad_{i}", gp, g[f"proj_grad_{i}"], g64[f"proj_grad_{i}"],
                           e32=e32_of(f"proj_grad_{i}"))
    pn = {str(k): float(v) for k, v in zip(g["post_names"], g["post_norms"])}
    for k, v in pn.items():
        mine = float(sd[k].double().norm())
        assert abs(mine - v) <= 1e-3 * max(v, 1.0), (k, mine, v)
    bud.finish_family("stored gradient tensors")
    bud.finish_family("projection-head gradients")
    bud.finish()
    return out
