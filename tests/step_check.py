"""Shared by the CPU (emulated kernels) and GPU (real kernels) train-step tests: run one
``TrainStep.step`` on the seeded synthetic batch of a golden fixture and compare losses, outputs,
gradients, post-step parameters and BatchNorm buffers with what the reference produced."""
import numpy as np
import torch

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
    the reference's class must be within rtol*max|logit| of our maximum (and such pixels must be rare)."""
    lg = logits.detach().cpu()
    am = lg.argmax(1).numpy().astype(np.uint8)
    bad = am != ref_argmax
    frac = float(bad.mean())
    if frac == 0.0:
        return
    assert frac < 1e-3, f"argmax mismatch fraction {frac}"
    ref = torch.from_numpy(ref_argmax.astype(np.int64)).unsqueeze(1)
    gap = (lg.max(1, keepdim=True)[0] - lg.gather(1, ref)).squeeze(1).numpy()
    assert gap[bad].max() <= rtol * float(lg.abs().max()), (frac, float(gap[bad].max()))


def run_and_check_step(ts, g, criterion, shape, rng_seed, rtol, device="cpu", grad_rtol=None):
    """rtol: outputs/losses.  grad_rtol: gradients and post-step state (defaults to rtol)."""
    grad_rtol = grad_rtol or rtol
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
    close(out["left_seg_beforeup"][:, :, ::2, ::2], g["before_sub"], rtol, "before")
    close(out["fine_feat"][:, :, ::4, ::4], g["fine_feat_sub"], rtol, "fine_feat")
    close(out["left_seg"][:, :, ::8, ::8], g["seg_logits_sub"], rtol, "seg")
    check_argmax(out["left_seg"], g["seg_argmax"], rtol)
    if criterion != "crossentropy":
        assert np.array_equal(out["labels"].cpu().numpy().astype(np.int16), g["labels_after"])
    if "anchor_y" in g.files:
        img_i, cls, pix, n_view = ts.pixelcontrast_criterion.last_anchors
        T = len(cls)
        assert np.array_equal(np.asarray(cls, dtype=np.float32), g["anchor_y"])
        ff = out["fine_feat"].detach().cpu()[:b].permute(0, 2, 3, 1).reshape(b, -1, 128)
        pixc = pix.cpu().long()
        mine = torch.stack([ff[torch.tensor(img_i), pixc[v]] for v in range(n_view)], dim=1)   # [T, n_view, C]
        close(mine, g["anchor_x"], rtol, "anchors (same pixels sampled as the reference)")
    names = [str(s) for s in g["grad_names"]]
    params = dict(ts.model.named_parameters())
    for k, n in zip(names, g["grad_norms"]):
        p = params[k]
        if n < 0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        gn = float(p.grad.norm())
        assert abs(gn - n) <= grad_rtol * max(n, 1e-6) + 1e-7, (k, gn, n)
    sd = ts.model.state_dict()
    for key in g.files:
        if key.startswith("grad::"):
            close_l2(params[key[6:]].grad, g[key], grad_rtol, key)
        if key.startswith("post::"):
            v = sd[key[6:]]
            if "num_batches" in key:
                assert int(v) == int(g[key]), key
            elif "running_" in key:
                close(v, g[key], grad_rtol, key)
            else:
                # Adam's first step moves every weight by ~lr*sign(g): elements whose gradient is ~0 (|g| ~ eps)
                # flip with rounding noise, so allow a tiny fraction of elements to differ by up to 2*lr.
                a = v.detach().cpu().double().numpy()
                d = np.abs(a - g[key].astype(np.float64))
                tol = grad_rtol * max(np.abs(g[key]).max(), 1e-30)
                assert (d > tol).mean() < 2e-3 and d.max() <= 2.5 * ts.opts.lr, (key, float(d.max()))
    pr = ts.supcon_criterion.projection
    for i, gp in enumerate([pr[0].weight.grad, pr[0].bias.grad, pr[2].weight.grad, pr[2].bias.grad]):
        if f"proj_grad_{i}" in g.files:
            close_l2(gp, g[f"proj_grad_{i}"], grad_rtol, f"proj{i}")
    pn = {str(k): float(v) for k, v in zip(g["post_names"], g["post_norms"])}
    for k, v in pn.items():
        mine = float(sd[k].double().norm())
        assert abs(mine - v) <= max(grad_rtol * 0.1, 2e-5) * max(v, 1.0), (k, mine, v)
    return out
