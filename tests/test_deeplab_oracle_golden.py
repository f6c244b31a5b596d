"""CPU: the DeepLabV3+/ResNet-101 oracle (oracle/deeplab_oracle.py, BASELINE config 5) against golden vectors produced
by running the reference's own model (tests/golden/make_golden_deeplab.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import deeplab_oracle as D
from oracle import swiftnet_oracle as O


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def close(a, b, rtol=2e-5):
    a = np.asarray(a.detach() if torch.is_tensor(a) else a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=rtol * max(np.abs(b).max(), 1e-30))


def oracle_deeplab_step(state, proj, img, labels, ldw, weather, cw, b, seed):
    names = D.trainable_names(state)
    for k in names:
        state[k].requires_grad_(True)
    pw = [p.requires_grad_(True) for p in proj]
    torch.manual_seed(seed)
    seg, before, ff, ff0 = D.deeplab_forward(img, state, True, True)
    sup = O.supcon_loss(ff, pw, weather)
    pix, sel = O.pixel_contrast_loss(ff0, labels, before, return_indices=True)
    segl = O.boundary_aware_focal_loss(seg, labels, ldw, cw)
    total = 1 / b * (sup + pix) + segl * 1.2
    g = torch.autograd.grad(total, [state[k] for k in names] + pw, allow_unused=True)
    for k in names:
        state[k].requires_grad_(False)
    out = dict(total=total.detach(), supcon=sup.detach(), pixel=pix.detach(), seg=segl.detach(), before=before.detach(),
               fine_feat=ff.detach(), fine_feat0=ff0.detach(), seg_logits=seg.detach(),
               anchors=(sel[0].numpy(), sel[2].numpy(), sel[1].numpy().astype(np.float32)))     # img [T], pix [T, n_view], cls [T]
    return out, dict(zip(names, g[:len(names)])), list(g[len(names):])


# name -> (batch, height, width, data seed, generator seed, residual gain): tests/golden/make_golden_deeplab.py::FIXTURES
STEP_FIXTURES = {
    "deeplab_step_b2_128x256": (2, 128, 256, 51, 321, 1.0),
    "deeplab_step_b4_256x512": (4, 256, 512, 53, 322, 0.25),
}


def sub_strides(g):
    return tuple(int(v) for v in g["sub_strides"]) if "sub_strides" in g.files else (1, 8, 1, 16, 2, 4)


@pytest.mark.parametrize("name", list(STEP_FIXTURES))
def test_deeplab_train_step_matches_reference(golden_dir, name):
    g = load(golden_dir, name + ".npz")
    b, h, w, dseed, rseed, gain = STEP_FIXTURES[name]
    state = D.make_state(seed=7, residual_gain=gain)
    proj = O.make_proj(seed=9, dim_in=2048)
    img, labels, ldw, weather, cw = O.synthetic_batch(b, h, w, seed=dseed, two_crops=True, cell=32)
    out, grads, gproj = oracle_deeplab_step(state, proj, img, labels.clone(), ldw, weather, cw, b, rseed)
    sb, sfc, sfs, s0c, s0s, sl = sub_strides(g)
    for k in ("total", "supcon", "pixel", "seg"):
        close(out[k], g[k])
    close(out["before"][:, :, ::sb, ::sb], g["before"])
    close(out["fine_feat"][:, ::sfc, ::sfs, ::sfs], g["fine_feat_sub"])
    close(out["fine_feat0"][:, ::s0c, ::s0s, ::s0s], g["fine_feat0_sub"])
    assert np.array_equal(out["seg_logits"].argmax(1).numpy().astype(np.uint8), g["seg_argmax"])
    for k, n in zip([str(s) for s in g["grad_names"]], g["grad_norms"]):
        gn = float(grads[k].norm())
        assert abs(gn - n) <= 1e-4 * max(n, 1e-6) + 1e-7, (k, gn, n)
    for key in g.files:
        if key.startswith("grad::"):
            gg = grads[key[6:]]
            close(gg if gg.numel() < 400000 else gg.flatten()[::37], g[key], rtol=1e-4)
    for k, n in zip([str(s) for s in g["rs_names"]], g["rs_norms"]):
        assert abs(float(state[k].double().norm()) - n) <= 1e-5 * max(n, 1.0), k
    for gp, n in zip(gproj, g["proj_grad_norms"]):
        assert abs(float(gp.norm()) - n) <= 1e-4 * max(n, 1e-6)


def test_deeplab_eval_forward_odd_size(golden_dir):
    g = load(golden_dir, "deeplab_eval_b1_104x168.npz")
    state = D.make_state(seed=7)
    img = O.synthetic_batch(1, 104, 168, seed=52)[0]
    with torch.no_grad():
        seg, before, ff, ff0 = D.deeplab_forward(img, state, training=False)
    close(before, g["before"])
    close(ff[:, ::8], g["fine_feat_sub"])
    close(ff0[:, ::16], g["fine_feat0_sub"])
    assert np.array_equal(seg.argmax(1).numpy().astype(np.uint8), g["seg_argmax"])


def test_resnet_plan_os16_matches_reference_rules():
    plan = D.resnet_plan()
    assert len(plan) == 33
    l4 = [p for p in plan if p[0] == "layer4"]
    assert [(p[4], p[5]) for p in l4] == [(1, 1), (1, 2), (1, 2)]          # stride 1; dilation 1 then 2 (resnet.py:173-195)
    assert sum(int(np.prod(s)) for s, k in D.state_spec().values() if k in ("conv", "bn_w", "bn_b", "bias")) == 58753459
