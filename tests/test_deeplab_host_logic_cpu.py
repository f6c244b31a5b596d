"""CPU: host logic of the DeepLabV3+/ResNet-101 engine (dcs_amd/deeplab.py) with emulated kernels:
  * fp32 against the golden vectors produced by the reference's own model (config 5),
  * float64 against the oracle at a non-multiple-of-32 size (exactness of the hand-written backward,
    dilated convolutions, virtual concatenations, dropout mask hand-off)."""
import os

import numpy as np
import pytest
import torch

import emu_ops
from oracle import deeplab_oracle as D
from oracle import swiftnet_oracle as O
from test_deeplab_oracle_golden import oracle_deeplab_step


@pytest.fixture()
def emu(monkeypatch):
    emu_ops.install(monkeypatch)


def build(dt=torch.float32, flat=True, b=2, cw=None, lazy=False, residual_gain=1.0):
    from dcs_amd.trainer import TrainStep, make_opts
    opts = make_opts(criterion="supcon_pixelcontrast_focal", batch_size=b, deeplab=True, model="deeplabv3plus_resnet101",
                     dtype=dt, flat_params=flat, lazy_fine_feat0=lazy)
    ts = TrainStep(opts, class_weight=cw, device="cpu")
    if dt == torch.float64:
        ts.model.double(); ts.supcon_criterion.double(); ts.weather_clf.double()
    state = {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in D.make_state(seed=7, residual_gain=residual_gain).items()}
    ts.model.load_state_dict(state, strict=True)
    proj = [p.to(dt) for p in O.make_proj(seed=9, dim_in=2048)]
    with torch.no_grad():
        p = ts.supcon_criterion.projection
        for dst, src in zip((p[0].weight, p[0].bias, p[2].weight, p[2].bias), proj):
            dst.copy_(src)
    # dropout noise drawn on the host from the CPU generator exactly like F.dropout does in the reference
    ts.model._get_engine().dropout_noise = lambda shape: torch.empty(shape, dtype=dt).bernoulli_(0.9)
    return ts, state, proj


def test_deeplab_step_matches_reference_golden(emu, golden_dir):
    g = np.load(os.path.join(golden_dir, "deeplab_step_b2_128x256.npz"), allow_pickle=False)
    b = 2
    img, labels, ldw, weather, cw = O.synthetic_batch(b, 128, 256, seed=51, two_crops=True, cell=32)
    ts, _, _ = build(cw=cw)
    s0 = dict(left=img[:b], label=labels.clone(), weather=weather, label_distance_weight=ldw)
    torch.manual_seed(321)
    out = ts.step((s0, dict(left=img[b:])), do_optimizer_step=False)
    # fp32 resolution of this 101-layer fixture: the reference's own fp32 and fp64 forwards differ by 2e-3 (logits)
    # and 7e-4 (layer4 features) relative to max, so fp32-vs-fp32 comparisons use 1e-2 / 5e-3; exactness of the
    # graph wiring is pinned at 1e-7 in float64 by the next test.
    for k in ("total", "supcon", "pixel", "seg"):
        assert abs(float(out[k].detach()) - float(g[k])) <= 5e-3 * abs(float(g[k])), (k, float(out[k].detach()), float(g[k]))
    np.testing.assert_allclose(out["left_seg_beforeup"].detach().numpy(), g["before"], rtol=0, atol=1e-2 * np.abs(g["before"]).max())
    ff = out["fine_feat"].detach().numpy()
    np.testing.assert_allclose(ff[:, ::8], g["fine_feat_sub"], rtol=0, atol=5e-3 * np.abs(g["fine_feat_sub"]).max())
    params = dict(ts.model.named_parameters())
    for k, n in zip([str(s) for s in g["grad_names"]], g["grad_norms"]):
        gn = float(params[k].grad.norm())
        assert abs(gn - n) <= 1e-1 * max(n, 1e-6) + 1e-7, (k, gn, n)   # reference fp32 vs fp64: up to 4.2e-2


@pytest.mark.parametrize("lazy", [False, True])
def test_deeplab_step_matches_well_conditioned_golden_on_injected_anchors(emu, golden_dir, lazy):
    """deeplab_step_b4_256x512 (8 crops of 256x512, residual gain 0.25): the reference's own fp32 run is 2e-5 (logits) /
    2e-3 (worst gradient norm) from its float64 run, so the north-star 1e-3 on outputs and 1e-2 on every gradient norm
    hold in absolute terms.  The anchors of the fixture are injected (PixelContrastLoss.forced_anchors): the loss and
    gradient comparison must not depend on which way an argmax near-tie of the forward pass falls (the reference's own
    channels_last run draws other pixels on this very fixture).  Both the materialised and the lazy fine_feat0."""
    g = np.load(os.path.join(golden_dir, "deeplab_step_b4_256x512.npz"), allow_pickle=False)
    b = 4
    img, labels, ldw, weather, cw = O.synthetic_batch(b, 256, 512, seed=53, two_crops=True, cell=32)
    ts, _, _ = build(b=b, cw=cw, lazy=lazy, residual_gain=0.25)
    ts.pixelcontrast_criterion.forced_anchors = (g["anchor_img"], g["anchor_pix"], g["anchor_y"])
    s0 = dict(left=img[:b], label=labels.clone(), weather=weather, label_distance_weight=ldw)
    torch.manual_seed(322)
    out = ts.step((s0, dict(left=img[b:])), do_optimizer_step=False)
    sb, sfc, sfs, s0c, s0s, sl = (int(v) for v in g["sub_strides"])
    for k in ("total", "supcon", "pixel", "seg"):
        assert abs(float(out[k].detach()) - float(g[k])) <= 1e-4 * abs(float(g[k])), (k, float(out[k].detach()), float(g[k]))
    before = out["left_seg_beforeup"].detach().numpy()[:, :, ::sb, ::sb]
    np.testing.assert_allclose(before, g["before"], rtol=0, atol=1e-3 * np.abs(g["before"]).max())
    ff = out["fine_feat"].detach().numpy()[:, ::sfc, ::sfs, ::sfs]
    np.testing.assert_allclose(ff, g["fine_feat_sub"], rtol=0, atol=1e-3 * np.abs(g["fine_feat_sub"]).max())
    am = out["left_seg"].detach().argmax(1).numpy().astype(np.uint8)
    assert int((am != g["seg_argmax"]).sum()) <= 36, int((am != g["seg_argmax"]).sum())   # reference fp32 vs fp64: 18
    params = dict(ts.model.named_parameters())
    worst = 0.0
    for k, n in zip([str(s) for s in g["grad_names"]], g["grad_norms"]):
        worst = max(worst, abs(float(params[k].grad.norm()) - n) / max(n, 1e-9))
    assert worst <= 1e-2, worst
    for key in g.files:
        if key.startswith("grad::"):
            gr = params[key[6:]].grad
            mine = (gr if gr.numel() < 400000 else gr.flatten()[::37]).numpy()
            err = np.linalg.norm((mine - g[key]).ravel()) / np.linalg.norm(g[key].ravel())
            assert err <= 3e-2, (key, err)          # reference fp32 vs fp64 on these tensors: up to 9e-3


def test_deeplab_engine_matches_oracle_fp64_odd_size(emu):
    dt = torch.float64
    b, h, w = 2, 104, 168
    img, labels, ldw, weather, cw = O.synthetic_batch(b, h, w, seed=61, two_crops=True, cell=16)
    ts, state, proj = build(dt, flat=False, b=b, cw=cw.to(dt))
    s0 = dict(left=img[:b].to(dt), label=labels.clone(), weather=weather, label_distance_weight=ldw.to(dt))
    torch.manual_seed(5)
    out = ts.step((s0, dict(left=img[b:].to(dt))), do_optimizer_step=False)
    ref, grads, gproj = oracle_deeplab_step(state, proj, img.to(dt), labels.clone(), ldw.to(dt), weather, cw.to(dt), b, 5)
    assert abs(float(out["total"]) - float(ref["total"])) < 1e-9 * abs(float(ref["total"]))
    np.testing.assert_allclose(out["fine_feat"].detach().numpy(), ref["fine_feat"].numpy(), rtol=0, atol=1e-9)
    np.testing.assert_allclose(out["left_seg"].detach().numpy(), ref["seg_logits"].numpy(), rtol=0, atol=1e-9)
    params = dict(ts.model.named_parameters())
    for k, gref in grads.items():
        err = float((params[k].grad - gref).abs().max()) / max(float(gref.abs().max()), 1e-12)
        assert err < 1e-7, (k, err)
    sd = ts.model.state_dict()
    for k, v in state.items():
        if "running_" in k:
            assert float((sd[k] - v).abs().max()) < 1e-9, k
        if "num_batches" in k:
            assert int(sd[k]) == int(v), k


def test_lazy_fine_feat0_is_the_same_step(emu):
    """with ``lazy_fine_feat0`` the [B,2048,h,w] upsampled feature is never built; anchors are
    interpolated row by row.  Same losses, weather logits and gradients as the materialised path (float64: to rounding)."""
    from dcs_amd.losses import LazyUpsampled
    dt = torch.float64
    b, h, w = 2, 72, 104
    img, labels, ldw, weather, cw = O.synthetic_batch(b, h, w, seed=71, two_crops=True, cell=16)
    outs, grads = [], []
    for lazy in (False, True):
        torch.manual_seed(0)                       # same random weather-classifier head in both builds
        ts, _, _ = build(dt, flat=False, b=b, cw=cw.to(dt), lazy=lazy)
        s0 = dict(left=img[:b].to(dt), label=labels.clone(), weather=weather, label_distance_weight=ldw.to(dt))
        torch.manual_seed(5)
        out = ts.step((s0, dict(left=img[b:].to(dt))), do_optimizer_step=False)
        outs.append(out)
        grads.append({k: p.grad.clone() for k, p in ts.model.named_parameters()})
        if lazy:
            ff0 = ts.model(img[:b].to(dt))[3]
            assert isinstance(ff0, LazyUpsampled) and tuple(ff0.shape) == (b, 2048, h // 4, w // 4)
    for k in ("total", "pixel", "supcon", "seg"):
        assert abs(float(outs[0][k]) - float(outs[1][k])) < 1e-10 * max(1.0, abs(float(outs[0][k]))), k
    # pooled weights of the lazy path are built in fp32 like the kernels' interpolation weights
    assert float((outs[0]["pred_weather"] - outs[1]["pred_weather"]).abs().max()) < 1e-6
    for k, g0 in grads[0].items():
        err = float((grads[1][k] - g0).abs().max()) / max(float(g0.abs().max()), 1e-12)
        assert err < 1e-9, (k, err)
