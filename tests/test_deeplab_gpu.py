"""GPU: DeepLabV3+/ResNet-101 (BASELINE config 5) train step on the HIP kernels against the reference golden and the
oracle.  No tolerance constants: on this randomly initialised 101-layer model with batch 2 the reference's OWN float32
result is 2e-3 (logits) / 4e-2 (gradient norms) / 131 argmax pixels away from its float64 result, so every quantity is
held to K = 2 x the reference's float32 error measured over its float32 execution paths (tests/budget.py; anchors from
tests/golden/make_golden_deeplab.py).  The golden test runs the materialised AND the lazy fine_feat0 path."""
import os

import numpy as np
import pytest
import torch

from budget import Budget, K, rel_l2, rel_max
from oracle import deeplab_oracle as D
from oracle import swiftnet_oracle as O
from test_deeplab_oracle_golden import oracle_deeplab_step

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def build(b, cw, lazy=False, residual_gain=1.0):
    from dcs_amd.trainer import TrainStep, make_opts
    opts = make_opts(criterion="supcon_pixelcontrast_focal", batch_size=b, deeplab=True, model="deeplabv3plus_resnet101",
                     lazy_fine_feat0=lazy)
    ts = TrainStep(opts, class_weight=cw, device=DEV)
    ts.model.load_state_dict(D.make_state(seed=7, residual_gain=residual_gain), strict=True)
    with torch.no_grad():
        p = ts.supcon_criterion.projection
        for dst, src in zip((p[0].weight, p[0].bias, p[2].weight, p[2].bias), O.make_proj(seed=9, dim_in=2048)):
            dst.copy_(src)
    ts.model._get_engine().dropout_noise = lambda shape: torch.empty(shape).bernoulli_(0.9)   # CPU generator, like F.dropout
    return ts


def rel(a, b):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _argmax_budget(bud, seg, g, g64, logits_stride):
    """Class ids: every pixel that differs from the reference's float32 result must be one the reference itself cannot
    decide in float32 (its margin <= 2K x its own fp32 logit error); the count may not exceed K x the reference's own
    fp32-vs-fp64 count (+ a floor of 8 pixels)."""
    am = seg.argmax(1).cpu().numpy().astype(np.uint8)
    bad = am != g["seg_argmax"]
    n_bad, n_ref = int(bad.sum()), int((g["seg_argmax"] != g64["seg_argmax"]).sum())
    e32_abs = float(g64["e32::seg_logits_sub"]) * float(np.abs(g64["seg_logits_sub"]).max()) if "e32::seg_logits_sub" in g64.files \
        else float(np.abs(g["seg_logits_sub"].astype(np.float64) - g64["seg_logits_sub"]).max())
    worst = float(g["seg_margin"].astype(np.float64)[bad].max()) if n_bad else 0.0
    bud.note("argmax", mismatches_hip_vs_ref32=n_bad, mismatches_ref32_vs_ref64=n_ref, pixels=int(bad.size),
             worst_ref_margin_at_mismatch=worst, ref32_logit_abs_err=e32_abs)
    bud.check_abs("argmax mismatch count", n_bad, K * n_ref + 8)
    bud.check_abs("argmax worst reference margin at a mismatch", worst, 2 * K * e32_abs * 1.002 + 1e-3 * worst)


# name -> (batch, height, width, data seed, generator seed, residual gain, well conditioned):
# tests/golden/make_golden_deeplab.py::FIXTURES
STEP_FIXTURES = {
    "deeplab_step_b2_128x256": (2, 128, 256, 51, 321, 1.0, False),
    "deeplab_step_b4_256x512": (4, 256, 512, 53, 322, 0.25, True),
}


@pytest.mark.parametrize("lazy", [False, True])
@pytest.mark.parametrize("name", list(STEP_FIXTURES))
def test_deeplab_step_matches_reference_golden(golden_dir, name, lazy):
    """Everything here runs on the DEFAULT kernels (split-bf16 MFMA convolutions: the benched path).

    The anchors are sampled from the argmax-derived hard / easy split of the 1/4-resolution predictions, so ONE argmax
    near-tie that falls the other way redirects the sampler (the reference does it to itself: its channels_last run
    draws other pixels than its default run on both fixtures, its fp32 and fp64 runs disagree on 131 / 18 class ids).
    The forward budgets and the sampler's result are taken on the free run; when that run did not draw the fixture's
    pixels, the anchor-dependent part (losses, gradients, running statistics) is taken from a second run of the SAME
    kernels with the fixture's anchors injected (PixelContrastLoss.forced_anchors -- what make_golden.pixel_loss does to
    the reference for its float64 run).

    deeplab_step_b2_128x256 is ill-conditioned (random 101-layer init, deepest maps 8x16: the reference's own fp32 run is
    2e-3 / 4e-2 from its fp64 run on logits / gradient norms) and is held to K x the reference's own error only;
    deeplab_step_b4_256x512 (8 crops of 256x512, residual gain 0.25; reference 2e-5 / 2e-3) ALSO meets the north star's
    1e-3 on logits / features / losses and 1e-2 on every gradient norm in absolute terms."""
    b, h, w, dseed, rseed, gain, well = STEP_FIXTURES[name]
    g = np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)
    g64 = np.load(os.path.join(golden_dir, name + ".f64.npz"), allow_pickle=False)
    e32 = lambda k: g64["e32::" + k]
    sb, sfc, sfs, s0c, s0s, sl = (int(v) for v in g["sub_strides"]) if "sub_strides" in g.files else (1, 8, 1, 16, 2, 4)
    img, labels, ldw, weather, cw = O.synthetic_batch(b, h, w, seed=dseed, two_crops=True, cell=32)
    bud = Budget(f"gpu_{name}_lazy{int(lazy)}")
    anchor_y = g["anchor_y"] if "anchor_y" in g.files else None

    def run(forced=None):
        ts_ = build(b, cw, lazy=lazy, residual_gain=gain)
        ts_.pixelcontrast_criterion.forced_anchors = forced
        s0 = dict(left=img[:b], label=labels.clone(), weather=weather, label_distance_weight=ldw)
        torch.manual_seed(rseed)
        out_ = ts_.step((s0, dict(left=img[b:])), do_optimizer_step=False)
        img_i, cls, pix, n_view = ts_.pixelcontrast_criterion.last_anchors
        same = np.array_equal(np.asarray(img_i), g["anchor_img"]) and \
            np.array_equal(pix.cpu().numpy().T.astype(np.int32), g["anchor_pix"])
        return ts_, out_, same, np.asarray(cls, dtype=np.float32)

    ts, out, same, cls = run()
    outputs = (("before", out["left_seg_beforeup"][:, :, ::sb, ::sb], "before"),
               ("fine_feat", out["fine_feat"][:, ::sfc, ::sfs, ::sfs], "fine_feat_sub"),
               ("seg logits", out["left_seg"][:, :, ::sl, ::sl], "seg_logits_sub"))
    for what, mine, key in outputs:
        bud.check(what, mine, g[key], g64[key], metric=rel_max, e32=float(e32(key)))
        if well:
            bud.check_abs(what + " (north star 1e-3 vs the reference's fp32 result)", rel_max(mine, g[key]), 1e-3)
    _argmax_budget(bud, out["left_seg"], g, g64, sl)
    bud.note("anchors", free_run_drew_the_reference_anchors=bool(same))
    if not same:
        if anchor_y is None:                        # round-2 fixture: classes of the anchors = labels at their pixels
            lab = torch.nn.functional.interpolate(labels.float().unsqueeze(1), (h // 4, w // 4), mode="nearest").long()
            anchor_y = lab.reshape(b, -1)[torch.from_numpy(g["anchor_img"]).long(),
                                          torch.from_numpy(g["anchor_pix"][:, 0]).long()].numpy().astype(np.float32)
        del ts, out
        ts, out, same, cls = run((g["anchor_img"], g["anchor_pix"], anchor_y))
        assert same
    if anchor_y is not None:
        assert np.array_equal(cls, anchor_y)
    # losses: north-star tolerance (1e-3) against the float64 anchor; the ratio to the reference's own fp32 error is
    # recorded (one scalar is one draw of a heavy-tailed ratio: no K bound on it)
    for k in ("total", "supcon", "pixel", "seg"):
        err = abs(float(out[k].detach()) - float(g64[k])) / abs(float(g64[k]))
        bud.check_abs("loss " + k, err, 1e-3)
        bud.note("loss " + k, err_hip=err, err_ref32=max(float(e32(k)), abs(float(g[k]) - float(g64[k])) / abs(float(g64[k]))))
    params = dict(ts.model.named_parameters())
    names = [str(s) for s in g["grad_names"]]
    e32n = e32("grad_norms")
    worst_n = float(e32n.max())                       # norms are single numbers: see step_check.run_and_check_step
    worst_abs = 0.0
    for i, (k, n) in enumerate(zip(names, g["grad_norms"])):
        gn = float(params[k].grad.double().norm())
        bud.check("|grad| " + k, gn, float(n), float(g64["grad_norms"][i]), metric=rel_max, floor=worst_n)
        worst_abs = max(worst_abs, abs(gn - float(g64["grad_norms"][i])) / max(float(g64["grad_norms"][i]), 1e-30))
    if well:
        bud.check_abs("worst gradient norm error (absolute bound 1e-2)", worst_abs, 1e-2)
    for key in g.files:
        if key.startswith("grad::"):
            gr = params[key[6:]].grad
            mine = gr if gr.numel() < 400000 else gr.flatten()[::37]
            bud.family("stored gradient tensors", key, mine, g[key], g64[key], e32=float(e32(key)))
    sd = ts.model.state_dict()
    worst_rs = float(e32("rs_norms").max())
    # b2_128x256 only: the ASPP pooling branch normalises TWO values per channel (nn.BatchNorm2d on [2,256,1,1]); where they
    # nearly coincide x_hat = d / sqrt(d^2 + eps) amplifies fp32 noise of the pooled features up to 316x, and the branch
    # feeds aspp.project and everything behind it.  Measured (tools/aspp_project_probe.py, profiles/r03_aspp_probe.txt): the
    # projection kernel itself reproduces the per-channel mean of a float64 projection of ITS OWN inputs to 2e-9, while
    # those inputs differ by 4e-3 between the split-bf16 and the exact-fp32 kernels -- so the running statistics behind
    # that branch are held to K x the reference's fp32 error on the quantity behind the same amplification (the logits),
    # all others to K x the reference's worst running-statistics error.  b4_256x512 normalises FOUR values per channel in
    # that branch: the amplification is smaller but not gone -- measured: with bitwise identical stem outputs, merely
    # another grouping of the stem BatchNorm's partial sums (stem7_h2_kernel's 8 x 32-pixel tiles instead of 128-pixel runs:
    # a 1e-8 perturbation of bn1's batch statistics) moves |aspp.project.1.running_mean| from 1.8e-6 to 3.1e-6 off the
    # float64 anchor, the exact-fp32 kernels sit at < 5e-7, the reference's own fp32 run at 6.5e-8: single draws of a
    # noise-amplifying quantity, so the same floor applies on both fixtures.
    behind_pool = ("classifier.aspp.project.1.", "classifier.classifier.1.")
    for i, (k, n) in enumerate(zip([str(s) for s in g["rs_names"]], g["rs_norms"])):
        floor = max(worst_rs, 1e-6)
        if k.startswith(behind_pool):
            floor = max(floor, float(e32("before")))
        bud.check("|running| " + k, float(sd[k].double().norm()), float(n), float(g64["rs_norms"][i]), metric=rel_max,
                  floor=floor)
    bud.finish_family("stored gradient tensors")
    bud.finish()


def test_deeplab_eval_forward_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "deeplab_eval_b1_104x168.npz"), allow_pickle=False)
    g64 = np.load(os.path.join(golden_dir, "deeplab_eval_b1_104x168.f64.npz"), allow_pickle=False)
    ts = build(1, None)
    ts.model.eval()
    img = O.synthetic_batch(1, 104, 168, seed=52)[0]
    with torch.no_grad():
        seg, before, ff, ff0 = ts.model(img.to(DEV))
    bud = Budget("gpu_deeplab_eval_b1_104x168")
    bud.check("before", before, g["before"], g64["before"], metric=rel_max)
    bud.check("fine_feat", ff[:, ::8], g["fine_feat_sub"], g64["fine_feat_sub"], metric=rel_max)
    bud.check("fine_feat0", ff0[:, ::16], g["fine_feat0_sub"], g64["fine_feat0_sub"], metric=rel_max)
    am = seg.argmax(1).cpu().numpy().astype(np.uint8)
    bad = am != g["seg_argmax"]
    n_ref = int((g["seg_argmax"] != g64["seg_argmax"]).sum())
    e32_abs = float(np.abs(g["before"].astype(np.float64) - g64["before"]).max())
    worst = float(g["seg_margin"].astype(np.float64)[bad].max()) if bad.any() else 0.0
    bud.note("argmax", mismatches_hip_vs_ref32=int(bad.sum()), mismatches_ref32_vs_ref64=n_ref, pixels=int(bad.size))
    bud.check_abs("argmax mismatch count", int(bad.sum()), K * n_ref + 8)
    bud.check_abs("argmax worst reference margin at a mismatch", worst, 2 * K * e32_abs * 1.002)
    bud.finish()


def test_deeplab_step_matches_oracle_and_trains(golden_dir):
    """A second input (odd size 160x288) against the oracle, on the well-conditioned state (residual gain 0.25).  Two
    fp32 evaluations are compared with each other: the north star's 1e-3 on features / logits, 1e-4 on the total loss and
    1e-2 on the worst gradient norm, all in absolute terms (round 2: 17 % on the ill-conditioned state; the reference's
    own fp32-vs-fp64 figures on this state are 2e-5 / 2e-3, tests/golden/deeplab_step_b4_256x512.f64.npz).  If the free run's sampler is redirected by an argmax near-tie, the oracle's anchors are injected."""
    b, h, w = 2, 160, 288
    img, labels, ldw, weather, cw = O.synthetic_batch(b, h, w, seed=81, two_crops=True, cell=32)
    state, proj = D.make_state(seed=7, residual_gain=0.25), O.make_proj(seed=9, dim_in=2048)
    ref, grads, _ = oracle_deeplab_step(state, proj, img, labels.clone(), ldw, weather, cw, b, 3)
    a_img, a_pix, a_cls = ref["anchors"]
    bud = Budget("gpu_deeplab_oracle_160x288")

    def run(forced=None):
        ts_ = build(b, cw, residual_gain=0.25)
        ts_.pixelcontrast_criterion.forced_anchors = forced
        s0 = dict(left=img[:b], label=labels.clone(), weather=weather, label_distance_weight=ldw)
        torch.manual_seed(3)
        out_ = ts_.step((s0, dict(left=img[b:])), do_optimizer_step=False)
        img_i, cls, pix, n_view = ts_.pixelcontrast_criterion.last_anchors
        same = np.array_equal(np.asarray(img_i), a_img) and np.array_equal(pix.cpu().numpy().T, a_pix)
        return ts_, out_, same

    ts, out, same = run()
    bud.note("anchors", free_run_drew_the_oracle_anchors=bool(same))
    if not same:
        ts, out, same = run((a_img, a_pix, a_cls))
        assert same
    bud.check_abs("total", abs(float(out["total"].detach()) - float(ref["total"])) / abs(float(ref["total"])),
                  1e-4)
    bud.check_abs("fine_feat", rel(out["fine_feat"], ref["fine_feat"].numpy()), 1e-3)
    bud.check_abs("seg logits", rel(out["left_seg"], ref["seg_logits"].numpy()), 1e-3)
    params = dict(ts.model.named_parameters())
    worst = max(abs(float(params[k].grad.norm()) - float(gr.norm())) / max(float(gr.norm()), 1e-9) for k, gr in grads.items())
    bud.check_abs("worst gradient norm", worst, 1e-2)
    bud.finish()
    losses = []
    ts.model._get_engine().dropout_noise = None            # device-side dropout mask from here on
    for it in range(3):
        s0 = dict(left=img[:b], label=labels.clone(), weather=weather, label_distance_weight=ldw)
        losses.append(float(ts.step((s0, dict(left=img[b:])))["total"].detach()))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_lazy_fine_feat0_same_step_less_memory():
    """Rows interpolated on demand (dcs_gather_rows_bilinear / dcs_scatter_rows_bilinear) against the materialised
    [B,2048,h,w] tensor: identical sampled anchors and losses, gradients to fp32 summation order, lower peak memory."""
    b, h, w = 2, 256, 512
    img, labels, ldw, weather, cw = O.synthetic_batch(b, h, w, seed=83, two_crops=True, cell=32)
    res = []
    for lazy in (False, True):
        torch.manual_seed(0)                       # same random weather-classifier head in both builds
        ts = build(b, cw, lazy)
        s0 = dict(left=img[:b], label=labels.clone(), weather=weather, label_distance_weight=ldw)
        torch.manual_seed(3)
        torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        out = ts.step((s0, dict(left=img[b:])), do_optimizer_step=False)
        torch.cuda.synchronize()
        peak = torch.cuda.max_memory_allocated() - base          # what the step itself needed
        keep = {k: out[k].detach().clone() for k in ("pixel", "total", "pred_weather")}
        res.append((keep, {k: p.grad.clone() for k, p in ts.model.named_parameters()}, peak,
                    ts.pixelcontrast_criterion.last_anchors[2].clone()))
        del ts, out
    (o0, g0, m0, a0), (o1, g1, m1, a1) = res
    assert torch.equal(a0, a1)
    assert float(o0["pixel"]) == float(o1["pixel"]) and abs(float(o0["total"]) - float(o1["total"])) < 1e-5 * abs(float(o0["total"]))
    assert rel(o1["pred_weather"], o0["pred_weather"].detach().cpu().numpy()) < 1e-5
    worst = max(float((g1[k] - g0[k]).norm()) / max(float(g0[k].norm()), 1e-12) for k in g0)
    assert worst < 1e-4, worst
    assert m1 < m0 - 2 * b * 2048 * (h // 4) * (w // 4) * 4 * 0.9, (m0, m1)    # feature + its gradient no longer allocated


def test_deeplab_single_image_batch_raises_like_the_reference():
    """nn.BatchNorm2d on the [B,256,1,1] ASPP pooling branch raises for B == 1 in training mode."""
    img, labels, ldw, weather, cw = O.synthetic_batch(1, 64, 128, seed=82, two_crops=True, cell=32)
    ts = build(1, cw)
    with pytest.raises(ValueError, match="more than 1 value per channel"):
        ts.model([img[:1].to(DEV), img[1:].to(DEV)], return_supcon_feature=True)
