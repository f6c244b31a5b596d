"""CPU: host logic of the product (graph wiring of the hand-written backward, autograd
Functions, sampler planning, criterion switch, optimizer groups) with every kernel replaced by
the torch emulation of its contract (tests/emu_ops.py), against the golden vectors produced by
the reference.  The real kernels are checked against the same contracts in the -m gpu tests."""
import os

import numpy as np
import pytest
import torch

import emu_ops
from oracle import swiftnet_oracle as O


@pytest.fixture()
def emu(monkeypatch):
    emu_ops.install(monkeypatch)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def close(a, b, rtol=2e-4):
    a = np.asarray(a.detach() if torch.is_tensor(a) else a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=rtol * max(np.abs(b).max(), 1e-30))


def build(criterion, batch_size=2):
    from dcs_amd.trainer import TrainStep, make_opts
    ts = TrainStep(make_opts(criterion=criterion, batch_size=batch_size), class_weight=None, device="cpu")
    state = O.make_state(seed=1)
    ts.model.load_state_dict(state, strict=True)
    assert list(ts.model.state_dict().keys()) == list(state.keys())
    proj = O.make_proj(seed=2)
    with torch.no_grad():
        p = ts.supcon_criterion.projection
        p[0].weight.copy_(proj[0]); p[0].bias.copy_(proj[1]); p[2].weight.copy_(proj[2]); p[2].bias.copy_(proj[3])
    return ts


CASES = [
    ("step_supcon_pixel_focal_b2_256x512.npz", "supcon_pixelcontrast_focal", dict(b=2, h=256, w=512, seed=10, two=True, cell=32), 123),
    ("step_pixel_focal_b2_200x328.npz", "pixelcontrast_focal", dict(b=2, h=200, w=328, seed=11, two=False, cell=24), 7),
    ("step_ce_b2_256x512.npz", "crossentropy", dict(b=2, h=256, w=512, seed=12, two=False, cell=32), 1),
]


@pytest.mark.parametrize("fname,criterion,shape,rng_seed", CASES)
def test_train_step_host_logic(emu, golden_dir, fname, criterion, shape, rng_seed):
    from step_check import load_anchor, run_and_check_step
    # gradients / BatchNorm buffers: K x the reference's own fp32-vs-fp64 error on this fixture (tests/budget.py);
    # exactness of the graph wiring itself is pinned at 1e-7 by test_engine_matches_oracle_fp64_odd_size below.
    run_and_check_step(build(criterion), load(golden_dir, fname), criterion, shape, rng_seed, rtol=5e-4,
                       g64=load_anchor(golden_dir, fname), name="cpu_emu_" + fname[:-4])


def test_eval_forward_host_logic(emu, golden_dir):
    g = load(golden_dir, "eval_fwd_b1_120x200.npz")
    ts = build("crossentropy")
    ts.model.eval()
    img = O.synthetic_batch(1, 120, 200, seed=13)[0]
    with torch.no_grad():
        seg, before, ff, ff0 = ts.model(img)
    close(before, g["before"])
    close(ff, g["fine_feat"])
    from step_check import check_argmax
    check_argmax(seg, g["seg_argmax"], 2e-4)


def test_product_refuses_cpu_tensors_without_emulation():
    """No CPU fallback: the un-patched product raises on CPU tensors."""
    from dcs_amd.trainer import TrainStep, make_opts
    ts = TrainStep(make_opts(criterion="crossentropy", batch_size=1), device="cpu")
    with pytest.raises(RuntimeError, match="no CPU path"):
        ts.model(torch.zeros(1, 3, 32, 64))


def test_state_dict_contract():
    from dcs_amd.model import WeatherNet
    from dcs_amd.trainer import make_opts
    m = WeatherNet(make_opts(), num_classes=19, backbone="resnet18")
    spec = O.state_spec()
    sd = m.state_dict()
    assert list(sd.keys()) == list(spec.keys())
    for k, (shape, _) in spec.items():
        assert tuple(sd[k].shape) == tuple(shape), k
    rnd = sum(p.numel() for p in m.random_init_params())
    fine = sum(p.numel() for p in m.fine_tune_params())
    assert rnd == 861440 and fine == 11176768
    assert sum(p.numel() for p in m.parameters()) == 12040915


@pytest.mark.parametrize("criterion,two", [("supcon_pixelcontrast_focal", True), ("supcon_simclr_focal", True),
                                           ("crossentropy", False)])
def test_engine_matches_oracle_fp64_odd_size(emu, criterion, two):
    """Exactness of the HOST logic, independent of fp32 conditioning: the hand-written forward/backward graph
    (kernels emulated in float64) must reproduce the oracle's float64 losses and gradients to ~1e-9 at a size
    that is not a multiple of 32 (odd maps, non-integer bilinear scales, 1-pixel deep maps)."""
    from dcs_amd.trainer import TrainStep, make_opts
    dt = torch.float64
    b, h, w = 2, 104, 184
    img, labels, ldw, weather, cw = O.synthetic_batch(b, h, w, seed=31, two_crops=two, cell=16)
    state = {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in O.make_state(seed=1).items()}
    proj = [p.to(dt) for p in O.make_proj(seed=2)]
    ts = TrainStep(make_opts(criterion=criterion, batch_size=b, dtype=dt, flat_params=False), class_weight=cw.to(dt),
                   device="cpu")          # flat buffers are fp32 views; this test casts the modules to float64
    ts.model.double(); ts.supcon_criterion.double(); ts.weather_clf.double()
    ts.model.load_state_dict(state, strict=True)
    with torch.no_grad():
        p = ts.supcon_criterion.projection
        p[0].weight.copy_(proj[0]); p[0].bias.copy_(proj[1]); p[2].weight.copy_(proj[2]); p[2].bias.copy_(proj[3])
    s0 = dict(left=img[:b].to(dt), label=labels.clone(), weather=weather, label_distance_weight=ldw.to(dt))
    sample = (s0, dict(left=img[b:].to(dt))) if two else s0
    torch.manual_seed(5)
    out = ts.step(sample, do_optimizer_step=False)
    torch.manual_seed(5)
    ref, grads, gproj = O.train_step(state, proj, None, img.to(dt), labels.clone(), ldw.to(dt), weather, cw.to(dt),
                                     criterion, b)
    assert abs(float(out["total"]) - float(ref["total"])) < 1e-9 * max(1.0, abs(float(ref["total"])))
    np.testing.assert_allclose(out["fine_feat"].detach().numpy(), ref["fine_feat"].numpy(), rtol=0, atol=1e-9)
    np.testing.assert_allclose(out["left_seg"].detach().numpy(), ref["seg_logits"].numpy(), rtol=0, atol=1e-9)
    params = dict(ts.model.named_parameters())
    for k, gref in grads.items():
        if gref is None:
            assert params[k].grad is None, k
            continue
        scale = max(float(gref.abs().max()), 1e-12)
        err = float((params[k].grad - gref).abs().max()) / scale
        assert err < 1e-7, (k, err)
    if gproj[0] is not None:
        pr = ts.supcon_criterion.projection
        for mine, r in zip([pr[0].weight.grad, pr[0].bias.grad, pr[2].weight.grad, pr[2].bias.grad], gproj):
            assert float((mine - r).abs().max()) / max(float(r.abs().max()), 1e-12) < 1e-7
    # BatchNorm buffers after forward + checkpoint-recompute replay
    sd = ts.model.state_dict()
    for k, v in state.items():
        if "running_" in k:
            assert float((sd[k] - v).abs().max()) < 1e-9, k
        if "num_batches" in k:
            assert int(sd[k]) == int(v), k


def test_second_backward_accumulates_on_the_flat_gradient_path(emu):
    """torch.autograd semantics the reference's loop relies on: without zero_grad(set_to_none=True) a second backward
    ADDS to .grad.  The engine writes gradients straight into the flat buffer views, so it must notice that .grad
    already is that view (gradient accumulation over micro-batches / zero_grad(set_to_none=False))."""
    ts = build("focal", batch_size=2)
    assert ts.flat is not None
    img, labels, ldw, weather, cw = O.synthetic_batch(2, 128, 256, seed=3, two_crops=False, cell=32)
    ts.model.eval()                                   # deterministic BatchNorm: both passes see the same function

    def run():
        seg, _, _, _ = ts.model(img)
        loss = ts.criterion(seg, labels.clone(), dict(label_distance_weight=ldw))
        loss.backward()

    run()
    params = [p for p in ts.model.parameters() if p.grad is not None]
    assert params and all(p.grad.data_ptr() == ts.flat.grad_view[p].data_ptr() for p in params if p in ts.flat.grad_view)
    once = [p.grad.detach().clone() for p in params]
    run()                                             # no zero_grad in between
    for p, g1 in zip(params, once):
        assert torch.allclose(p.grad, 2 * g1, rtol=1e-5, atol=1e-9)
    ts.model.zero_grad(set_to_none=False)             # grads zeroed in place, .grad still the flat view
    run()
    for p, g1 in zip(params, once):
        assert torch.allclose(p.grad, g1, rtol=1e-5, atol=1e-9)
    ts.model.zero_grad()                              # set_to_none=True: the fast path writes in place again
    run()
    for p, g1 in zip(params, once):
        assert torch.equal(p.grad, g1) and p.grad.data_ptr() == ts.flat.grad_view[p].data_ptr()


@pytest.mark.parametrize("n", [0, 1, 2, 7, 1000, 4097, 70000, 131072])
def test_sampler_randperm_dtype_does_not_change_the_reference_stream(n):
    """The sampler's host plan draws torch.randperm(n, dtype=int32) where the reference draws torch.randperm(n)
    (utils/loss.py:314-325): same permutation, same generator state afterwards (the CPU generator is consumed per
    element whatever the output type), so every later draw of the step -- and of the run -- is the reference's."""
    torch.manual_seed(1234 + n)
    a = torch.randperm(n)
    sa = torch.get_rng_state().clone()
    torch.manual_seed(1234 + n)
    b = torch.randperm(n, dtype=torch.int32)
    sb = torch.get_rng_state().clone()
    assert torch.equal(a, b.long())
    assert torch.equal(sa, sb)


def _random_counts(rng, B, C, big):
    cnt = torch.zeros((B, C, 2), dtype=torch.int64)
    for b in range(B):
        for c in range(C):
            kind = rng.integers(0, 6)
            if kind == 0:
                continue                                             # class absent
            if kind == 1:
                cnt[b, c, rng.integers(0, 2)] = int(rng.integers(1, 4))       # one, two or three pixels on one side
            elif kind == 2:
                cnt[b, c, 0] = int(rng.integers(0, 3)); cnt[b, c, 1] = int(rng.integers(0, 3))
            else:
                cnt[b, c, 0] = int(rng.integers(0, big)); cnt[b, c, 1] = int(rng.integers(0, big))
    return cnt


@pytest.mark.parametrize("seed,B,C,big,max_samples", [(0, 1, 5, 50, 1024), (1, 2, 19, 3000, 1024), (2, 4, 19, 140000, 1024),
                                                      (3, 16, 19, 20000, 1024), (4, 3, 19, 700, 40), (5, 8, 19, 1300, 100),
                                                      (6, 2, 7, 2000, 1024)])
def test_library_sampler_plan_is_the_torch_plan(seed, B, C, big, max_samples):
    """dcs_sampler_plan (csrc/sampler_host.cpp: sparse Fisher-Yates prefix + generator skip on the state of torch's CPU
    generator) against the statement in torch (_plan_anchor_requests: torch.randperm per kept class, utils/loss.py:264-337):
    same plan, and the SAME generator afterwards -- state bytes that matter and the next draws -- over absent classes,
    classes with one to three pixels, empty hard or easy sides, 1.4e5-pixel classes and more classes than samples."""
    from dcs_amd import losses
    rng = np.random.default_rng(seed)
    for rep in range(6):
        cnt = _random_counts(rng, B, C, big)
        torch.manual_seed(100 * seed + rep)
        torch.rand(rep * 211)                                         # start somewhere inside a state block
        s0 = torch.get_rng_state().clone()
        try:
            want = losses._plan_anchor_requests(cnt, C, max_samples, 2)
            err = None
        except Exception as e:                                        # the reference raises on some count patterns
            want, err = None, e
        after_want = torch.rand(5)
        torch.set_rng_state(s0)
        got = losses._plan_in_library(cnt, C, max_samples, 2)
        if got is NotImplemented:
            assert torch.equal(torch.get_rng_state(), s0)             # generator untouched: the general path takes over
            n_classes = int(((cnt.sum(-1)) > 2).sum())
            assert err is not None or (n_classes > 0 and max_samples // n_classes == 0), (seed, rep)
            continue
        assert err is None
        assert got == want, (seed, rep)
        assert torch.equal(torch.rand(5), after_want), (seed, rep)


def test_library_sampler_plan_is_what_the_criterion_uses(monkeypatch):
    from dcs_amd import losses
    cnt = torch.tensor([[[5, 9], [0, 0], [1, 7]], [[3000, 2], [2, 1], [40, 50]]])
    torch.manual_seed(3)
    a = losses.plan_anchor_requests(cnt, 3, 1024, 2)
    sa = torch.rand(3)
    monkeypatch.setenv("DCS_SAMPLER_PLAN", "torch")
    torch.manual_seed(3)
    b = losses.plan_anchor_requests(cnt, 3, 1024, 2)
    assert a == b and torch.equal(sa, torch.rand(3))
    assert a[0] == 2 and a[1] == 5 and len(a[2]) == 10


@pytest.mark.parametrize("criterion,two", [("supcon_pixelcontrast_focal", True), ("pixelcontrast_focal", False)])
def test_pixel_contrast_row_gradients_equal_the_dense_gradient(emu, monkeypatch, criterion, two):
    """losses._row_sink: the pixel-contrast gradient reaches the network as <= 608 rows handed to the autograd node of
    fine_feat0 (a memory-less zero goes through autograd) instead of a dense tensor that is zero almost everywhere.  Same
    step, same gradients as the dense route (DCS_SPARSE_FF0=0); and the sink is really taken."""
    from dcs_amd import losses, model
    b = 2
    img, labels, ldw, weather, cw = O.synthetic_batch(b, 128, 256, seed=21, two_crops=two, cell=32)
    taken = []
    orig = model._SwiftNetFn.accept_rows
    monkeypatch.setattr(model._SwiftNetFn, "accept_rows", staticmethod(lambda ctx, r, x: (taken.append(int(r.numel())), orig(ctx, r, x))[1]))
    res = []
    for mode in ("1", "0"):
        monkeypatch.setenv("DCS_SPARSE_FF0", mode)
        ts = build(criterion, batch_size=b)
        s0 = dict(left=img[:b], label=labels.clone(), weather=weather, label_distance_weight=ldw)
        torch.manual_seed(9)
        out = ts.step((s0, dict(left=img[b:])) if two else s0, do_optimizer_step=False)
        res.append((float(out["total"]), {k: p.grad.clone() for k, p in ts.model.named_parameters() if p.grad is not None}))
    assert len(taken) == 1 and taken[0] > 0                         # only the first run handed rows over
    assert res[0][0] == res[1][0]
    assert res[0][1].keys() == res[1][1].keys() and len(res[0][1]) > 60
    for k in res[0][1]:
        assert torch.allclose(res[0][1][k], res[1][1][k], rtol=1e-6, atol=1e-9), k
