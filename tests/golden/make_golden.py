#!/usr/bin/env python3
"""Generate golden vectors by RUNNING THE REFERENCE in the build container.

Run here only (``/root/reference`` does not exist on the GPU box):
    python tests/golden/make_golden.py

The reference's package ``__init__`` files import torchvision/cv2/tensorboard
(absent here), so the two packages are pre-registered as empty namespace
modules and only the hot-path modules are imported.  The
reference code itself is executed unmodified; only ``pretrained=True`` (a
network fetch) is replaced by ``pretrained=False``.  Weights and inputs come
from platform-stable numpy streams (oracle.swiftnet_oracle.make_state /
synthetic_batch), so the fixtures hold only expected OUTPUTS plus the seeds.

Fixtures written: tests/golden/*.npz (data only, no reference source).

Float64 anchors (``<fixture>.f64.npz``).  Every train-step fixture is produced TWICE by the same reference modules:
in float32 (the fixture proper) and in float64 (``model.double()``, float64 inputs).  The float64 run is the
numerical truth of the reference's function on these inputs; ``|ref32 - ref64|`` is the reference's OWN float32
error on each stored quantity and is the unit in which the GPU tests budget the HIP path's error
(tests/step_check.py: ``err(HIP fp32 vs ref fp64) <= K * err(ref fp32 vs ref fp64)``).  Two reference functions
hard-code float32 (utils/loss.py:60 ``log_softmax(input.to(torch.float32))``, :292/:410 the anchor sampler), so in
the float64 run the focal loss is the unmodified module (its internal fp32 log-softmax contributes ~1e-7) and the
pixel loss is the reference's ``_contrastive`` (no cast inside) on float64 anchors gathered at the pixel indices the
reference's own sampler drew in the float32 run (recovered by re-running ``_hard_anchor_sampling`` on an
index-encoding tensor under the same RNG state).  Float64 results are stored rounded to float32 (6e-8 relative,
far below every budget) except the loss scalars.
"""
import importlib
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

warnings.filterwarnings("ignore")


def import_reference():
    sys.path.insert(0, REF)
    for pkg in ("network", "utils"):
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF, pkg)]
        sys.modules[pkg] = m
    mods = types.SimpleNamespace()
    mods.loss = importlib.import_module("utils.loss")
    mods.weathernet = importlib.import_module("network.weathernet")
    mods.classifier = importlib.import_module("network.classifier")
    rp = importlib.import_module("network.backbone.resnet_pyramid")
    orig = rp.resnet18_pyramid
    mods.weathernet.resnet18_pyramid = lambda pretrained=True, **kw: orig(pretrained=False, **kw)
    return mods


def make_opts(criterion):
    return types.SimpleNamespace(deeplab=False, criterion=criterion, with_depth_level_loss=False,
                                 no_class_weights=False, no_EDT=False, batch_size=2)


def build_ref_model(mods, opts, state):
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        model = mods.weathernet.WeatherNet(opts, num_classes=19, device=torch.device("cpu"),
                                           backbone="resnet18", train_semantic=True)
    # strict=False like the reference's own restore (utils/init_trainer.py:246-279): its custom
    # _load_from_state_dict (resnet_pyramid.py:381-393) also looks for ImageNet-style "bn1.*" keys.
    missing, unexpected = model.load_state_dict({k: v.clone() for k, v in state.items()}, strict=False)
    assert not unexpected and all(".bn1." in "." + k and "feature_extractor.bn1." in k for k in missing), missing
    sd = model.state_dict()
    assert list(sd.keys()) == list(state.keys())
    assert all(torch.equal(sd[k], state[k]) for k in state)
    return model


def pixel_loss(pixc, fine_feat0, labels, before, captured, plan):
    """float32 (plan is None): the reference's PixelContrastLoss.forward, unmodified, with two spies that record the
    sampled anchors and -- by running the reference sampler a second time on an index-encoding tensor under the same
    RNG state -- which pixels they are.  float64 (plan given): utils/loss.py:339-389 on float64 rows of the same pixels."""
    import contextlib, io
    if plan is not None:
        img_i, pix_i, y_ = plan
        b, c, h, w = fine_feat0.shape
        flat = fine_feat0.permute(0, 2, 3, 1).reshape(b, h * w, c)
        X_ = torch.stack([flat[img_i, pix_i[:, v]] for v in range(pix_i.shape[1])], dim=1)      # [T, n_view, C]
        captured["x_"] = X_.detach().clone(); captured["y_"] = y_.clone()
        return pixc._contrastive(X_, y_.to(X_.dtype))
    orig_c, orig_s = pixc._contrastive, pixc._hard_anchor_sampling
    def spy_c(feats_, labels_):
        captured["x_"] = feats_.detach().clone(); captured["y_"] = labels_.detach().clone()
        return orig_c(feats_, labels_)
    def spy_s(X, y_hat, y):
        st = torch.get_rng_state()
        out = orig_s(X, y_hat, y)
        st2 = torch.get_rng_state()
        torch.set_rng_state(st)
        code = torch.zeros_like(X)
        code[:, :, 0] = torch.arange(X.shape[1], dtype=X.dtype)[None, :]
        code[:, :, 1] = torch.arange(X.shape[0], dtype=X.dtype)[:, None]
        Xi, yi = orig_s(code, y_hat, y)
        torch.set_rng_state(st2)
        assert torch.equal(yi, out[1])
        captured["pix"] = Xi[:, :, 0].long(); captured["img"] = Xi[:, 0, 1].long()
        assert torch.equal(X[captured["img"], captured["pix"][:, 0]], out[0][:, 0])
        return out
    pixc._contrastive, pixc._hard_anchor_sampling = spy_c, spy_s
    with contextlib.redirect_stdout(io.StringIO()):
        return pixc(fine_feat0, labels=labels, predict=before)


def np_(t):
    return t.detach().cpu().numpy().copy()


def ref_train_step(mods, state, proj, batch, criterion, batch_size, rng_seed, lr=4e-4, wd=1e-4, dtype=torch.float32,
                   plan=None, argmax_stride=1, sub=(2, 4, 8), store_labels=True, variant=None):
    """trainer.py:62-215 replayed by hand around the imported reference modules.
    dtype=float64: the float64 anchor run (see the module docstring); ``plan`` = (img [T], pix [T,n_view], y [T]) of the
    float32 run's sampled anchors.  Returns (res, plan)."""
    import contextlib, io
    img, labels, ldw, weather, cw = batch
    f64 = dtype == torch.float64
    img, ldw = img.to(dtype), ldw.to(dtype)
    opts = make_opts(criterion)
    opts.batch_size = batch_size
    dev = torch.device("cpu")
    model = build_ref_model(mods, opts, state).to(dtype)
    if variant == "channels_last":                      # another float32 execution path of the same program (oneDNN nhwc kernels)
        model = model.to(memory_format=torch.channels_last)
        img = img.contiguous(memory_format=torch.channels_last)
    model.train()
    crit = mods.loss.BoundaryAwareFocalLoss(gamma=0.5, num_classes=19, ignore_id=255, weight=cw,
                                            device=dev, opts=opts)
    supc = mods.loss.SupConLoss(temperature=0.07, contrast_mode="all", base_temperature=0.07,
                                weight=cw, device=dev, opts=opts)
    with torch.no_grad():
        supc.projection[0].weight.copy_(proj[0]); supc.projection[0].bias.copy_(proj[1])
        supc.projection[2].weight.copy_(proj[2]); supc.projection[2].bias.copy_(proj[3])
    supc = supc.to(dtype)
    pixc = mods.loss.PixelContrastLoss(device=dev)
    ce_c = torch.nn.CrossEntropyLoss(weight=None, ignore_index=255)
    optim = torch.optim.Adam([
        {"params": model.random_init_params(), "lr": lr, "weight_decay": wd},
        {"params": model.fine_tune_params(), "lr": lr / 4, "weight_decay": wd / 4}], betas=(0.9, 0.99))

    labels = labels.clone()
    sample = {"label_distance_weight": ldw}
    torch.manual_seed(rng_seed)
    supcon_flag = "supcon" in criterion
    seg, before, fine_feat, fine_feat0 = model(img, return_supcon_feature=supcon_flag)
    res = {}
    zero = torch.tensor([0.])
    sup = pix = segl = ce = zero
    captured = {}
    if criterion == "supcon_pixelcontrast_focal":
        sup = supc(fine_feat, class_labels=weather, mask=None)
        pix = pixel_loss(pixc, fine_feat0, labels, before, captured, plan)
        segl = crit(seg, labels, sample)
        total = 1 / batch_size * (sup + pix) + segl * 1.2
    elif criterion == "pixelcontrast_focal":
        pix = pixel_loss(pixc, fine_feat0, labels, before, captured, plan)
        segl = crit(seg, labels, sample)
        total = pix * 1 / batch_size + segl * 1.2
    elif criterion == "crossentropy":
        ce = ce_c(seg, labels)
        total = ce
    else:
        raise KeyError(criterion)
    optim.zero_grad()
    supc.zero_grad()
    total.backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    proj_grads = [supc.projection[0].weight.grad, supc.projection[0].bias.grad,
                  supc.projection[2].weight.grad, supc.projection[2].bias.grad]
    optim.step()
    np32 = (lambda t: np_(t).astype(np.float32))            # float64 results are stored rounded to float32
    res.update(total=np_(total).reshape(()), supcon=np_(sup).reshape(()), pixel=np_(pix).reshape(()),
               seg=np_(segl).reshape(()), ce=np_(ce).reshape(()))
    # outputs are stored spatially subsampled to keep the fixtures small (argmax is kept in full unless argmax_stride)
    res["before_sub"] = np32(before[:, :, ::sub[0], ::sub[0]])
    res["fine_feat_sub"] = np32(fine_feat[:, :, ::sub[1], ::sub[1]])
    res["seg_logits_sub"] = np32(seg[:, :, ::sub[2], ::sub[2]])
    if tuple(sub) != (2, 4, 8):
        res["sub_strides"] = np.array(sub, dtype=np.int32)
    a = argmax_stride
    res["seg_argmax"] = np_(seg[:, :, ::a, ::a].argmax(1)).astype(np.uint8)
    # logit margin (best - second best) at every stored argmax pixel: lets a test tell a numerical near-tie from an error
    top2 = seg[:, :, ::a, ::a].detach().topk(2, dim=1)[0]
    res["seg_margin"] = np_((top2[:, 0] - top2[:, 1])).astype(np.float16)
    res["seg_absmax"] = np.float64(seg.detach().abs().max())
    if store_labels:
        res["labels_after"] = np_(labels).astype(np.int16)
    if captured:
        res["anchor_x"] = np32(captured["x_"]); res["anchor_y"] = np32(captured["y_"])
        if "pix" in captured:
            res["anchor_pix"] = np_(captured["pix"]).astype(np.int32); res["anchor_img"] = np_(captured["img"]).astype(np.int32)
            plan = (captured["img"], captured["pix"], captured["y_"])
    full = ("feature_extractor.conv1.weight", "feature_extractor.layer1.0.conv1.weight",
            "feature_extractor.layer2.0.downsample.0.weight", "feature_extractor.upsample_bottlenecks1.weight",
            "feature_extractor.upsample_blends5.blend_conv.conv.weight",
            "feature_extractor.upsample_blends1.blend_conv.norm.weight",
            "feature_extractor.layer4.1.bn2.bias", "feature_extractor.bn1_2.weight",
            "segmentation.conv.weight", "segmentation.conv.bias", "segmentation.norm.weight")
    names = [k for k, _ in model.named_parameters()]
    res["grad_names"] = np.array(names)
    res["grad_norms"] = np.array([float(grads[k].norm()) if k in grads else -1.0 for k in names], dtype=np.float64)
    res["grad_sums"] = np.array([float(grads[k].double().sum()) if k in grads else 0.0 for k in names], dtype=np.float64)
    for k in full:
        if k in grads:
            res["grad::" + k] = np32(grads[k])
    if proj_grads[0] is not None:
        for i, g in enumerate(proj_grads):
            res[f"proj_grad_{i}"] = np32(g)
    sd = model.state_dict()
    res["post_names"] = np.array(list(sd.keys()))
    res["post_norms"] = np.array([float(v.double().norm()) for v in sd.values()], dtype=np.float64)
    for k, v in sd.items():
        if "running_" in k:
            res["post::" + k] = np32(v)
        elif "num_batches" in k:
            res["post::" + k] = np_(v)
    for k in full:
        res["post::" + k] = np32(sd[k])
    return res, plan


def _rel_l2(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))


def _rel_max(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def fp32_errors(r, r64):
    """The reference's float32 error on every budgeted quantity of a fixture: {key: error of r vs r64}."""
    e = {}
    for k in r:
        if k.startswith("grad::") or k.startswith("proj_grad_"):
            e[k] = _rel_l2(r[k], r64[k])
        elif k.startswith("post::") and "running_" in k:
            e[k] = _rel_max(r[k], r64[k])
        elif k in ("before_sub", "fine_feat_sub", "seg_logits_sub"):
            e[k] = _rel_max(r[k], r64[k])
        elif k in ("total", "supcon", "pixel", "seg", "ce") and float(r64[k]) != 0.0:
            e[k] = abs(float(r[k]) - float(r64[k])) / abs(float(r64[k]))
    n32, n64 = r["grad_norms"], r64["grad_norms"]
    e["grad_norms"] = np.where(n64 > 0, np.abs(n32 - n64) / np.maximum(n64, 1e-300), 0.0)
    return e


VARIANTS = ("nomkldnn", "channels_last")


def step_fixture(mods, name, state, proj, batch, criterion, batch_size, rng_seed, argmax_stride=1, reuse=False, **kw):
    """float32 fixture <name>.npz + float64 anchor <name>.f64.npz.

    The anchor file also carries ``e32::<key>``: the reference's OWN float32 error on every budgeted quantity, as the
    maximum over its float32 execution paths available here -- the fixture's run (oneDNN, 8 threads), the same program
    with oneDNN disabled (ATen's native convolution) and in channels_last memory format (oneDNN nhwc kernels).  One
    float32 run is a single draw of the rounding error; on the small fixtures (BatchNorm over 8..64 samples) draws of
    equally valid evaluations differ by factors, so the budget unit is the largest of the three.
    reuse=True: keep the existing <name>.npz / .f64.npz values and only (re)measure the variants."""
    f32_path, f64_path = os.path.join(HERE, name + ".npz"), os.path.join(HERE, name + ".f64.npz")
    drop = ("grad_names", "post_names", "labels_after", "anchor_y")
    if reuse:
        r = dict(np.load(f32_path, allow_pickle=False))
        r64 = {k: v for k, v in np.load(f64_path, allow_pickle=False).items() if not k.startswith("e32::")}
        plan = (torch.from_numpy(r["anchor_img"]).long(), torch.from_numpy(r["anchor_pix"]).long(),
                torch.from_numpy(r["anchor_y"])) if "anchor_pix" in r else None
    else:
        r, plan = ref_train_step(mods, state, proj, batch, criterion, batch_size, rng_seed, argmax_stride=argmax_stride, **kw)
        np.savez_compressed(f32_path, **r)
        r64, _ = ref_train_step(mods, state, proj, batch, criterion, batch_size, rng_seed, dtype=torch.float64, plan=plan,
                                argmax_stride=argmax_stride, **kw)
        r64 = {k: v for k, v in r64.items() if k not in drop}
    e32 = fp32_errors(r, r64)
    used = ["fixture"]
    for v in VARIANTS:
        import contextlib
        ctx = torch.backends.mkldnn.flags(enabled=False) if v == "nomkldnn" else contextlib.nullcontext()
        with ctx:
            rv, pv = ref_train_step(mods, state, proj, batch, criterion, batch_size, rng_seed, argmax_stride=argmax_stride,
                                    variant=v, **kw)
        if plan is not None and not (torch.equal(pv[0], plan[0]) and torch.equal(pv[1], plan[1])):
            print(f"  variant {v}: samples other anchors than the fixture (an argmax near-tie) -- not used", flush=True)
            continue
        ev = fp32_errors(rv, r64)
        e32 = {k: np.maximum(e32[k], ev[k]) for k in e32}
        used.append(v)
        print(f"  variant {v}: max grad err {max(float(np.max(ev[k])) for k in ev if k.startswith('grad::')):.2e}", flush=True)
    out = dict(r64)
    for k, v in e32.items():
        out["e32::" + k] = np.asarray(v, dtype=np.float64)
    out["e32_variants"] = np.array(used)
    np.savez_compressed(f64_path, **out)
    rel = lambda a, b: float(np.linalg.norm((a.astype(np.float64) - b).ravel()) / max(np.linalg.norm(b.astype(np.float64).ravel()), 1e-300))
    rep = {k: float(r[k]) for k in ("total", "supcon", "pixel", "seg", "ce")}
    rep["loss_err32"] = max(abs(float(r[k]) - float(r64[k])) / max(abs(float(r64[k])), 1e-30) for k in ("total", "supcon", "pixel", "seg", "ce") if float(r64[k]) != 0)
    rep["logit_err32"] = float(np.abs(r["seg_logits_sub"].astype(np.float64) - r64["seg_logits_sub"]).max() / np.abs(r64["seg_logits_sub"]).max())
    rep["argmax_mismatch_32_vs_64"] = int((r["seg_argmax"] != r64["seg_argmax"]).sum())
    gk = [k for k in r if k.startswith("grad::")]
    rep["grad_err32_max"] = max(rel(r[k], r64[k]) for k in gk)
    n32, n64 = r["grad_norms"], r64["grad_norms"]
    rep["gradnorm_err32_max"] = float(np.max(np.abs(n32 - n64)[n64 > 0] / n64[n64 > 0]))
    print(name, rep, flush=True)
    return r, r64


def main():
    from oracle import swiftnet_oracle as O
    mods = import_reference()
    torch.set_num_threads(8)
    out = {}

    # Sizes are chosen so that the deepest pyramid map (level 2, layer4 = H/128) still has >= 2x3 pixels:
    # with smaller inputs the deep BatchNorms normalise over 2-4 values and gradients become numerically
    # ill-conditioned (fp32 vs fp64 runs of the reference itself then differ by several percent).
    # ---- G1: doubly-contrastive train step, B=2 (Bm=4), 256x512 ---------------
    state = O.make_state(seed=1)
    proj = O.make_proj(seed=2)
    only = set(a for a in sys.argv[1:] if a != "--reuse")
    reuse = "--reuse" in sys.argv                    # keep stored fp32 / fp64 values, only re-measure the fp32 variants
    if not only or "G1" in only:
        batch = O.synthetic_batch(2, 256, 512, seed=10, two_crops=True, cell=32)
        step_fixture(mods, "step_supcon_pixel_focal_b2_256x512", state, proj, batch, "supcon_pixelcontrast_focal", 2, 123, reuse=reuse)

    # ---- G2: pixelcontrast_focal, B=2, 200x328 (not a multiple of 32: odd maps everywhere) ---
    if not only or "G2" in only:
        batch = O.synthetic_batch(2, 200, 328, seed=11, two_crops=False, cell=24)
        step_fixture(mods, "step_pixel_focal_b2_200x328", state, proj, batch, "pixelcontrast_focal", 2, 7, reuse=reuse)

    # ---- G3: CE-only (config 1 miniature), B=2, 256x512 -----------------------
    if not only or "G3" in only:
        batch = O.synthetic_batch(2, 256, 512, seed=12, two_crops=False, cell=32)
        step_fixture(mods, "step_ce_b2_256x512", state, proj, batch, "crossentropy", 2, 1, reuse=reuse)

    # ---- G6: WELL-CONDITIONED gradients: B=8 at 1024x2048, pixelcontrast_focal.  The coarsest pyramid map (level 2,
    #          layer4 = H/128 x W/128 = 8x16) then has 8*8*16 = 1024 samples per channel in its BatchNorms (the other
    #          fixtures: 8..32), so fp32 gradients resolve to ~1e-3 and the test holds the HIP path to 1e-2 absolute.
    if not only or "G6" in only:
        batch = O.synthetic_batch(8, 1024, 2048, seed=14, two_crops=False, cell=64)
        step_fixture(mods, "step_pixel_focal_b8_1024x2048", state, proj, batch, "pixelcontrast_focal", 8, 5, argmax_stride=4,
                     sub=(8, 16, 32), store_labels=False, reuse=reuse)    # outputs subsampled harder: the fixture stays ~8 MB
    if only and not ({"G4", "G5"} & only):
        return

    # ---- G4: eval forward at a size that is NOT a multiple of 32 (validate path,
    #          trainer.py:342-349; default val size 1920x1080 has the same property); float32 + float64 anchor
    opts = make_opts("crossentropy")
    img = O.synthetic_batch(1, 120, 200, seed=13)[0]
    for dt, suffix in ((torch.float32, ""), (torch.float64, ".f64")):
        model = build_ref_model(mods, opts, state).to(dt)
        model.eval()
        with torch.no_grad():
            seg, before, ff, ff0 = model(img.to(dt))
        top2 = seg.topk(2, dim=1)[0]
        f32 = lambda t: np_(t).astype(np.float32)
        np.savez_compressed(os.path.join(HERE, "eval_fwd_b1_120x200%s.npz" % suffix),
                            before=f32(before), fine_feat=f32(ff), seg_argmax=np_(seg.argmax(1)).astype(np.uint8),
                            seg_logits_sub=f32(seg[:, :, ::4, ::4]), seg_margin=np_(top2[:, 0] - top2[:, 1]).astype(np.float16))
    print("G4 eval", tuple(seg.shape), tuple(before.shape))

    # ---- G5: loss unit vectors on synthetic features --------------------------
    g = np.random.default_rng(21)
    dev = torch.device("cpu")
    b, c, h, w = 3, 128, 12, 20
    feats = torch.from_numpy(g.standard_normal((b, c, h, w), dtype=np.float32))
    logits_lr = torch.from_numpy(g.standard_normal((b, 19, h, w), dtype=np.float32))
    labels = torch.from_numpy(g.integers(0, 6, size=(b, 4 * h, 4 * w)).astype(np.int64))
    labels[:, :3, :] = 255
    labels[0, 10:14, 10:14] = 17          # class with exactly 1 low-res pixel -> filtered (<= max_views)
    import contextlib, io
    pixc = mods.loss.PixelContrastLoss(device=dev)
    cap = {}
    orig_c = pixc._contrastive
    def spy(feats_, labels_):
        cap["x_"] = feats_.detach().clone(); cap["y_"] = labels_.detach().clone()
        return orig_c(feats_, labels_)
    pixc._contrastive = spy
    feats.requires_grad_(True)
    torch.manual_seed(99)
    with contextlib.redirect_stdout(io.StringIO()):
        pl = pixc(feats, labels=labels, predict=logits_lr)
    pl.backward()
    u = dict(pix_feats=np_(feats), pix_logits=np_(logits_lr), pix_labels=np_(labels).astype(np.int16),
             pix_loss=np_(pl).reshape(()), pix_anchor_x=np_(cap["x_"]), pix_anchor_y=np_(cap["y_"]),
             pix_grad_feats=np_(feats.grad))
    # supcon / simclr
    opts = make_opts("supcon_focal")
    supc = mods.loss.SupConLoss(device=dev, opts=opts)
    proj = O.make_proj(seed=5)
    with torch.no_grad():
        supc.projection[0].weight.copy_(proj[0]); supc.projection[0].bias.copy_(proj[1])
        supc.projection[2].weight.copy_(proj[2]); supc.projection[2].bias.copy_(proj[3])
    f = torch.from_numpy(g.standard_normal((8, c, 6, 10), dtype=np.float32)).requires_grad_(True)
    wl = torch.tensor([[0], [1], [0], [3]], dtype=torch.int64)
    l1 = supc(f, class_labels=wl, mask=None)
    l1.backward()
    u.update(sup_feats=np_(f), sup_weather=np_(wl), sup_loss=np_(l1).reshape(()), sup_grad_feats=np_(f.grad),
             sup_grad_w1=np_(supc.projection[0].weight.grad), sup_grad_b2=np_(supc.projection[2].bias.grad))
    f2 = f.detach().clone().requires_grad_(True)
    l2 = supc(f2, class_labels=None, mask=None)
    l2.backward()
    u.update(simclr_loss=np_(l2).reshape(()), simclr_grad_feats=np_(f2.grad))
    # focal variants + CE on full-res logits
    cw = torch.from_numpy((1.0 / np.log(1.1 + g.random(19) * 0.2)).astype(np.float32))
    lg = torch.from_numpy(g.standard_normal((2, 19, 24, 40), dtype=np.float32) * 2).requires_grad_(True)
    tgt = torch.from_numpy(g.integers(0, 19, size=(2, 24, 40)).astype(np.int64))
    tgt[:, :2, :] = 255
    ldw = torch.from_numpy(g.random((2, 24, 40), dtype=np.float32))
    ldw[tgt == 255] = 0
    u.update(foc_logits=np_(lg), foc_target=np_(tgt).astype(np.int16), foc_ldw=np_(ldw), foc_cw=np_(cw))
    for variant in ("full", "plain_focal", "no_class_weights", "no_EDT"):
        o = make_opts("plain_focal" if variant == "plain_focal" else "focal")
        o.no_class_weights = variant == "no_class_weights"
        o.no_EDT = variant == "no_EDT"
        crit = mods.loss.BoundaryAwareFocalLoss(gamma=0.5, num_classes=19, ignore_id=255, weight=cw, device=dev, opts=o)
        lgv = lg.detach().clone().requires_grad_(True)
        t2 = tgt.clone()
        lv = crit(lgv, t2, {"label_distance_weight": ldw})
        lv.backward()
        u[f"foc_loss_{variant}"] = np_(lv).reshape(())
        u[f"foc_grad_{variant}"] = np_(lgv.grad)
    # focal with low-res logits (the reference upsamples inside, loss.py:41-42)
    lr_ = torch.from_numpy(g.standard_normal((2, 19, 6, 10), dtype=np.float32)).requires_grad_(True)
    crit = mods.loss.BoundaryAwareFocalLoss(gamma=0.5, num_classes=19, ignore_id=255, weight=cw, device=dev, opts=make_opts("focal"))
    lv = crit(lr_, tgt.clone(), {"label_distance_weight": ldw})
    lv.backward()
    u.update(foc_lr_logits=np_(lr_), foc_lr_loss=np_(lv).reshape(()), foc_lr_grad=np_(lr_.grad))
    lgv = lg.detach().clone().requires_grad_(True)
    lce = torch.nn.CrossEntropyLoss(ignore_index=255)(lgv, tgt)
    lce.backward()
    u.update(ce_loss=np_(lce).reshape(()), ce_grad=np_(lgv.grad))
    # weather classifier
    clf = mods.classifier.WeatherClassifier(make_opts("focal"), 4)
    fw = torch.from_numpy(g.standard_normal((4, 128), dtype=np.float32) * 0.1)
    fb = torch.from_numpy(g.standard_normal((4,), dtype=np.float32) * 0.1)
    with torch.no_grad():
        clf.fc.weight.copy_(fw); clf.fc.bias.copy_(fb)
    u.update(clf_w=np_(fw), clf_b=np_(fb), clf_out=np_(clf(f.detach())))
    np.savez_compressed(os.path.join(HERE, "loss_units.npz"), **u)
    print("G5", {k: float(u[k]) for k in u if k.endswith("_loss") or "_loss_" in k})


if __name__ == "__main__":
    main()
