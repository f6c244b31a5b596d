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


def np_(t):
    return t.detach().cpu().numpy().copy()


def ref_train_step(mods, state, proj, batch, criterion, batch_size, rng_seed, lr=4e-4, wd=1e-4):
    """trainer.py:62-215 replayed by hand around the imported reference modules."""
    import contextlib, io
    img, labels, ldw, weather, cw = batch
    opts = make_opts(criterion)
    opts.batch_size = batch_size
    dev = torch.device("cpu")
    model = build_ref_model(mods, opts, state)
    model.train()
    crit = mods.loss.BoundaryAwareFocalLoss(gamma=0.5, num_classes=19, ignore_id=255, weight=cw,
                                            device=dev, opts=opts)
    supc = mods.loss.SupConLoss(temperature=0.07, contrast_mode="all", base_temperature=0.07,
                                weight=cw, device=dev, opts=opts)
    with torch.no_grad():
        supc.projection[0].weight.copy_(proj[0]); supc.projection[0].bias.copy_(proj[1])
        supc.projection[2].weight.copy_(proj[2]); supc.projection[2].bias.copy_(proj[3])
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
        # capture sampled anchors without touching the reference code: wrap _contrastive
        orig_c = pixc._contrastive
        def spy(feats_, labels_):
            captured["x_"] = feats_.detach().clone(); captured["y_"] = labels_.detach().clone()
            return orig_c(feats_, labels_)
        pixc._contrastive = spy
        with contextlib.redirect_stdout(io.StringIO()):
            pix = pixc(fine_feat0, labels=labels, predict=before)
        segl = crit(seg, labels, sample)
        total = 1 / batch_size * (sup + pix) + segl * 1.2
    elif criterion == "pixelcontrast_focal":
        orig_c = pixc._contrastive
        def spy(feats_, labels_):
            captured["x_"] = feats_.detach().clone(); captured["y_"] = labels_.detach().clone()
            return orig_c(feats_, labels_)
        pixc._contrastive = spy
        with contextlib.redirect_stdout(io.StringIO()):
            pix = pixc(fine_feat0, labels=labels, predict=before)
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
    res.update(total=np_(total).reshape(()), supcon=np_(sup).reshape(()), pixel=np_(pix).reshape(()),
               seg=np_(segl).reshape(()), ce=np_(ce).reshape(()))
    # outputs are stored spatially subsampled to keep the fixtures small (argmax is kept in full)
    res["before_sub"] = np_(before[:, :, ::2, ::2])
    res["fine_feat_sub"] = np_(fine_feat[:, :, ::4, ::4])
    res["seg_logits_sub"] = np_(seg[:, :, ::8, ::8])
    res["seg_argmax"] = np_(seg.argmax(1)).astype(np.uint8)
    res["labels_after"] = np_(labels).astype(np.int16)
    if captured:
        res["anchor_x"] = np_(captured["x_"]); res["anchor_y"] = np_(captured["y_"])
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
            res["grad::" + k] = np_(grads[k])
    if proj_grads[0] is not None:
        for i, g in enumerate(proj_grads):
            res[f"proj_grad_{i}"] = np_(g)
    sd = model.state_dict()
    res["post_names"] = np.array(list(sd.keys()))
    res["post_norms"] = np.array([float(v.double().norm()) for v in sd.values()], dtype=np.float64)
    for k, v in sd.items():
        if "running_" in k or "num_batches" in k:
            res["post::" + k] = np_(v)
    for k in full:
        res["post::" + k] = np_(sd[k])
    return res


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
    batch = O.synthetic_batch(2, 256, 512, seed=10, two_crops=True, cell=32)
    r = ref_train_step(mods, state, proj, batch, "supcon_pixelcontrast_focal", 2, rng_seed=123)
    np.savez_compressed(os.path.join(HERE, "step_supcon_pixel_focal_b2_256x512.npz"), **r)
    print("G1", {k: float(r[k]) for k in ("total", "supcon", "pixel", "seg")})

    # ---- G2: pixelcontrast_focal, B=2, 200x328 (not a multiple of 32: odd maps everywhere) ---
    batch = O.synthetic_batch(2, 200, 328, seed=11, two_crops=False, cell=24)
    r = ref_train_step(mods, state, proj, batch, "pixelcontrast_focal", 2, rng_seed=7)
    np.savez_compressed(os.path.join(HERE, "step_pixel_focal_b2_200x328.npz"), **r)
    print("G2", {k: float(r[k]) for k in ("total", "pixel", "seg")})

    # ---- G3: CE-only (config 1 miniature), B=2, 256x512 -----------------------
    batch = O.synthetic_batch(2, 256, 512, seed=12, two_crops=False, cell=32)
    r = ref_train_step(mods, state, proj, batch, "crossentropy", 2, rng_seed=1)
    np.savez_compressed(os.path.join(HERE, "step_ce_b2_256x512.npz"), **r)
    print("G3", {k: float(r[k]) for k in ("total", "ce")})

    # ---- G4: eval forward at a size that is NOT a multiple of 32 (validate path,
    #          trainer.py:342-349; default val size 1920x1080 has the same property)
    opts = make_opts("crossentropy")
    model = build_ref_model(mods, opts, state)
    model.eval()
    img = O.synthetic_batch(1, 120, 200, seed=13)[0]
    with torch.no_grad():
        seg, before, ff, ff0 = model(img)
    np.savez_compressed(os.path.join(HERE, "eval_fwd_b1_120x200.npz"),
                        before=np_(before), fine_feat=np_(ff), seg_argmax=np_(seg.argmax(1)).astype(np.uint8),
                        seg_logits_sub=np_(seg[:, :, ::4, ::4]))
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
