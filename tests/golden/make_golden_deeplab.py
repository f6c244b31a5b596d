#!/usr/bin/env python3
"""Golden vectors for BASELINE config 5 (DeepLabV3+ / ResNet-101, OS16) by RUNNING THE REFERENCE here.
Same conventions as make_golden.py: reference imported through the namespace shim, unmodified code,
``pretrained_backbone=False``, seeded numpy weights/inputs, fixtures hold expected OUTPUTS only."""
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402


def main():
    from oracle import deeplab_oracle as D
    from oracle import swiftnet_oracle as O
    import importlib
    mods = MG.import_reference()
    modeling = importlib.import_module("network.modeling")
    torch.set_num_threads(8)
    dev = torch.device("cpu")
    opts = MG.make_opts("supcon_pixelcontrast_focal")
    opts.deeplab = True
    state = D.make_state(seed=7)
    with contextlib.redirect_stdout(io.StringIO()):
        model = modeling.deeplabv3plus_resnet101(opts, num_classes=19, output_stride=16, pretrained_backbone=False)
    missing, unexpected = model.load_state_dict({k: v.clone() for k, v in state.items()}, strict=True)
    assert list(model.state_dict().keys()) == list(state.keys())
    model.train()
    b, h, w = 2, 128, 256
    img, labels, ldw, weather, cw = O.synthetic_batch(b, h, w, seed=51, two_crops=True, cell=32)
    crit = mods.loss.BoundaryAwareFocalLoss(gamma=0.5, num_classes=19, ignore_id=255, weight=cw, device=dev, opts=opts)
    supc = mods.loss.SupConLoss(temperature=0.07, contrast_mode="all", base_temperature=0.07, weight=cw, device=dev, opts=opts)
    proj = O.make_proj(seed=9, dim_in=2048)
    with torch.no_grad():
        supc.projection[0].weight.copy_(proj[0]); supc.projection[0].bias.copy_(proj[1])
        supc.projection[2].weight.copy_(proj[2]); supc.projection[2].bias.copy_(proj[3])
    pixc = mods.loss.PixelContrastLoss(device=dev)
    labels = labels.clone()
    torch.manual_seed(321)
    seg, before, fine_feat, fine_feat0 = model(img, return_supcon_feature=True)
    sup = supc(fine_feat, class_labels=weather, mask=None)
    with contextlib.redirect_stdout(io.StringIO()):
        pix = pixc(fine_feat0, labels=labels, predict=before)
    segl = crit(seg, labels, {"label_distance_weight": ldw})
    total = 1 / b * (sup + pix) + segl * 1.2
    total.backward()
    np_ = MG.np_
    res = dict(total=np_(total).reshape(()), supcon=np_(sup).reshape(()), pixel=np_(pix).reshape(()), seg=np_(segl).reshape(()))
    res["before"] = np_(before)
    res["fine_feat_sub"] = np_(fine_feat[:, ::8])
    res["fine_feat0_sub"] = np_(fine_feat0[:, ::16, ::2, ::2])
    res["seg_argmax"] = np_(seg.argmax(1)).astype(np.uint8)
    res["seg_logits_sub"] = np_(seg[:, :, ::4, ::4])
    names = [k for k, _ in model.named_parameters()]
    grads = {k: p.grad for k, p in model.named_parameters()}
    res["grad_names"] = np.array(names)
    res["grad_norms"] = np.array([float(grads[k].norm()) if grads[k] is not None else -1.0 for k in names])
    for k in ("backbone.conv1.weight", "backbone.layer1.0.conv1.weight", "backbone.layer4.2.conv2.weight",
              "classifier.aspp.convs.4.1.weight", "classifier.aspp.project.0.weight", "classifier.classifier.0.weight",
              "classifier.classifier.3.bias", "classifier.project.0.weight", "backbone.layer3.22.bn3.weight"):
        g = grads[k]
        res["grad::" + k] = np_(g if g.numel() < 400000 else g.flatten()[::37])
    sd = model.state_dict()
    bn_keys = [k for k in sd if "running_" in k]
    res["rs_names"] = np.array(bn_keys)
    res["rs_norms"] = np.array([float(sd[k].double().norm()) for k in bn_keys])
    res["proj_grad_norms"] = np.array([float(p.grad.norm()) for p in supc.projection.parameters()])
    np.savez_compressed(os.path.join(HERE, "deeplab_step_b2_128x256.npz"), **res)
    print("deeplab", {k: float(res[k]) for k in ("total", "supcon", "pixel", "seg")})
    # eval forward at an odd size (fresh state: the training forward above updated the running statistics)
    model.load_state_dict({k: v.clone() for k, v in state.items()}, strict=True)
    model.eval()
    img2 = O.synthetic_batch(1, 104, 168, seed=52)[0]
    with torch.no_grad():
        seg, before, ff, ff0 = model(img2)
    np.savez_compressed(os.path.join(HERE, "deeplab_eval_b1_104x168.npz"), before=np_(before),
                        fine_feat_sub=np_(ff[:, ::8]), fine_feat0_sub=np_(ff0[:, ::16]),
                        seg_argmax=np_(seg.argmax(1)).astype(np.uint8))
    print("deeplab eval", tuple(seg.shape), tuple(ff.shape), tuple(ff0.shape))


if __name__ == "__main__":
    main()
