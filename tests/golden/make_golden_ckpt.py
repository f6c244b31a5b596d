#!/usr/bin/env python3
"""Golden for checkpoint compatibility, produced by RUNNING THE REFERENCE here (not on the GPU box):

  * ImageNet initialisation: the reference's ``ResNet`` (network/backbone/resnet_pyramid.py) built with
    ``pretrained=False`` and then given a torchvision-style state dict through the exact call of
    resnet_pyramid.py:404 (``load_state_dict(state, strict=False)``, with the ``bn1.*`` fan-out of :381-393);
  * restore: the key filter + ``strict=False`` load of utils/init_trainer.py:268-274 applied to a checkpoint that
    has one foreign key and lacks one key of the model.

Stored: per-key SHA-1 digests of the resulting ``state_dict()`` (data only), in tests/golden/ckpt_compat.json.
"""
import contextlib
import io
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import ckpt_util  # noqa: E402
from make_golden import import_reference, make_opts  # noqa: E402


def main():
    mods = import_reference()
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        model = mods.weathernet.WeatherNet(make_opts("crossentropy"), num_classes=19, device=torch.device("cpu"),
                                           backbone="resnet18", train_semantic=True)
    fe = model.feature_extractor
    before = {k: ckpt_util.digest(v) for k, v in fe.state_dict().items()}
    res = fe.load_state_dict(ckpt_util.tv_resnet18_state(), strict=False)          # resnet_pyramid.py:404
    after = {k: ckpt_util.digest(v) for k, v in fe.state_dict().items()}
    out = {"imagenet": {"missing": list(res.missing_keys), "unexpected": list(res.unexpected_keys),
                        "changed": {k: after[k] for k in after if after[k] != before[k]},
                        "unchanged": [k for k in after if after[k] == before[k]]}}

    # utils/init_trainer.py:268-274 on a checkpoint with a foreign key and a missing key
    loaded = {k: (v.clone() + 1 if v.is_floating_point() else v.clone()) for k, v in model.state_dict().items()}
    dropped = "segmentation.conv.bias"
    del loaded[dropped]
    loaded["module.not_in_model"] = torch.zeros(3)
    model_dict = model.state_dict()
    pretrained_dict = {k: v for k, v in loaded.items() if k in model_dict}
    model_dict.update(pretrained_dict)
    model.load_state_dict(model_dict, strict=False)
    fin = model.state_dict()
    out["restore"] = {"dropped": dropped,
                      "plus_one": all(torch.equal(fin[k], loaded[k]) for k in loaded if k in fin),
                      "kept": ckpt_util.digest(fin[dropped]),
                      "keys": list(fin.keys())}
    with open(os.path.join(HERE, "ckpt_compat.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("changed", len(out["imagenet"]["changed"]), "unchanged", len(out["imagenet"]["unchanged"]),
          "missing", len(res.missing_keys), "unexpected", res.unexpected_keys)


if __name__ == "__main__":
    main()
