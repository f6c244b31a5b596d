"""Checkpoint compatibility with the reference.

* ``checkpoint_state`` / ``save_checkpoint``: the dict ``Trainer.save_checkpoints_sem`` builds (trainer.py:407-423) written the
  way ``Saver.save_checkpoint`` writes it (utils/saver.py:45-70: legacy, non-zip serialisation), with every tensor in the
  reference's own layout (contiguous OIHW weights, stand-alone storages) although the live parameters are
  channels_last views of flat buffers.
* ``load_checkpoint``: the restore of utils/init_trainer.py:246-279 (keep the keys the model knows, ``strict=False``,
  with ``continue_training`` also optimizer / counters), read with ``weights_only=True``.
* ``load_imagenet_backbone`` / ``find_cached_imagenet``: ``resnet18_pyramid(pretrained=True)`` of the reference fetches
  the torchvision ImageNet weights with ``model_zoo.load_url`` (resnet_pyramid.py:14-20, :404); there is no network here,
  so the same file is taken from a LOCAL path (explicit, ``$DCS_IMAGENET_DIR`` or the torch hub cache the reference's
  download would have filled) and loaded with the same ``strict=False`` + ``bn1.*`` fan-out (resnet_pyramid.py:381-393).
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Optional

import torch

IMAGENET_FILES = {          # basenames of resnet_pyramid.py:14-20 (model_urls)
    "resnet18": "resnet18-5c106cde.pth",
    "resnet34": "resnet34-333f7ec4.pth",
    "resnet50": "resnet50-19c8e357.pth",
    "resnet101": "resnet101-5d3b4d8f.pth",
}


def plain_state_dict(module) -> "OrderedDict[str, torch.Tensor]":
    """state_dict() with every tensor cloned into its own contiguous storage (the layout the reference saves)."""
    out = OrderedDict()
    for k, v in module.state_dict().items():
        out[k] = v.detach().clone(memory_format=torch.contiguous_format) if torch.is_tensor(v) else v
    return out


def checkpoint_state(model, optimizer, epoch, num_iter, score=None, best_score=0.0, best_score_epoch=-1):
    """trainer.py:413-423."""
    return {"epoch": epoch, "num_iter": num_iter, "model_state": plain_state_dict(model),
            "optimizer_state": optimizer.state_dict(), "score": score, "best_score": best_score,
            "best_score_epoch": best_score_epoch}


def save_checkpoint(state, path):
    """utils/saver.py:69: legacy serialisation so that older torch versions can read the file."""
    torch.save(state, path, _use_new_zipfile_serialization=False)


def _numpy_scalar_globals():
    """The reference stores ``score`` = Evaluator.get_results() (a dict of numpy float64 scalars / arrays) and
    ``best_score`` = score['Mean IoU'] (np.float64) in its checkpoints (trainer.py:383, :393-399, :413-423).  The
    weights-only unpickler rebuilds them only when numpy's scalar / array reconstructors and dtype classes are
    allow-listed; nothing else from the file can run."""
    import numpy as np
    try:
        from numpy._core import multiarray as _ma
    except ImportError:                                  # numpy < 2
        from numpy.core import multiarray as _ma
    names = ("float64", "float32", "float16", "int64", "int32", "int16", "int8", "uint8", "uint16", "uint32", "uint64",
             "bool")
    return [_ma.scalar, _ma._reconstruct, np.ndarray, np.dtype] + sorted({type(np.dtype(n)) for n in names}, key=repr)


def safe_load(path, map_location=None):
    """torch.load(weights_only=True) that also accepts numpy scalars / arrays (see _numpy_scalar_globals)."""
    with torch.serialization.safe_globals(_numpy_scalar_globals()):
        return torch.load(path, map_location=map_location or "cpu", weights_only=True)


def load_checkpoint(path, model, optimizer=None, continue_training=False, map_location=None):
    """utils/init_trainer.py:246-279.  Returns the bookkeeping fields (empty unless ``continue_training``)."""
    if not os.path.isfile(path):
        raise RuntimeError("=> no checkpoint found at '{}'".format(path))
    ckpt = safe_load(path, map_location)
    loaded = ckpt["model_state"]
    model_dict = model.state_dict()
    model_dict.update({k: v for k, v in loaded.items() if k in model_dict})
    model.load_state_dict(model_dict, strict=False)
    meta = {}
    if continue_training:
        if optimizer is not None:
            optimizer.load_state_dict(ckpt["optimizer_state"])
        meta = {"start_epoch": ckpt["epoch"] + 1, "cur_epochs": ckpt["epoch"] + 1, "num_iter": ckpt["num_iter"] + 1,
                "best_score": ckpt["best_score"], "best_score_epoch": ckpt["best_score_epoch"]}
    return meta


def find_cached_imagenet(arch: str, directory: Optional[str] = None) -> Optional[str]:
    name = IMAGENET_FILES[arch]
    dirs = [directory, os.environ.get("DCS_IMAGENET_DIR"), os.path.join(torch.hub.get_dir(), "checkpoints")]
    for d in dirs:
        if d and os.path.isfile(os.path.join(d, name)):
            return os.path.join(d, name)
    return None


def load_imagenet_backbone(backbone, path):
    """``model.load_state_dict(zoo_state, strict=False)`` of resnet_pyramid.py:404 from a local file.
    Returns (missing_keys, unexpected_keys) like nn.Module.load_state_dict."""
    sd = torch.load(path, map_location="cpu", weights_only=True)
    res = backbone.load_state_dict(sd, strict=False)
    return list(res.missing_keys), list(res.unexpected_keys)
