"""Shared by the ``network`` and ``utils`` drop-in packages: make each of them a *merged* package.

The reference imports its model and criteria from ``network`` / ``utils`` (network/__init__.py:1-5,
utils/__init__.py:1-5) and everything else (``utils.init_trainer``, ``utils.saver``, ``utils.logger``,
``network.backbone...``) from those same two packages (trainer.py:13,20-21, utils/init_trainer.py:6-19,
main.py:6-9,23).  A package here that merely shadowed them would hide all of that, so each drop-in package

  * appends the reference's same-named package directory (found further down ``sys.path`` or under
    ``$DCS_REFERENCE_ROOT``) to its own ``__path__``: sub-modules that exist here (``utils.loss``,
    ``network.modeling``) win, every other sub-module resolves to the reference's file;
  * never runs the reference's package ``__init__`` -- the names that ``__init__`` would have re-exported are
    resolved lazily (PEP 562) from the sub-modules it star-imports, so ``import utils`` stays cheap and does not
    pull torchvision / tensorboard until somebody touches e.g. ``utils.Denormalize``.
"""
from __future__ import annotations

import importlib
import os
import sys


def reference_dirs(pkg_name: str, own_dir: str):
    """Directories ``<root>/<pkg_name>`` of OTHER packages with this name, in search order."""
    own = os.path.realpath(own_dir)
    roots = []
    env = os.environ.get("DCS_REFERENCE_ROOT")
    if env:
        roots.append(env)
    roots += [p if p else os.getcwd() for p in sys.path]
    out = []
    for r in roots:
        d = os.path.join(r, pkg_name)
        if os.path.isfile(os.path.join(d, "__init__.py")) and os.path.realpath(d) != own and d not in out:
            out.append(d)
    return out


def extend(pkg_name: str, pkg_path: list, own_dir: str):
    for d in reference_dirs(pkg_name, own_dir):
        if d not in pkg_path:
            pkg_path.append(d)


def lazy_getattr(pkg_name: str, star_modules):
    """PEP 562 ``__getattr__`` for a merged package: ``pkg.name`` is a sub-module of that name, or a public name
    of one of ``star_modules`` (the sub-modules the reference's ``__init__`` star-imports), first hit wins."""
    def __getattr__(name):
        if name.startswith("__"):
            raise AttributeError(name)
        full = pkg_name + "." + name
        try:
            return importlib.import_module(full)
        except ModuleNotFoundError as e:
            if e.name != full:
                raise
        for sub in star_modules:
            try:
                mod = importlib.import_module(pkg_name + "." + sub)
            except ModuleNotFoundError as e:
                if e.name != pkg_name + "." + sub:
                    raise
                continue
            public = getattr(mod, "__all__", None)
            if hasattr(mod, name) and (public is None or name in public) and not name.startswith("_"):
                value = getattr(mod, name)
                sys.modules[pkg_name].__dict__[name] = value          # cache like a real re-export
                return value
        raise AttributeError(f"module {pkg_name!r} has no attribute {name!r}")
    return __getattr__
