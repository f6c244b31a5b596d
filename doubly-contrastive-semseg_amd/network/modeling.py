"""Drop-in for the DeepLabV3+ factories of the reference's network/modeling.py (looked up by name through
``modeling.__dict__[opts.model]`` at utils/init_trainer.py:102).

``deeplabv3plus_resnet101`` / ``deeplabv3plus_resnet50`` are the MI355X implementations.  The reference's other
factories (mobilenet / hrnet / xception / plain deeplabv3 heads: SURVEY.md 9.3, out of scope) stay reachable: when
the reference's own ``network/modeling.py`` is importable it is loaded under ``network._reference_modeling`` and its
remaining public names are copied in here (``__dict__`` lookups need real entries, PEP 562 would not do)."""
import importlib.util as _ilu
import os as _os
import sys as _sys

from dcs_amd.deeplab import deeplabv3plus_resnet101, deeplabv3plus_resnet50     # noqa: F401

_OWN = {"deeplabv3plus_resnet101", "deeplabv3plus_resnet50"}


def _adopt_reference_factories():
    pkg = _sys.modules.get(__package__)
    own = _os.path.realpath(_os.path.dirname(_os.path.abspath(__file__)))
    for d in list(getattr(pkg, "__path__", [])):
        f = _os.path.join(d, "modeling.py")
        if _os.path.realpath(d) == own or not _os.path.isfile(f):
            continue
        name = __package__ + "._reference_modeling"
        spec = _ilu.spec_from_file_location(name, f)
        mod = _ilu.module_from_spec(spec)
        _sys.modules[name] = mod
        try:
            spec.loader.exec_module(mod)
        except ImportError:                      # a dependency of the out-of-scope models is absent: keep ours only
            _sys.modules.pop(name, None)
            return None
        for k, v in vars(mod).items():
            if not k.startswith("_") and k not in _OWN and k not in globals():
                globals()[k] = v
        return mod
    return None


_reference_modeling = _adopt_reference_factories()
