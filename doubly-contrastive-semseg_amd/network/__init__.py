"""Drop-in for the reference's ``network`` package on the train-step path (network/__init__.py:1-5).

Put this directory in FRONT of the reference root on ``sys.path`` (``python -m dcs_amd.launch <reference>/main.py ...``
does that) and ``main.py`` runs unchanged: ``network.WeatherNet`` / ``network.WeatherClassifier``
and ``network.modeling.deeplabv3plus_resnet{50,101}`` are the MI355X implementations, every other sub-module
(``network.backbone``, ``network._deeplab``, ``network.enet``, ...) is the reference's own file (merged package, see
``_dropin.py``); the reference's ``network/__init__.py`` is not executed."""
import os as _os

import _dropin

_dropin.extend(__name__, __path__, _os.path.dirname(_os.path.abspath(__file__)))

from dcs_amd.model import WeatherNet, WeatherClassifier     # noqa: F401,E402  (network/__init__.py:3-4)
from . import modeling                                      # noqa: F401,E402  (network/__init__.py:1)
from .modeling import deeplabv3plus_resnet101, deeplabv3plus_resnet50   # noqa: F401,E402

# network/__init__.py:1-5 re-exports: `from .modeling import *`, `convert_to_separable_conv`, `ENet`
__getattr__ = _dropin.lazy_getattr(__name__, ("modeling", "_deeplab", "enet"))
