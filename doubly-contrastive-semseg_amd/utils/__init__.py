"""Drop-in for the reference's ``utils`` package (utils/__init__.py:1-5): ``utils.loss`` is the MI355X implementation,
every other sub-module (``utils.init_trainer``, ``utils.saver``, ``utils.logger``, ``utils.utils``, ...) is the
reference's own file (merged package, see ``_dropin.py``); the reference's ``utils/__init__.py`` is not executed and
the names it star-imports (``utils.Denormalize``, ``utils.count_parameters``, ``utils.PolyLR``, ``utils.seed_all_rng`` ...)
resolve lazily."""
import os as _os

import _dropin

_dropin.extend(__name__, __path__, _os.path.dirname(_os.path.abspath(__file__)))

from .loss import SupConLoss, BoundaryAwareFocalLoss        # noqa: F401,E402  (utils/__init__.py:3)

# utils/__init__.py:1-5: `from .utils import *`, `from .scheduler import PolyLR`, `from .tsne import *`,
# `from .logger import *`
__getattr__ = _dropin.lazy_getattr(__name__, ("utils", "scheduler", "logger", "tsne"))
