"""dcs_amd: MI355X-native hot path of andyj1/doubly-contrastive-semseg.

Public surface mirrors the reference modules on the train-step path:
``WeatherNet`` / ``WeatherClassifier`` (network/), the losses of utils/loss.py,
and ``TrainStep`` (the body of trainer.py:62-215).  Compute is libdcs_hip.so
(include/dcs_hip.h); there is no CPU fallback.
"""
from .model import WeatherNet, WeatherClassifier            # noqa: F401
from .losses import (BoundaryAwareFocalLoss, FocalLoss2, SupConLoss, PixelContrastLoss,   # noqa: F401
                     SemsegCrossEntropy)
