"""Drop-in for the DeepLabV3+ factories of the reference's network/modeling.py (looked up by name through
``modeling.__dict__[opts.model]`` at utils/init_trainer.py:102)."""
from dcs_amd.deeplab import deeplabv3plus_resnet101, deeplabv3plus_resnet50     # noqa: F401
