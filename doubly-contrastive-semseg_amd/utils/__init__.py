"""Drop-in for the loss re-exports of the reference's ``utils`` package (utils/__init__.py:3)."""
from .loss import SupConLoss, BoundaryAwareFocalLoss        # noqa: F401
