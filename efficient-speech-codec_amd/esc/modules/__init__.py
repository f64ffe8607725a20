"""Training losses of the reference (`/root/reference/esc/modules/__init__.py` re-exports them from esc/modules/loss)."""
from .loss import ComplexSTFTLoss, GANLoss, MelSpectrogramLoss  # noqa: F401

__all__ = ["ComplexSTFTLoss", "MelSpectrogramLoss", "GANLoss"]
