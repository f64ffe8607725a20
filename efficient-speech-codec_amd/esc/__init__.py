"""MI355X-native ESC (Efficient Speech Codec) encode/decode hot path.

Drop-in for the reference package's public surface (`/root/reference/esc/__init__.py:1`,
`esc/models/__init__.py:1`): `from esc import ESC`, `from esc.models import make_model`.
"""
from .models import ESC, make_model  # noqa: F401

__all__ = ["ESC", "make_model"]
