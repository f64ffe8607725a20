"""MI355X-native ESC (Efficient Speech Codec) encode/decode hot path.

Drop-in for the reference package's public surface (`/root/reference/esc/__init__.py:1`,
`esc/models/__init__.py:1`): `from esc import ESC`, `from esc.models import make_model`.
"""
import os as _os

# Batch halves run on two HIP streams; give the runtime enough hardware queues that they do not share one with
# RCCL's / torch's streams (only effective if set before the first HIP call of the process -- see DESIGN.md section 5).
from . import _native as _native_mod  # noqa: E402

_native_mod.ensure_hw_queues()             # also called by _native.load(), i.e. by esc.distributed / scripts.* users that never import this module first

from .models import ESC, make_model  # noqa: F401,E402

__all__ = ["ESC", "make_model"]
