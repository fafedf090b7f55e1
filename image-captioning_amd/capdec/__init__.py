"""capdec: MI355X (gfx950) kernels for the caption-decoder hot path, bound through a C ABI (include/capdec.h)."""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
