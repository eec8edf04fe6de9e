"""fs-nerf ray-rendering hot path, MI355X-native.  See DESIGN.md / INTEGRATION.md."""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
