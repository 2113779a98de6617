"""emsar_amd -- MI355X-native abundance-estimation core for EMSAR (host mirror + ctypes bindings).

The compute lives in libemsar_hip.so (hand-written HIP for gfx950, C ABI in include/emsar_hip.h); this
package only loads it.  There is no CPU fallback: without the library or without a GPU every entry point
raises.
"""
from .hip import EmsarHip, EmsarHipError, load_library, layout_selfcheck_tiled  # noqa: F401

__all__ = ["EmsarHip", "EmsarHipError", "load_library", "layout_selfcheck_tiled"]
