"""MI355X (gfx950) render path for RayTracing_Simple, behind the C ABI of include/rt_api.h.

The compute path is the hand-written HIP library `librt_hip.so` built from `csrc/`; this
package is the thin host side above it: ctypes bindings (`api`) and scene / image helpers
(`host`, `scenes`).  The reference-side binding -- `HipConfig : Config` -- is C++ like the
reference (adapter/HipConfig.{hpp,cpp}).  There is no CPU fallback: importing works anywhere, but creating a renderer
without the built library or without a gfx950 device raises.
"""
from .api import RtContext, RtError, Stats, lib_path, load_library, render  # noqa: F401

__all__ = ["RtContext", "RtError", "Stats", "render", "load_library", "lib_path"]
