"""MI355X (gfx950) render path for RayTracing_Simple, behind the C ABI of include/rt_api.h.

The compute path is the hand-written HIP library `librt_hip.so` built from `csrc/`; this
package is the thin host side above it: ctypes bindings (`api`), the mirror of the
reference's backend interface (`config.HipConfig`), scene / image helpers (`host`,
`scenes`).  There is no CPU fallback: importing works anywhere, but creating a renderer
without the built library or without a gfx950 device raises.
"""
from .api import RtContext, RtError, Stats, lib_path, load_library, render  # noqa: F401
from .config import HipConfig, MemType, SupportType, createConfig, selectType  # noqa: F401

__all__ = ["RtContext", "RtError", "Stats", "render", "load_library", "lib_path", "HipConfig",
           "createConfig", "selectType", "SupportType", "MemType"]
