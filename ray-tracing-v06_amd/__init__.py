"""ray-tracing-v06_amd — MI355X-native hot path of SuperCat908809/Ray-Tracing-v06.

Holds only what the per-pixel sample loop needs: csrc/ (HIP kernels for gfx950 + the C ABI of
include/rt06.h, built into csrc/librt06.so) and the host-side mirror of the reference interface.
The directory name is not a Python identifier; load it with __graft_entry__.load_package().
"""
from . import api, capi  # noqa: F401  (multigpu imports torch; import it explicitly where needed)
from .api import (DefocusBlurCamera, MotionBlurCamera, MultiRenderer, PinholeCamera, Renderer, Scene)  # noqa: F401
from .capi import build_native, lib  # noqa: F401
