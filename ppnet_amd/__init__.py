"""ppnet_amd — MI355X-native (gfx950) implementation of PPNet's two data-parallel loops:
the EDaGe-PP map+path generator and the PPNet inference path.  The compute path is
libppnet_hip.so (hand-written HIP behind a C ABI, include/ppnet_hip.h); importing this package
fails loudly if that library has not been built — there is no CPU fallback.
"""
from . import _lib  # noqa: F401  (raises ImportError when libppnet_hip.so is missing)

__version__ = "0.1.0"
