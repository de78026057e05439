"""ppnet_amd — MI355X-native (gfx950) implementation of PPNet's two data-parallel loops:
the EDaGe-PP map+path generator and the PPNet inference path.  The compute path is
libppnet_hip.so (hand-written HIP behind a C ABI, include/ppnet_hip.h); importing this package
fails loudly if that library has not been built — there is no CPU fallback.
"""
import os as _os

# MIOpen's find step also times its naive reference convolutions (seconds per call at batch 256); they are never
# the winner, so keep them out of the search.  Must be set before the first convolution initialises MIOpen.
for _k in ("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD",
           "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW"):
    _os.environ.setdefault(_k, "0")

from . import _lib  # noqa: F401,E402  (raises ImportError when libppnet_hip.so is missing)

__version__ = "0.1.0"
