"""Device-side Philox4x32-10 draws as tensors (same streams the generator kernels consume)."""
import ctypes as C

import torch

from . import _lib as L

STREAM_PATH, STREAM_POCKET, STREAM_PLACE, STREAM_OBST = 1, 2, 3, 4


def doubles_device(seed, stream_id, instance, first, count, device):
    device = torch.device(device)
    out = torch.empty(count, dtype=torch.float64, device=device)
    with torch.cuda.device(device):
        rc = L.lib.ppn_philox_doubles(seed, stream_id, instance, first, count, C.c_void_p(out.data_ptr()),
                                      C.c_void_p(torch.cuda.current_stream(device).cuda_stream))
    L.check(rc, "ppn_philox_doubles")
    return out
