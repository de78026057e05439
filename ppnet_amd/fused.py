"""Host side of the fused residual / LayerScale / LayerNorm kernel (ppn_residual_layernorm). GPU only."""
import ctypes

import torch

from . import _lib as L

_DT = {torch.float32: 0, torch.bfloat16: 1}


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(None)


def _call(x, a, gamma, ln, x_out, y_out):
    if not x.is_cuda:
        raise RuntimeError("ppnet_amd.fused: GPU tensors only (no CPU fallback)")
    C = x.shape[-1]
    rows = x.numel() // C
    dt = _DT[x.dtype]
    # parameters follow the activation dtype (a no-op when the module holds weights in that dtype already)
    w = ln.weight.detach().to(x.dtype) if ln is not None else None
    b = ln.bias.detach().to(x.dtype) if ln is not None else None
    gamma = gamma.to(x.dtype) if gamma is not None else None
    for t in (a, gamma, w, b):
        assert t is None or (t.dtype == x.dtype and t.is_contiguous())
    with torch.cuda.device(x.device):
        rc = L.lib.ppn_residual_layernorm(_p(x), _p(a), _p(gamma), _p(w), _p(b), _p(x_out), _p(y_out), rows, C,
                                          float(ln.eps) if ln is not None else 0.0, dt,
                                          ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    L.check(rc, "ppn_residual_layernorm")


def layer_norm(x, ln):
    """y = ln(x) for a torch.nn.LayerNorm over the last dimension."""
    x = x.contiguous()
    y = torch.empty_like(x)
    _call(x, None, None, ln, None, y)
    return y


def residual_layer_norm(x, a, gamma, ln_next):
    """x' = x + gamma * a (gamma None = 1) in place of x; returns (x', ln_next(x')) — y is None when ln_next is None."""
    x = x.contiguous()
    a = a.to(x.dtype).contiguous()
    y = torch.empty_like(x) if ln_next is not None else None
    _call(x, a, gamma.detach() if gamma is not None else None, ln_next, x, y)
    return x, y
