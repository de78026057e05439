"""Neighbourhood attention on MI355X: `NeighborhoodAttention2D` with the constructor, forward signature and
state-dict keys (`qkv.*`, `rpb`, `proj.*`) of natten.NeighborhoodAttention2D as the reference uses it
(SegNet/nat.py:111-120,144), backed by the fused HIP kernel (ppn_na2d_fwd). Forward only; no CPU fallback."""
import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L

# Measurement hook (bench.py): a list here makes every kernel launch record (start event, end event, real tokens, channels)
# on its stream, so the kernel's own duration — and with 8*C bytes per token its HBM fraction — can be read live.
TIMING = None


def na2d_forward(qkv, rpb, heads, dilation, scale, real_hw=None, pad_kv=None, padded_hw=None):
    """qkv: [B,H,W,3*C] contiguous CUDA tensor (float32 or bfloat16) straight from the qkv Linear;
    rpb: [heads,13,13]. Returns [B,Hr,Wr,C] in the layout the output projection consumes.

    Padded layers, two forms.  Materialised: qkv covers the zero-padded H x W grid and real_hw=(Hr,Wr) names the real
    tokens (padded tokens are keys/values only).  Virtual: qkv covers only the real tokens, padded_hw=(H,W) names the
    padded grid and pad_kv [3*C] is the k / v every padded position has (the qkv bias)."""
    if not qkv.is_cuda:
        raise RuntimeError("ppnet_amd.na: the neighbourhood-attention kernel runs on the GPU only (no CPU fallback)")
    B, H, W, C3 = qkv.shape
    ch = C3 // 3
    if ch // heads != 32:
        raise NotImplementedError("head_dim must be 32 (every NAT/DiNAT level)")
    dtype = {torch.float32: 0, torch.bfloat16: 1}.get(qkv.dtype)
    if dtype is None:
        raise NotImplementedError(f"dtype {qkv.dtype}")
    qkv = qkv.contiguous()
    if rpb.dtype != torch.float32 or not rpb.is_contiguous():
        rpb = rpb.detach().to(torch.float32).contiguous()
    if TIMING is None:
        return _launch(qkv, rpb, heads, dilation, scale, real_hw, pad_kv, padded_hw, dtype)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    out = _launch(qkv, rpb, heads, dilation, scale, real_hw, pad_kv, padded_hw, dtype)
    ev1.record()
    TIMING.append((ev0, ev1, out.numel() // ch, ch, qkv.element_size()))
    return out


def _launch(qkv, rpb, heads, dilation, scale, real_hw, pad_kv, padded_hw, dtype):
    B, H, W, C3 = qkv.shape
    ch = C3 // 3
    stream = ctypes.c_void_p(torch.cuda.current_stream(qkv.device).cuda_stream)
    if pad_kv is not None:
        Hp, Wp = padded_hw
        pad_kv = pad_kv.detach().to(qkv.dtype).contiguous()
        assert pad_kv.numel() == C3 and real_hw is None
        out = torch.empty(B, H, W, ch, dtype=qkv.dtype, device=qkv.device)
        with torch.cuda.device(qkv.device):
            rc = L.lib.ppn_na2d_fwd_vpad(ctypes.c_void_p(qkv.data_ptr()), ctypes.c_void_p(pad_kv.data_ptr()),
                                         ctypes.c_void_p(rpb.data_ptr()), ctypes.c_void_p(out.data_ptr()), B, Hp, Wp, H, W,
                                         heads, dilation, float(scale), dtype, stream)
        L.check(rc, "ppn_na2d_fwd_vpad")
        return out
    Hr, Wr = real_hw if real_hw is not None else (H, W)
    out = torch.empty(B, Hr, Wr, ch, dtype=qkv.dtype, device=qkv.device)
    with torch.cuda.device(qkv.device):
        rc = L.lib.ppn_na2d_fwd_padded(ctypes.c_void_p(qkv.data_ptr()), ctypes.c_void_p(rpb.data_ptr()),
                                       ctypes.c_void_p(out.data_ptr()), B, H, W, Hr, Wr, heads, dilation, float(scale), dtype,
                                       stream)
    L.check(rc, "ppn_na2d_fwd_padded")
    return out


class _NA2DFunction(torch.autograd.Function):
    """qkv [B,H,W,3C] (H, W >= 7 * dilation), rpb [heads,13,13] float32 -> [B,H,W,C]; backward on ppn_na2d_bwd."""

    @staticmethod
    def forward(ctx, qkv, rpb, heads, dilation, scale):
        qkv = qkv.contiguous()
        rpb32 = rpb.detach().to(torch.float32).contiguous()
        ctx.save_for_backward(qkv, rpb32)
        ctx.meta = (heads, dilation, scale, rpb.dtype)
        return na2d_forward(qkv.detach(), rpb32, heads, dilation, scale)

    @staticmethod
    def backward(ctx, dout):
        qkv, rpb32 = ctx.saved_tensors
        heads, dilation, scale, rpb_dtype = ctx.meta
        B, H, W, C3 = qkv.shape
        dout = dout.to(qkv.dtype).contiguous()
        dqkv = torch.empty_like(qkv)
        drpb = torch.empty_like(rpb32)
        need = L.lib.ppn_na2d_bwd_workspace(B, H, W, heads, dilation)     # softmax statistics + drpb partial sums; P is recomputed
        if need < 0:
            raise ValueError(f"ppn_na2d_bwd: shape {(B, H, W, heads, dilation)} is outside the kernel")
        ws = torch.empty(need, dtype=torch.float32, device=qkv.device)
        dtype = {torch.float32: 0, torch.bfloat16: 1}[qkv.dtype]
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        with torch.cuda.device(qkv.device):
            rc = L.lib.ppn_na2d_bwd(p(qkv), p(rpb32), p(dout), p(dqkv), p(drpb), p(ws), need, B, H, W, heads, dilation, float(scale), dtype,
                                    ctypes.c_void_p(torch.cuda.current_stream(qkv.device).cuda_stream))
        L.check(rc, "ppn_na2d_bwd")
        return dqkv, drpb.to(rpb_dtype), None, None, None


def na2d_autograd(qkv, rpb, heads, dilation, scale):
    """Differentiable neighbourhood attention on qkv [B,H,W,3C] (the caller pads to 7 * dilation first, as NATTEN's module does)."""
    return _NA2DFunction.apply(qkv, rpb, heads, dilation, scale)


class NeighborhoodAttention2D(nn.Module):
    def __init__(self, dim, kernel_size, dilation=None, num_heads=1, qkv_bias=True, qk_scale=None, attn_drop=0.0,
                 proj_drop=0.0):
        super().__init__()
        if kernel_size != 7:
            raise NotImplementedError("kernel_size 7 only (every configuration in SegNet/configs)")
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = qk_scale or self.head_dim ** -0.5
        self.kernel_size = kernel_size
        self.dilation = dilation or 1
        self.window_size = self.kernel_size * self.dilation
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.rpb = nn.Parameter(torch.zeros(num_heads, 2 * kernel_size - 1, 2 * kernel_size - 1))
        nn.init.trunc_normal_(self.rpb, std=0.02, mean=0.0, a=-2.0, b=2.0)
        self.proj = nn.Linear(dim, dim)
        # attn_drop / proj_drop are identities at inference; kept for signature compatibility
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj_drop = nn.Dropout(proj_drop)
        self._rpb32 = None

    def padded_hw(self, H, W):
        """Token grid after NATTEN's pad-to-kernel*dilation rule, or None when (H, W) is large enough."""
        if H >= self.window_size and W >= self.window_size:
            return None
        return max(H, self.window_size), max(W, self.window_size)

    def forward(self, x, real_hw=None):
        """x [B,H,W,C].  NATTEN's module zero-pads bottom/right to kernel*dilation BEFORE the qkv projection, so every
        padded token's q/k/v is the projection's bias and only the real tokens are queries (no crop needed).  The padded
        grid is never built here: the projection runs on the real tokens and the kernel substitutes the bias for padded
        keys/values (ppn_na2d_fwd_vpad).  real_hw=(Hr,Wr) keeps the materialised form for a caller that already holds a
        zero-padded x."""
        if real_hw is not None:
            o = na2d_forward(self.qkv(x), self.rpb, self.num_heads, self.dilation, self.scale, real_hw)
            return self.proj_drop(self.proj(o))
        if torch.is_grad_enabled() and (x.requires_grad or self.rpb.requires_grad):
            # training: NATTEN's module order — zero-pad bottom / right to kernel * dilation, qkv, NA over the padded grid
            # (differentiable: ppn_na2d_bwd), crop, proj
            B, H, W, _ = x.shape
            pad = self.padded_hw(H, W)
            xp = x if pad is None else F.pad(x, (0, 0, 0, pad[1] - W, 0, pad[0] - H))
            o = na2d_autograd(self.qkv(xp), self.rpb, self.num_heads, self.dilation, self.scale)
            return self.proj_drop(self.proj(o[:, :H, :W]))
        return self.proj_drop(self.proj(self.attend(x)))

    def _rpb_f32(self):
        """The position bias as the float32 tensor the kernel reads; cached while inference leaves the parameter alone
        (a bf16 module would otherwise convert it on every call)."""
        r = self.rpb
        if r.dtype == torch.float32:
            return r.detach()
        key = (r.device, r._version, r.data_ptr())
        if self._rpb32 is None or self._rpb32[0] != key:
            self._rpb32 = (key, r.detach().to(torch.float32).contiguous())
        return self._rpb32[1]

    def attend(self, x, qkv=None):
        """The attention output BEFORE the output projection ([B,H,W,C]) for an unpadded x (padding virtual).  qkv: the
        projection of x when the caller already holds it ([B,H,W,3C]; x is then only read for its shape)."""
        pad = self.padded_hw(x.shape[1], x.shape[2])
        if qkv is None:
            qkv = self.qkv(x)
        if pad is None:
            o = na2d_forward(qkv, self._rpb_f32(), self.num_heads, self.dilation, self.scale)
        else:
            bias = self.qkv.bias if self.qkv.bias is not None else torch.zeros(qkv.shape[-1], dtype=qkv.dtype, device=qkv.device)
            o = na2d_forward(qkv, self._rpb_f32(), self.num_heads, self.dilation, self.scale, pad_kv=bias, padded_hw=pad)
        return o
