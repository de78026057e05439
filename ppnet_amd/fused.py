"""Host side of the fused residual / LayerScale / LayerNorm kernel (ppn_residual_layernorm). GPU only.

Inference runs the HIP kernels below.  When autograd is recording (a training step, ppnet_amd/train.py) the same functions
compose differentiable torch ops instead — the fused kernels are forward-only; the one hand-written backward is the
neighbourhood attention's (ppn_na2d_bwd)."""
import ctypes

import torch
import torch.nn.functional as F

from . import _lib as L

_DT = {torch.float32: 0, torch.bfloat16: 1}


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(None)


class WeightCache:
    """A value derived from module parameters (a packed / folded / converted copy for a kernel), rebuilt whenever a source changes:
    the key holds each source tensor's device, `_version` (bumped by every in-place update: optimizer.step, load_state_dict's
    copy_) and `data_ptr` (a new storage after .to(dtype) / .to(device) or re-assignment).  An eval -> step -> eval flow therefore
    never runs a kernel on stale weights.  NOT covered: writes through `param.data` (EMA / weight-tying code, legacy checkpoint
    loaders) — `.data` is a separate tensor object with its own version counter, so neither the version nor the pointer of the
    parameter moves.  After such a write call `invalidate_caches()` (or `WeightCache.clear()` on the one cache)."""
    __slots__ = ("key", "value", "__weakref__")
    _all = None            # weak set of every live cache (invalidate_caches)

    def __init__(self):
        self.key = self.value = None
        if WeightCache._all is None:
            import weakref
            WeightCache._all = weakref.WeakSet()
        WeightCache._all.add(self)

    def clear(self):
        """Forget the derived value: the next get() rebuilds it."""
        self.key = self.value = None

    def get(self, sources, build, *extra):
        key = tuple((t.device, t._version, t.data_ptr()) if t is not None else None for t in sources) + extra
        if key != self.key:
            self.value, self.key = build(), key
        return self.value


def invalidate_caches():
    """Drop every packed / folded / converted weight copy in the process (all WeightCache instances).  Needed only after
    in-place writes through `param.data`, which no key can see; ordinary updates (optimizer.step, load_state_dict, .to()) are
    tracked by themselves."""
    for c in list(WeightCache._all or ()):
        c.clear()


_PADDED = {}


def _padded_buffer(B, Hp, Wp, C, dtype, device):
    """Persistent zero-initialised [B,Hp,Wp,C] buffer: the kernels only ever write the real tokens, so the pad
    region stays zero; consumers (the qkv projection) read it before the next same-shaped producer runs."""
    key = (B, Hp, Wp, C, dtype, device, torch.cuda.current_stream(device).cuda_stream)     # batches in flight on two streams must not share it
    buf = _PADDED.get(key)
    if buf is None:
        buf = _PADDED[key] = torch.zeros(B, Hp, Wp, C, dtype=dtype, device=device)
    return buf


def _call(x, a, gamma, ln, x_out, y_out, pad=None):
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
    hr, wr, hp, wp = pad if pad is not None else (0, 0, 0, 0)
    with torch.cuda.device(x.device):
        rc = L.lib.ppn_residual_layernorm_padded(_p(x), _p(a), _p(gamma), _p(w), _p(b), _p(x_out), _p(y_out), rows, C,
                                                 float(ln.eps) if ln is not None else 0.0, dt, hr, wr, hp, wp,
                                                 ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    L.check(rc, "ppn_residual_layernorm")


def _y_for(x, pad_to):
    if pad_to is None:
        return torch.empty_like(x), None
    B, Hr, Wr, C = x.shape
    return _padded_buffer(B, pad_to[0], pad_to[1], C, x.dtype, x.device), (Hr, Wr, pad_to[0], pad_to[1])


def recording(*tensors):
    """True when autograd must see the op: grad mode on and an input / parameter requires grad."""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def _layer_norm_autograd(x, ln, pad_to, offset):
    y = F.layer_norm(x if offset is None else x + offset.to(x.dtype), ln.normalized_shape, ln.weight, ln.bias, ln.eps)
    if pad_to is not None:
        y = F.pad(y, (0, 0, 0, pad_to[1] - x.shape[2], 0, pad_to[0] - x.shape[1]))
    return y


def layer_norm(x, ln, pad_to=None, offset=None):
    """y = ln(x) for a torch.nn.LayerNorm over the last dimension. pad_to=(Hp,Wp): x is [B,H,W,C] and y is the
    zero-padded (bottom/right) [B,Hp,Wp,C] grid the next neighbourhood attention wants.  offset [C]: y = ln(x + offset)
    (a residual stream whose constant part is carried outside the tensor, ppn_layernorm_offset)."""
    if recording(x, ln.weight, ln.bias):
        return _layer_norm_autograd(x, ln, pad_to, offset)
    x = x.contiguous()
    if offset is not None:
        assert pad_to is None
        C = x.shape[-1]
        y = torch.empty_like(x)
        w, b = ln.weight.detach().to(x.dtype), ln.bias.detach().to(x.dtype)
        off = offset
        assert off.dtype == torch.float32 and off.is_contiguous() and off.device == x.device
        with torch.cuda.device(x.device):
            rc = L.lib.ppn_layernorm_offset(_p(x), _p(off), _p(w), _p(b), _p(y), x.numel() // C, C, float(ln.eps), _DT[x.dtype],
                                            ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        L.check(rc, "ppn_layernorm_offset")
        return y
    y, pad = _y_for(x, pad_to)
    _call(x, None, None, ln, None, y, pad)
    return y


def residual_layer_norm(x, a, gamma, ln_next, pad_to=None):
    """x' = x + gamma * a (gamma None = 1) in place of x; returns (x', ln_next(x')) — y is None when ln_next is None.
    (Under autograd x' is a new tensor.)"""
    if recording(x, a, gamma, *((ln_next.weight, ln_next.bias) if ln_next is not None else ())):
        x2 = x + (a if gamma is None else gamma * a)
        return x2, (_layer_norm_autograd(x2, ln_next, pad_to, None) if ln_next is not None else None)
    x = x.contiguous()
    a = a.to(x.dtype).contiguous()
    y, pad = _y_for(x, pad_to) if ln_next is not None else (None, None)
    _call(x, a, gamma.detach() if gamma is not None else None, ln_next, x, y, pad)
    return x, y


def upsample2x_nhwc(x_nchw_cl, relu=False, bias=None):
    """Bilinear x2 (align_corners=False) of a channels_last [B,C,H,W] tensor, optionally with a per-channel bias and a
    ReLU folded into the loads (conv -> folded BN -> ReLU -> Upsample with a bias-free library convolution).  Returns a
    channels_last [B,C,2H,2W] tensor (zero-copy views on both sides)."""
    if recording(x_nchw_cl, bias):
        t = x_nchw_cl if bias is None else x_nchw_cl + bias.view(1, -1, 1, 1)
        return F.interpolate(F.relu(t) if relu else t, scale_factor=2.0, mode="bilinear", align_corners=False)
    if not x_nchw_cl.is_cuda:
        raise RuntimeError("ppnet_amd.fused: GPU tensors only (no CPU fallback)")
    x = x_nchw_cl.permute(0, 2, 3, 1)
    if not x.is_contiguous():
        x = x.contiguous()
    B, H, W, C = x.shape
    y = torch.empty(B, 2 * H, 2 * W, C, dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        bias = bias.detach().to(x.dtype).contiguous() if bias is not None else None
        rc = L.lib.ppn_upsample2x_nhwc_bias(_p(x), _p(bias), _p(y), B, H, W, C, 1 if relu else 0, _DT[x.dtype],
                                            ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    L.check(rc, "ppn_upsample2x_nhwc_bias")
    return y.permute(0, 3, 1, 2)


def upsample2x_add_(fine, coarse):
    """fine += bilinear x2 (align_corners=False) of coarse, in place: the FPN's top-down step (uper_head.py:103-108) on channels_last
    [B,C,2H,2W] / [B,C,H,W] tensors, one kernel (ppn_upsample2x_add_nhwc) instead of interpolate + add."""
    f, c = fine.permute(0, 2, 3, 1), coarse.permute(0, 2, 3, 1)
    if not c.is_contiguous():
        c = c.contiguous()
    B, H, W, C = c.shape
    assert f.is_contiguous() and f.is_cuda and f.dtype == c.dtype and f.dtype in _DT and tuple(f.shape) == (B, 2 * H, 2 * W, C)
    with torch.cuda.device(f.device):
        rc = L.lib.ppn_upsample2x_add_nhwc(_p(c), _p(f), _p(f), B, H, W, C, _DT[f.dtype], ctypes.c_void_p(torch.cuda.current_stream(f.device).cuda_stream))
    L.check(rc, "ppn_upsample2x_add_nhwc")
    return fine


def resize_concat(levels):
    """Up to eight channels_last [B,C_l,H_l,W_l] tensors, each bilinearly resized (align_corners=False) to the FIRST one's size and
    concatenated over channels in ONE kernel (ppn_resize_concat_nhwc): UPerHead's FPN output assembly (uper_head.py:117-127) and
    its pyramid pooling module's output (psp_head.py:48-60).  Returns a channels_last [B, sum C_l, H_0, W_0] tensor."""
    assert 1 <= len(levels) <= 8
    xs = []
    for t in levels:
        x = t.permute(0, 2, 3, 1)
        xs.append(x if x.is_contiguous() else x.contiguous())
    B, H0, W0, _ = xs[0].shape
    assert all(x.is_cuda and x.dtype == xs[0].dtype and x.shape[0] == B and x.shape[3] % 8 == 0 for x in xs) and xs[0].dtype in _DT
    n = len(xs)
    out = torch.empty(B, H0, W0, sum(x.shape[3] for x in xs), dtype=xs[0].dtype, device=xs[0].device)
    ptrs = (ctypes.c_void_p * n)(*[x.data_ptr() for x in xs])
    hw = (ctypes.c_int32 * (2 * n))(*[v for x in xs for v in (x.shape[1], x.shape[2])])
    ch = (ctypes.c_int32 * n)(*[x.shape[3] for x in xs])
    with torch.cuda.device(out.device):
        rc = L.lib.ppn_resize_concat_nhwc(ptrs, hw, ch, n, _p(out), B, _DT[xs[0].dtype], ctypes.c_void_p(torch.cuda.current_stream(out.device).cuda_stream))
    L.check(rc, "ppn_resize_concat_nhwc")
    return out.permute(0, 3, 1, 2)


def resize_concat4(levels):
    """The four-level form (UPerHead's FPN output assembly)."""
    assert len(levels) == 4
    return resize_concat(levels)


def adaptive_pools(x_nchw_cl, scales):
    """nn.AdaptiveAvgPool2d(s) for every s in `scales` (<= 4) of one channels_last [B,C,H,W] tensor in ONE kernel
    (ppn_adaptive_pools_nhwc; psp_head.py:33-38).  Returns channels_last [B,C,s,s] tensors."""
    x = x_nchw_cl.permute(0, 2, 3, 1)
    if not x.is_contiguous():
        x = x.contiguous()
    B, H, W, C = x.shape
    n = len(scales)
    assert x.is_cuda and x.dtype in _DT and 1 <= n <= 4 and C % 8 == 0
    ys = [torch.empty(B, s, s, C, dtype=x.dtype, device=x.device) for s in scales]
    ptrs = (ctypes.c_void_p * n)(*[y.data_ptr() for y in ys])
    sc = (ctypes.c_int32 * n)(*[int(s) for s in scales])
    with torch.cuda.device(x.device):
        rc = L.lib.ppn_adaptive_pools_nhwc(_p(x), ptrs, sc, n, B, H, W, C, _DT[x.dtype], ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    L.check(rc, "ppn_adaptive_pools_nhwc")
    return [y.permute(0, 3, 1, 2) for y in ys]


def bias_act_(x_nchw_cl, bias, negative_slope):
    """In place leaky_relu(x + bias[c], negative_slope) on a channels_last [B,C,H,W] tensor (slope 0 = ReLU, 1 = bias
    only).  Returns x."""
    if not x_nchw_cl.is_cuda:
        raise RuntimeError("ppnet_amd.fused: GPU tensors only (no CPU fallback)")
    assert x_nchw_cl.is_contiguous(memory_format=torch.channels_last)
    C = x_nchw_cl.shape[1]
    with torch.cuda.device(x_nchw_cl.device):
        rc = L.lib.ppn_bias_act_nhwc(_p(x_nchw_cl), _p(bias.detach().to(x_nchw_cl.dtype).contiguous()), x_nchw_cl.numel(), C,
                                     float(negative_slope), _DT[x_nchw_cl.dtype],
                                     ctypes.c_void_p(torch.cuda.current_stream(x_nchw_cl.device).cuda_stream))
    L.check(rc, "ppn_bias_act_nhwc")
    return x_nchw_cl


def conv3x3_c1(x, w32, b32, negative_slope):
    """leaky_relu(conv2d(x [B,1,H,W], weight [Cout,1,3,3], bias, stride 1, padding 1), slope) -> channels_last [B,Cout,H,W].
    w32 / b32: the weight and bias as contiguous float32 device tensors."""
    if not x.is_cuda:
        raise RuntimeError("ppnet_amd.fused: GPU tensors only (no CPU fallback)")
    B, _, H, W = x.shape
    Cout = w32.shape[0]
    x = x.contiguous()
    y = torch.empty(B, H, W, Cout, dtype=x.dtype, device=x.device)
    assert w32.dtype == torch.float32 and b32.dtype == torch.float32 and w32.is_contiguous() and b32.is_contiguous()
    with torch.cuda.device(x.device):
        rc = L.lib.ppn_conv3x3_c1_nhwc(_p(x), _p(w32), _p(b32), _p(y), B, H, W, Cout, float(negative_slope), _DT[x.dtype],
                                       ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    L.check(rc, "ppn_conv3x3_c1_nhwc")
    return y.permute(0, 3, 1, 2)


def conv3x3_to1(x_nchw_cl, w32, bias):
    """conv2d(x channels_last [B,Cin,H,W], weight [1,Cin,3,3], bias, stride 1, padding 1) -> [B,1,H,W].
    w32: the weight as a contiguous float32 device tensor; bias: a Python float."""
    if not x_nchw_cl.is_cuda:
        raise RuntimeError("ppnet_amd.fused: GPU tensors only (no CPU fallback)")
    x = x_nchw_cl.permute(0, 2, 3, 1)
    if not x.is_contiguous():
        x = x.contiguous()
    B, H, W, Cin = x.shape
    y = torch.empty(B, 1, H, W, dtype=x.dtype, device=x.device)
    assert w32.dtype == torch.float32 and w32.is_contiguous()
    with torch.cuda.device(x.device):
        rc = L.lib.ppn_conv3x3_to1_nhwc(_p(x), _p(w32), float(bias), _p(y), B, H, W, Cin,
                                        _DT[x.dtype], ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    L.check(rc, "ppn_conv3x3_to1_nhwc")
    return y


def seg_labels_2class(logits_lo, out_hw):
    """u8 labels [B,Ho,Wo] of a two-class segmentor from its low-resolution logits [B,2,h,w]: x2 bilinear, bilinear to out_hw,
    float32 softmax, argmax — the three library kernels of the reference tail in one (ppn_seg_labels_2class)."""
    if not logits_lo.is_cuda:
        raise RuntimeError("ppnet_amd.fused: GPU tensors only (no CPU fallback)")
    x = logits_lo.contiguous()
    B, C2, h, w = x.shape
    assert C2 == 2
    labels = torch.empty(B, out_hw[0], out_hw[1], dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        rc = L.lib.ppn_seg_labels_2class(_p(x), _p(labels), B, h, w, out_hw[0], out_hw[1], _DT[x.dtype],
                                         ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    L.check(rc, "ppn_seg_labels_2class")
    return labels


def grid_to_image(grid_u8, mean, std, dtype):
    """Normalised SegNet input, channels_last [B,3,R,R] of `dtype`, from stage B's u8 codes [B,R,R] (ppn_grid_to_image)."""
    if not grid_u8.is_cuda:
        raise RuntimeError("ppnet_amd.fused: GPU tensors only (no CPU fallback)")
    g = grid_u8.contiguous()
    B, H, W = g.shape
    img = torch.empty(B, H, W, 3, dtype=dtype, device=g.device)
    m3, s3 = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    with torch.cuda.device(g.device):
        rc = L.lib.ppn_grid_to_image(_p(g), _p(img), g.numel(), m3, s3, _DT[dtype], ctypes.c_void_p(torch.cuda.current_stream(g.device).cuda_stream))
    L.check(rc, "ppn_grid_to_image")
    return img.permute(0, 3, 1, 2)


def heatmap_u8(y):
    """[B,R,R] u8 = per-sample min-max normalised y [B,1,R,R] or [B,R,R] (float32 / bfloat16) times 255, truncated
    (ppn_heatmap_u8; GenNet/predict.py:95-102)."""
    if not y.is_cuda or y.dtype not in _DT:
        raise RuntimeError("ppnet_amd.fused.heatmap_u8: float32 / bfloat16 GPU tensors only")
    y = y.contiguous()
    B, H, W = y.shape[0], y.shape[-2], y.shape[-1]
    assert y.numel() == B * H * W
    out = torch.empty(B, H, W, dtype=torch.uint8, device=y.device)
    with torch.cuda.device(y.device):
        rc = L.lib.ppn_heatmap_u8(_p(y), _p(out), B, H * W, _DT[y.dtype], ctypes.c_void_p(torch.cuda.current_stream(y.device).cuda_stream))
    L.check(rc, "ppn_heatmap_u8")
    return out


def tokenizer_lut(conv, mean, std):
    """The [2][64][32] bfloat16 table ppn_tokenizer_conv1_codes_bf16 reads, from the tokenizer's first convolution
    (Conv2d(3, 64, 3, 2, 1), bfloat16 parameters) and the image normalisation: column 3 * (ky * 3 + kx) + colour holds
    sum_ci w[co][ci][ky][kx] * image_ci(colour) with the image values rounded to bfloat16 as ppn_grid_to_image stores them
    (colour 0 = free (255,255,255), 1 = marker (255,0,0), 2 = other (0,0,0)); column 27 the bias; hi + lo split."""
    w = conv.weight.detach().double()                                                    # [64,3,3,3]
    assert w.shape == (64, 3, 3, 3)
    lo = torch.tensor([(0.0 - m) / s for m, s in zip(mean, std)], dtype=torch.float32).to(torch.bfloat16).double()
    hi = torch.tensor([(255.0 - m) / s for m, s in zip(mean, std)], dtype=torch.float32).to(torch.bfloat16).double()
    pal = torch.stack([hi, torch.stack([hi[0], lo[1], lo[2]]), lo]).to(w.device)         # [colour][ci]
    table = torch.zeros(64, 32, dtype=torch.float64, device=w.device)
    table[:, :27] = torch.einsum("oikl,ci->oklc", w, pal).reshape(64, 27)               # k = (ky * 3 + kx) * 3 + colour
    if conv.bias is not None:
        table[:, 27] = conv.bias.detach().double()
    t_hi = table.to(torch.float32).to(torch.bfloat16)
    t_lo = (table - t_hi.double()).to(torch.float32).to(torch.bfloat16)
    t_lo[:, 27:] = 0
    return torch.stack([t_hi, t_lo]).contiguous()        # rows = channels; each kernel applies its own row order when it loads


def tokenizer_pack(conv2, ln):
    """(w2p, vec) of ppn_tokenizer_codes_bf16 from the tokenizer's second convolution (Conv2d(64, 128, 3, 2, 1), bfloat16) and its
    LayerNorm: the weight as MFMA A fragments [8][9][2][16][32] and [3][128] float32 = (conv bias, LN weight, LN bias)."""
    w = conv2.weight.detach().float()                                      # [co 128][ci 64][ky][kx]
    assert w.shape == (128, 64, 3, 3)
    dev = w.device
    co = torch.tensor([[(nt >> 2) * 64 + 16 * (i >> 2) + 4 * (nt & 3) + (i & 3) for i in range(16)] for nt in range(8)], device=dev)         # [8][16]
    ci = torch.tensor([[(2 * s + (e >> 2)) * 16 + 4 * g + (e & 3) for g in range(4) for e in range(8)] for s in range(2)], device=dev)      # [2][32]
    taps = w.reshape(128, 64, 9)
    w2p = taps[co][:, :, ci]                                               # [8][16][2][32][9]
    w2p = w2p.permute(0, 4, 2, 1, 3).contiguous()                          # [8][9][2][16][32]
    bias = conv2.bias.detach().float() if conv2.bias is not None else torch.zeros(128, device=dev)
    vec = torch.stack([bias, ln.weight.detach().float(), ln.bias.detach().float()]).contiguous()
    return w2p.to(torch.bfloat16).contiguous(), vec


def tokenizer_codes(grid_u8, lut, w2p, vec, eps):
    """Tokens [B,R/4,R/4,128] bfloat16 of the palette image of u8 occupancy codes [B,R,R]: both tokenizer convolutions and the
    LayerNorm in one kernel (ppn_tokenizer_codes_bf16)."""
    if not grid_u8.is_cuda or grid_u8.dtype != torch.uint8:
        raise RuntimeError("ppnet_amd.fused.tokenizer_codes: u8 GPU code grids only")
    g = grid_u8.contiguous()
    B, H, W = g.shape
    assert lut.shape == (2, 64, 32) and w2p.shape == (8, 9, 2, 16, 32) and vec.shape == (3, 128)
    assert lut.dtype == w2p.dtype == torch.bfloat16 and vec.dtype == torch.float32 and lut.is_contiguous() and w2p.is_contiguous() and vec.is_contiguous()
    out = torch.empty(B, H // 4, W // 4, 128, dtype=torch.bfloat16, device=g.device)
    with torch.cuda.device(g.device):
        rc = L.lib.ppn_tokenizer_codes_bf16(_p(g), _p(lut), _p(w2p), _p(vec), _p(out), B, H, W, float(eps),
                                            ctypes.c_void_p(torch.cuda.current_stream(g.device).cuda_stream))
    L.check(rc, "ppn_tokenizer_codes_bf16")
    return out


def tokenizer_conv1_codes(grid_u8, lut):
    """[B,R/2,R/2,64] bfloat16 = the tokenizer's first convolution applied to the palette image of the u8 occupancy codes
    [B,R,R] (ppn_tokenizer_conv1_codes_bf16); lut from tokenizer_lut."""
    if not grid_u8.is_cuda or grid_u8.dtype != torch.uint8:
        raise RuntimeError("ppnet_amd.fused.tokenizer_conv1_codes: u8 GPU code grids only")
    g = grid_u8.contiguous()
    B, H, W = g.shape
    assert lut.shape == (2, 64, 32) and lut.dtype == torch.bfloat16 and lut.is_contiguous() and lut.device == g.device
    out = torch.empty(B, H // 2, W // 2, 64, dtype=torch.bfloat16, device=g.device)
    with torch.cuda.device(g.device):
        rc = L.lib.ppn_tokenizer_conv1_codes_bf16(_p(g), _p(lut), _p(out), B, H, W, ctypes.c_void_p(torch.cuda.current_stream(g.device).cuda_stream))
    L.check(rc, "ppn_tokenizer_conv1_codes_bf16")
    return out


def conv3x3_mfma(x_nchw_cl, w_k, bias32, stride=1, relu=False):
    """3x3 convolution (padding 1) of a channels_last bfloat16 [B,Cin,H,W] tensor on the MFMA implicit-GEMM kernel
    (ppn_conv3x3_mfma_bf16).  w_k: the weight as [Cout,3,3,Cin] bfloat16 (weight.permute(0,2,3,1).contiguous()); bias32:
    float32 [Cout].  Returns channels_last [B,Cout,Ho,Wo]."""
    if not x_nchw_cl.is_cuda or x_nchw_cl.dtype != torch.bfloat16:
        raise RuntimeError("ppnet_amd.fused.conv3x3_mfma: bfloat16 GPU tensors only")
    x = x_nchw_cl.permute(0, 2, 3, 1)
    if not x.is_contiguous():
        x = x.contiguous()
    B, H, W, Cin = x.shape
    Cout = w_k.shape[0]
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    assert w_k.dtype == torch.bfloat16 and w_k.is_contiguous() and w_k.shape == (Cout, 3, 3, Cin)
    assert bias32.dtype == torch.float32 and bias32.is_contiguous() and bias32.numel() == Cout
    y = torch.empty(B, Ho, Wo, Cout, dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        rc = L.lib.ppn_conv3x3_mfma_bf16(_p(x), _p(w_k), _p(bias32), _p(y), B, H, W, Cin, Cout, stride, 1 if relu else 0,
                                         ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    L.check(rc, "ppn_conv3x3_mfma_bf16")
    return y.permute(0, 3, 1, 2)


def conv3x3_relu_classify2(x_nchw_cl, w_k, bias32, w2_32, b2_32):
    """relu(conv3x3(x) + bias) followed by a 2-class 1x1 classifier, without writing the Cout-channel activation
    (ppn_conv3x3_relu_classify2_bf16).  w2_32 [2,Cout], b2_32 [2] float32.  Returns float32 logits [B,2,H,W]."""
    if not x_nchw_cl.is_cuda or x_nchw_cl.dtype != torch.bfloat16:
        raise RuntimeError("ppnet_amd.fused.conv3x3_relu_classify2: bfloat16 GPU tensors only")
    x = x_nchw_cl.permute(0, 2, 3, 1)
    if not x.is_contiguous():
        x = x.contiguous()
    B, H, W, Cin = x.shape
    Cout = w_k.shape[0]
    logits = b2_32.to(torch.float32).repeat(B * H * W).view(B, H, W, 2).contiguous()
    slots = L.lib.ppn_conv3x3_relu_classify2_slots(Cout)            # 0 for Cout <= 256: the kernel adds straight onto the logits
    partial = torch.empty(slots, B * H * W, 2, dtype=torch.float32, device=x.device) if slots > 0 else None
    with torch.cuda.device(x.device):
        rc = L.lib.ppn_conv3x3_relu_classify2_bf16(_p(x), _p(w_k), _p(bias32), _p(w2_32), _p(logits), _p(partial), B, H, W, Cin, Cout,
                                                   ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    L.check(rc, "ppn_conv3x3_relu_classify2_bf16")
    return logits.permute(0, 3, 1, 2)


def gemm_bf16(a, w, bias32, epilogue="bias", out=None, persistent_blocks=0):
    """out[M,N] = epilogue(a[M,K] @ w[N,K]^T) on the MFMA kernel (ppn_gemm_bf16).  epilogue: "bias", "bias_gelu", or "accum"
    (out += a @ w^T, bias unused)."""
    epi = {"bias": 0, "bias_gelu": 1, "accum": 2, "bias_relu": 3}[epilogue]
    M, K = a.shape
    N = w.shape[0]
    assert a.is_cuda and a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and a.is_contiguous() and w.is_contiguous()
    if out is None:
        assert epi != 2
        out = torch.empty(M, N, dtype=a.dtype, device=a.device)
    assert out.is_contiguous() and out.shape == (M, N)
    with torch.cuda.device(a.device):
        rc = L.lib.ppn_gemm_bf16(_p(a), _p(w), _p(bias32), _p(out), M, N, K, epi, persistent_blocks,
                                 ctypes.c_void_p(torch.cuda.current_stream(a.device).cuda_stream))
    L.check(rc, "ppn_gemm_bf16")
    return out


def nat_gemm(a, w, bias32, mode, out, colsum=None, stats_in=None, stats_out=None, eps=1e-5):
    """The NAT projections with the LayerNorm / residual / statistics in the epilogue (ppn_nat_gemm_bf16, csrc/nat_gemm.hip).
    mode "ln": out = LN(a) W0^T + b0 from the raw rows of a (w = W0 diag(gamma), bias32 = b0 + W0 beta, colsum, stats_in
    [P, M, 2]); "ln_gelu": gelu of that; "acc": out += a w^T + bias32 in place, row partials of the new out -> stats_out
    [nat_partials(N), M, 2]."""
    md = {"ln": 0, "ln_gelu": 1, "acc": 2}[mode]
    M, K = a.shape
    N = w.shape[0]
    assert a.is_cuda and a.dtype == w.dtype == out.dtype == torch.bfloat16 and a.is_contiguous() and w.is_contiguous() and out.is_contiguous()
    assert w.shape == (N, K) and out.shape == (M, N) and bias32.dtype == torch.float32 and bias32.numel() == N and bias32.is_contiguous()
    P = 0
    if md != 2:
        assert colsum.dtype == torch.float32 and colsum.numel() == N and colsum.is_contiguous()
        assert stats_in.dtype == torch.float32 and stats_in.is_contiguous() and stats_in.dim() == 3 and stats_in.shape[1:] == (M, 2)
        P = stats_in.shape[0]
    else:
        assert stats_out.dtype == torch.float32 and stats_out.is_contiguous() and stats_out.shape == (nat_partials(N), M, 2)
    with torch.cuda.device(a.device):
        rc = L.lib.ppn_nat_gemm_bf16(_p(a), _p(w), _p(bias32), _p(colsum), _p(stats_in), P, _p(stats_out), _p(out), M, N, K, md, float(eps),
                                     ctypes.c_void_p(torch.cuda.current_stream(a.device).cuda_stream))
    L.check(rc, "ppn_nat_gemm_bf16")
    return out


def nat_mlp_ok(M, C, hidden):
    """Shapes the fused MLP kernel serves (ppn_nat_mlp_supported): C = 256 streams, whole 128-token blocks."""
    return bool(L.lib.ppn_nat_mlp_supported(int(M), int(C), int(hidden)))


def nat_mlp_pack(w1_folded, w2):
    """The two MLP weights in the fused kernel's streaming order (ppn_nat_mlp_pack_bf16): w1_folded [hidden, C] = W1 diag(gamma)
    (LayerNorm folded in), w2 [C, hidden] (LayerScale folded in), both bfloat16 -> one bfloat16 tensor of 2 * C * hidden."""
    hid, Cc = w1_folded.shape
    assert w1_folded.is_cuda and w1_folded.dtype == w2.dtype == torch.bfloat16 and tuple(w2.shape) == (Cc, hid)
    w1_folded, w2 = w1_folded.contiguous(), w2.contiguous()
    out = torch.empty(2 * Cc * hid, dtype=torch.bfloat16, device=w1_folded.device)
    with torch.cuda.device(out.device):
        rc = L.lib.ppn_nat_mlp_pack_bf16(_p(w1_folded), _p(w2), _p(out), Cc, hid, ctypes.c_void_p(torch.cuda.current_stream(out.device).cuda_stream))
    L.check(rc, "ppn_nat_mlp_pack_bf16")
    return out


def nat_mlp_(s2d, wpk, hb, b2, hidden, stats_out=None, eps=1e-5):
    """s += gelu(LN(s) W1^T + b1) W2^T + b2 in place on the residual stream s2d [M, C] (bfloat16) in ONE kernel — the hidden
    activation never reaches HBM (ppn_nat_mlp_bf16, csrc/nat_mlp.hip).  wpk from nat_mlp_pack; hb [hidden, 2] float32 =
    (colsum of the folded w1 rows, b1 + W1 beta); b2 [C] float32; stats_out [C / 128, M, 2] receives the row partials of the new s."""
    M, Cc = s2d.shape
    assert s2d.is_cuda and s2d.dtype == torch.bfloat16 and s2d.is_contiguous() and wpk.dtype == torch.bfloat16 and wpk.numel() == 2 * Cc * hidden
    assert hb.dtype == torch.float32 and hb.is_contiguous() and tuple(hb.shape) == (hidden, 2) and b2.dtype == torch.float32 and b2.numel() == Cc
    assert stats_out is None or (stats_out.dtype == torch.float32 and stats_out.is_contiguous() and tuple(stats_out.shape) == (Cc // 128, M, 2))
    with torch.cuda.device(s2d.device):
        rc = L.lib.ppn_nat_mlp_bf16(_p(s2d), _p(wpk), _p(hb), _p(b2), _p(stats_out), M, Cc, hidden, float(eps),
                                    ctypes.c_void_p(torch.cuda.current_stream(s2d.device).cuda_stream))
    L.check(rc, "ppn_nat_mlp_bf16")
    return s2d


def nat_partials(C):
    """Row-statistics partials per row of a residual stream of width C (ppn_nat_gemm_partials: one per 128 columns up to
    C = 256, one per 256 columns above — what the accumulating GEMM's epilogue emits and the LayerNorm-folding GEMM reads)."""
    n = L.lib.ppn_nat_gemm_partials(C)
    if n < 0:
        raise ValueError(f"ppn_nat_gemm_partials({C})")
    return n


def nat_gemm_ok(M, N, K, acc):
    """Shapes ppn_nat_gemm_bf16 serves: whole 256 x 256 tiles, at least three k-tiles in a tile (the accumulating mode adds four)."""
    return M > 0 and M % 256 == 0 and N % 256 == 0 and K % 64 == 0 and (K // 64 + (4 if acc else 0)) >= 3


def row_stats(x2d):
    """[1, rows, 2] float32 = (sum, sum of squares) of every row of a bfloat16 [rows, C] tensor (ppn_row_stats_bf16)."""
    rows, Cc = x2d.shape
    assert x2d.is_cuda and x2d.dtype == torch.bfloat16 and x2d.is_contiguous()
    st = torch.empty(1, rows, 2, dtype=torch.float32, device=x2d.device)
    with torch.cuda.device(x2d.device):
        rc = L.lib.ppn_row_stats_bf16(_p(x2d), rows, Cc, _p(st), ctypes.c_void_p(torch.cuda.current_stream(x2d.device).cuda_stream))
    L.check(rc, "ppn_row_stats_bf16")
    return st


def gennet_conv_s2(x_nchw_cl, w_packed, bias32, negative_slope, transposed):
    """GenNet's 24-channel stride-2 conv (transposed=False) / transposed conv (True) + bias + LeakyReLU on MFMA
    (ppn_gennet_conv_s2_bf16).  x: channels_last bfloat16 [B,24,H,W]; w_packed / bias32 from gennet.pack_s2_weights."""
    if not x_nchw_cl.is_cuda or x_nchw_cl.dtype != torch.bfloat16:
        raise RuntimeError("ppnet_amd.fused.gennet_conv_s2: bfloat16 GPU tensors only")
    x = x_nchw_cl.permute(0, 2, 3, 1)
    if not x.is_contiguous():
        x = x.contiguous()
    B, H, W, C = x.shape
    assert C == 24
    Ho, Wo = (2 * H, 2 * W) if transposed else (H // 2, W // 2)
    y = torch.empty(B, Ho, Wo, C, dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        rc = L.lib.ppn_gennet_conv_s2_bf16(_p(x), _p(w_packed), _p(bias32), _p(y), B, H, W, float(negative_slope), 1 if transposed else 0,
                                           ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    L.check(rc, "ppn_gennet_conv_s2_bf16")
    return y.permute(0, 3, 1, 2)


def _nat128_args(s, offset, ln):
    if not s.is_cuda or s.dtype != torch.bfloat16 or not s.is_contiguous() or s.shape[-1] != 128:
        raise RuntimeError("ppnet_amd.fused.nat128: contiguous bfloat16 GPU token rows of 128 channels only")
    assert offset is None or (offset.dtype == torch.float32 and offset.is_contiguous() and offset.device == s.device and offset.numel() == 128)
    return ln.weight.detach().to(torch.bfloat16), ln.bias.detach().to(torch.bfloat16)


def nat128_ln_qkv(s, offset, ln, qkv_linear):
    """qkv = qkv_linear(ln(s + offset)) for 128-channel token rows s [...,128] in ONE kernel (ppn_nat128_ln_qkv_bf16): the
    normalised tokens go from registers straight into the matrix pipe.  Returns [...,384] bfloat16."""
    lw, lb = _nat128_args(s, offset, ln)
    w = qkv_linear.weight.detach()
    assert w.shape == (384, 128) and w.dtype == torch.bfloat16 and w.is_contiguous()
    bias = qkv_linear.bias.detach() if qkv_linear.bias is not None else None
    out = torch.empty(*s.shape[:-1], 384, dtype=s.dtype, device=s.device)
    with torch.cuda.device(s.device):
        rc = L.lib.ppn_nat128_ln_qkv_bf16(_p(s), _p(offset) if offset is not None else None, _p(lw), _p(lb), _p(w),
                                          _p(bias) if bias is not None else None, _p(out), s.numel() // 128, float(ln.eps),
                                          ctypes.c_void_p(torch.cuda.current_stream(s.device).cuda_stream))
    L.check(rc, "ppn_nat128_ln_qkv_bf16")
    return out


def nat128_ln_mlp_(s, offset, ln, fc1, fc2, final_add=None):
    """s += fc2.weight @ gelu(fc1(ln(s + offset))) (+ final_add) in place, ONE kernel (ppn_nat128_ln_mlp_add_bf16): the hidden
    activations never reach HBM.  fc2's bias is NOT added (the folded layer carries it in the next offset); final_add [128]
    float32: a per-channel constant added to the result (the level's accumulated biases, on its last layer)."""
    lw, lb = _nat128_args(s, offset, ln)
    w1, w2 = fc1.weight.detach(), fc2.weight.detach()
    assert w1.shape == (256, 128) and w2.shape == (128, 256) and w1.dtype == w2.dtype == torch.bfloat16
    assert w1.is_contiguous() and w2.is_contiguous()
    with torch.cuda.device(s.device):
        if final_add is not None:
            assert final_add.dtype == torch.float32 and final_add.is_contiguous() and final_add.numel() == 128 and final_add.device == s.device
        rc = L.lib.ppn_nat128_ln_mlp_add_bf16(_p(s), _p(offset) if offset is not None else None, _p(lw), _p(lb), _p(w1), _p(fc1.bias.detach()),
                                              _p(w2), _p(final_add), s.numel() // 128, float(ln.eps),
                                              ctypes.c_void_p(torch.cuda.current_stream(s.device).cuda_stream))
    L.check(rc, "ppn_nat128_ln_mlp_add_bf16")
    return s


def nat128_proj_add_(s, a, proj):
    """s += a @ proj.weight^T in place for 128-channel token rows (ppn_nat128_proj_add_bf16; no bias: the folded layer carries it in
    the next offset)."""
    w = proj.weight.detach()
    assert s.is_cuda and s.dtype == a.dtype == w.dtype == torch.bfloat16 and s.is_contiguous() and a.is_contiguous() and w.is_contiguous()
    assert s.shape[-1] == 128 and a.shape == s.shape and w.shape == (128, 128)
    with torch.cuda.device(s.device):
        rc = L.lib.ppn_nat128_proj_add_bf16(_p(s), _p(a), _p(w), s.numel() // 128, ctypes.c_void_p(torch.cuda.current_stream(s.device).cuda_stream))
    L.check(rc, "ppn_nat128_proj_add_bf16")
    return s


def gennet_dec_final(x_nchw_cl, w_packed, bias32, negative_slope, w_final32, bias_final):
    """GenNet's last decoder stage + final convolution as one kernel (ppn_gennet_dec_final_bf16): x channels_last bfloat16
    [B,24,H,W] -> [B,1,2H,2W]; w_packed / bias32 from gennet.pack_s2_weights (transposed), w_final32 the final convolution's
    float32 weight [1,24,3,3], bias_final a Python float."""
    if not (x_nchw_cl.is_cuda and x_nchw_cl.dtype == torch.bfloat16):
        raise RuntimeError("ppnet_amd.fused.gennet_dec_final: bfloat16 GPU tensors only")
    x = x_nchw_cl.permute(0, 2, 3, 1)
    if not x.is_contiguous():
        x = x.contiguous()
    B, H, W, C = x.shape
    assert C == 24 and w_final32.dtype == torch.float32 and w_final32.is_contiguous() and w_final32.numel() == 24 * 9
    y = torch.empty(B, 1, 2 * H, 2 * W, dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        rc = L.lib.ppn_gennet_dec_final_bf16(_p(x), _p(w_packed), _p(bias32), float(negative_slope), _p(w_final32), float(bias_final), _p(y), B, H, W,
                                             ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    L.check(rc, "ppn_gennet_dec_final_bf16")
    return y


def gennet_first_enc(x, w1, b1, wk2, b2, slope1, slope2):
    """GenNet's first convolution + first encoder stage as one kernel (ppn_gennet_first_enc_bf16): x [B,1,H,W] bfloat16 ->
    channels_last [B,24,H/2,W/2]; parameters from gennet.pack_first_enc_weights."""
    if not x.is_cuda or x.dtype != torch.bfloat16:
        raise RuntimeError("ppnet_amd.fused.gennet_first_enc: bfloat16 GPU tensors only")
    B, _, H, W = x.shape
    x = x.contiguous()
    assert w1.shape == (2, 16, 32) and wk2.shape == (2, 9, 16, 32) and w1.dtype == wk2.dtype == torch.bfloat16
    assert b1.dtype == b2.dtype == torch.float32 and b1.numel() == b2.numel() == 32
    y = torch.empty(B, H // 2, W // 2, 24, dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        rc = L.lib.ppn_gennet_first_enc_bf16(_p(x), _p(w1), _p(b1), _p(wk2), _p(b2), _p(y), B, H, W, float(slope1), float(slope2),
                                             ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    L.check(rc, "ppn_gennet_first_enc_bf16")
    return y.permute(0, 3, 1, 2)


def gennet_trunk(x_nchw_cl, params32, n_blocks):
    """GenNet's ViT blocks as one kernel (ppn_gennet_trunk_bf16): channels_last bfloat16 [B,24,H,W] in and out; params32 from
    gennet.pack_trunk_params."""
    if not x_nchw_cl.is_cuda or x_nchw_cl.dtype != torch.bfloat16:
        raise RuntimeError("ppnet_amd.fused.gennet_trunk: bfloat16 GPU tensors only")
    x = x_nchw_cl.permute(0, 2, 3, 1)
    if not x.is_contiguous():
        x = x.contiguous()
    B, H, W, C = x.shape
    assert C == 24 and params32.dtype == torch.float32 and params32.is_contiguous() and params32.numel() == n_blocks * 7224
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        rc = L.lib.ppn_gennet_trunk_bf16(_p(x), _p(y), _p(params32), B, H * W, n_blocks,
                                         ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    L.check(rc, "ppn_gennet_trunk_bf16")
    return y.permute(0, 3, 1, 2)
