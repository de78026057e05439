"""SegNet = NAT / DiNAT backbone + SETR-UP head (reference SegNet/nat.py:17-332, dinat.py:15-22,
mmseg/decode_heads/setr_up_head.py:28-81, mmseg/models/segmentors/encoder_decoder.py:63-80,200-265) as plain
inference nn.Modules with the reference's constructor arguments and checkpoint key names
(`backbone.patch_embed.proj.{0,1}`, `backbone.levels.i.blocks.j.{norm1,attn.{qkv,rpb,proj},norm2,mlp.{fc1,fc2},
gamma1,gamma2}`, `backbone.levels.i.downsample.{reduction,norm}`, `backbone.norm{i}`,
`decode_head.{norm,up_convs.i.0.{conv,bn},conv_seg}`), so mmcv `{'state_dict', 'meta'}` checkpoints load.

The neighbourhood attention is the hand-written HIP kernel (ppnet_amd/na.py); tokenizer / downsampler / head
convolutions and the linear projections run on the ROCm libraries through PyTorch.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fused
from .na import NeighborhoodAttention2D


def drop_path(x, rate, training):
    """Stochastic depth per sample (timm's DropPath as SegNet/nat.py:122,145-152 and GenNet/networks/vit.py:150-161 use it):
    identity unless training with rate > 0."""
    if not training or rate <= 0.0:
        return x
    keep = 1.0 - rate
    mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
    return x * (mask / keep)


def _use_mfma_conv(x, conv, narrow=False):
    """The hand-written MFMA implicit-GEMM convolution (ppn_conv3x3_mfma_bf16) serves bfloat16 inference on the GPU for 3x3,
    padding 1, Cin % 64 == 0, Cout % 256 == 0 (whole 256-wide output tiles); everything else stays on the library.
    narrow=True also takes Cout % 8 == 0 (UPerNet's 64-channel convolutions: a quarter of the 256-wide tile computes — they are
    3 % of the backbone's arithmetic, and a library call per convolution costs more than the idle columns)."""
    import os
    if fused.recording(x, conv.weight):
        return False                                   # training: the library convolution (differentiable)
    return (x.is_cuda and x.dtype == torch.bfloat16 and not conv.training and conv.kernel_size == (3, 3) and conv.padding == (1, 1)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.in_channels % 64 == 0
            and conv.out_channels % (8 if narrow else 256) == 0
            and conv.weight.dtype == torch.bfloat16 and not os.environ.get("PPNET_LIBRARY_CONV"))


# Per-shape gate between the build's GEMMs and the vendor library behind LayerNorm kernels: stream widths C with
# C >= LIBRARY_GEMM_FROM_C or C < LIBRARY_GEMM_BELOW_C take the library.  Defaults from the alternating A/B of tools/ppnet_ab.py
# (DESIGN.md section 4); PPNET_LIBRARY_GEMM_FROM_C / PPNET_LIBRARY_GEMM_BELOW_C override, PPNET_LIBRARY_GEMM=1 = the library everywhere.
# Round 5: NO width takes the library by default.  With the LayerNorm-folded and accumulating epilogues on the 256 x 256 core
# (csrc/mfma_gemm.h EPI_LN_BIAS[_GELU] / EPI_ACCUM_STATS: the old stream read in the epilogue, both wave groups' epilogues side by side)
# the build's kernels on every level take 22.02 ms per SegNet batch against 22.55 with the round-4 gate at 512 and 22.93 with the
# vendor's GEMMs everywhere (alternating on one box, profiles/r05_ppnet_ab.txt): no `Cijk_*` kernel is left in a bf16 batch.
LIBRARY_GEMM_FROM_C = 1 << 30
LIBRARY_GEMM_BELOW_C = 0


def _library_width(C):
    if os.environ.get("PPNET_LIBRARY_GEMM"):
        return True
    return (C >= int(os.environ.get("PPNET_LIBRARY_GEMM_FROM_C", LIBRARY_GEMM_FROM_C))
            or C < int(os.environ.get("PPNET_LIBRARY_GEMM_BELOW_C", LIBRARY_GEMM_BELOW_C)))


def _own_gemm_ok(x2, lin):
    """The build's own MFMA GEMM (ppn_gemm_bf16) serves bfloat16 inference with K % 64 == 0, K >= 128, N % 8 == 0."""
    K, N = lin.in_features, lin.out_features
    return (x2.is_cuda and x2.dtype == torch.bfloat16 and lin.weight.dtype == torch.bfloat16 and K % 64 == 0 and K >= 128 and N % 8 == 0
            and x2.is_contiguous() and not fused.recording(x2, lin.weight) and not _library_width(min(K, N)))


def _bias32(lin):
    """A Linear's bias as float32 (zeros if none), cached on the module against its parameter's version."""
    c = lin.__dict__.get("_ppn_b32")
    if c is None:
        c = lin.__dict__["_ppn_b32"] = fused.WeightCache()
    return c.get((lin.bias, lin.weight), lambda: (lin.bias.detach().float().contiguous() if lin.bias is not None
                                                  else torch.zeros(lin.out_features, dtype=torch.float32, device=lin.weight.device)))


def _linear(x2, lin, gelu=False):
    """lin(x2) (+ erf GELU) for 2-D x2 on the build's own GEMM where it applies; the framework's call otherwise (float32, odd sizes)."""
    if _own_gemm_ok(x2, lin):
        return fused.gemm_bf16(x2, lin.weight.detach(), _bias32(lin), "bias_gelu" if gelu else "bias")
    if gelu and x2.is_cuda and x2.dtype == torch.bfloat16:
        return torch._addmm_activation(lin.bias, x2, lin.weight.t(), use_gelu=True)
    y = F.linear(x2, lin.weight, lin.bias)
    return F.gelu(y) if gelu else y


def _accumulate(s2, x2, lin):
    """s2 += x2 @ lin.weight^T in place (no bias: the folded layer carries it outside), own GEMM where it applies."""
    if _own_gemm_ok(x2, lin) and s2.is_contiguous():
        return fused.gemm_bf16(x2, lin.weight.detach(), None, "accum", out=s2)
    return s2.addmm_(x2, lin.weight.t())


def _mfma_weights(conv):
    """(weight as [Cout,3,3,Cin] bfloat16 — the k order of the implicit GEMM — and the bias as float32, zeros if none)."""
    w = conv.weight.detach().permute(0, 2, 3, 1).contiguous()
    b = conv.bias.detach().float().contiguous() if conv.bias is not None else torch.zeros(conv.out_channels, dtype=torch.float32, device=w.device)
    return w, b


class ConvTokenizer(nn.Module):
    def __init__(self, in_chans=3, embed_dim=96, norm_layer=None):
        super().__init__()
        self.proj = nn.Sequential(nn.Conv2d(in_chans, embed_dim // 2, 3, 2, 1), nn.Conv2d(embed_dim // 2, embed_dim, 3, 2, 1))
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None

    def forward(self, x):
        # channels_last in, channels_last out: the NHWC token tensor is a zero-copy view of the conv output
        x = self.proj(x.contiguous(memory_format=torch.channels_last)).permute(0, 2, 3, 1)
        return fused.layer_norm(x, self.norm) if self.norm is not None else x

    _codes = None         # WeightCache of (lut, second convolution's bias as float32, packed second conv, vectors) for forward_codes

    def takes_codes(self, grid_u8):
        c0, c1 = self.proj[0], self.proj[1]
        return (grid_u8.is_cuda and grid_u8.dtype == torch.uint8 and grid_u8.dim() == 3 and grid_u8.shape[1] % 2 == 0
                and grid_u8.shape[2] % 32 == 0 and c0.weight.shape == (64, 3, 3, 3) and c0.weight.dtype == torch.bfloat16
                and self.norm is not None and c1.bias is not None and not os.environ.get("PPNET_LIBRARY_TOKENIZER"))

    def forward_codes(self, grid_u8):
        """The tokens of the palette image of u8 occupancy codes [B,R,R] (what ppn_grid_to_image would render): the first
        convolution is a table product on the matrix cores (ppn_tokenizer_conv1_codes_bf16), the second convolution runs
        without its bias, which the LayerNorm kernel adds in registers — the normalised image and two bias passes are
        never written."""
        c0, c1 = self.proj[0], self.proj[1]
        if self._codes is None:
            self._codes = fused.WeightCache()
        lut, b2, w2p, vec = self._codes.get(
            (c0.weight, c0.bias, c1.weight, c1.bias, self.norm.weight, self.norm.bias),
            lambda: (fused.tokenizer_lut(c0, IMG_MEAN, IMG_STD).to(grid_u8.device), c1.bias.detach().float().contiguous())
            + fused.tokenizer_pack(c1, self.norm))
        if (c1.weight.shape == (128, 64, 3, 3) and grid_u8.shape[1] % 4 == 0 and grid_u8.shape[2] % 64 == 0
                and not os.environ.get("PPNET_TOKENIZER_TWO_KERNELS")):
            # both convolutions and the LayerNorm in one kernel (ppn_tokenizer_codes_bf16): no library convolution, no intermediate
            return fused.tokenizer_codes(grid_u8, lut, w2p, vec, self.norm.eps)
        x = fused.tokenizer_conv1_codes(grid_u8, lut).permute(0, 3, 1, 2)
        x = F.conv2d(x, c1.weight, None, c1.stride, c1.padding).permute(0, 2, 3, 1)
        return fused.layer_norm(x, self.norm, offset=b2)


class ConvDownsampler(nn.Module):
    def __init__(self, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.reduction = nn.Conv2d(dim, 2 * dim, 3, 2, 1, bias=False)
        self.norm = norm_layer(2 * dim)

    _mfma = None          # WeightCache of (weight [2C,3,3,C] bf16, zero bias float32) for the MFMA implicit-GEMM kernel

    def forward(self, x):                      # x [B,H,W,C] contiguous == a channels_last [B,C,H,W] view: no layout copies
        if _use_mfma_conv(x, self.reduction):
            if self._mfma is None:
                self._mfma = fused.WeightCache()
            w, b = self._mfma.get((self.reduction.weight,), lambda: _mfma_weights(self.reduction))
            y = fused.conv3x3_mfma(x.permute(0, 3, 1, 2), w, b, stride=2, relu=False)
            return fused.layer_norm(y.permute(0, 2, 3, 1), self.norm)
        return fused.layer_norm(self.reduction(x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1), self.norm)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)

    def _erf_gelu(self):
        return isinstance(self.act, nn.GELU) and self.act.approximate == "none"

    def hidden(self, x):
        """act(fc1(x)) as a 2-D [tokens, hidden] tensor: bias + GELU in the projection's epilogue (one pass over the hidden
        activations less; in bf16 at least as close to float32 erf-GELU as the two-kernel form, tools/gelu_epilogue_check.py)."""
        x2 = x.reshape(-1, x.shape[-1])
        if x.is_cuda and x.dtype == torch.bfloat16 and self._erf_gelu() and not fused.recording(x, self.fc1.weight):
            return _linear(x2.contiguous(), self.fc1, gelu=True)
        return self.act(self.fc1(x2))

    def forward(self, x):
        if x.is_cuda and x.dtype == torch.bfloat16 and self._erf_gelu() and not fused.recording(x, self.fc1.weight):
            return _linear(self.hidden(x), self.fc2).view(*x.shape[:-1], -1)
        return self.fc2(self.act(self.fc1(x)))


class NATLayer(nn.Module):
    def __init__(self, dim, num_heads, kernel_size=7, dilation=None, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                 drop=0.0, attn_drop=0.0, drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm, layer_scale=None):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = NeighborhoodAttention2D(dim, kernel_size=kernel_size, dilation=dilation, num_heads=num_heads,
                                            qkv_bias=qkv_bias, qk_scale=qk_scale)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), act_layer=act_layer)
        self.drop_path_rate = float(drop_path)                              # stochastic depth, training only (nat.py:122)
        self.layer_scale = layer_scale is not None and type(layer_scale) in (int, float)
        if self.layer_scale:
            self.gamma1 = nn.Parameter(layer_scale * torch.ones(dim))
            self.gamma2 = nn.Parameter(layer_scale * torch.ones(dim))

    folded = False        # set by NATBlock.fold(): LayerScale in the projection weights, biases carried as offsets
    _c = _c_dev = None

    def _forward_folded(self, s, y, next_norm, restore=False):
        """s: the residual stream minus the level's accumulated projection biases (see _fold_doc); y = norm1(s + c_in).
        restore (the level's last layer): the returned stream is the TRUE one, s + c_out — where the layer's own kernel can add the
        constant in its epilogue (the 128-channel streaming form) it does, and reports so by the third return value."""
        C = s.shape[-1]
        c_in, c_mid, c_out = self.offsets(s.device)
        if self._streams_c128(s):
            # 128-channel level: LN -> qkv and LN -> MLP -> residual are one token-streaming kernel each (weights in LDS)
            qkv = fused.nat128_ln_qkv(s, c_in, self.norm1, self.attn.qkv)
            fused.nat128_proj_add_(s, self.attn.attend(s, qkv=qkv), self.attn.proj)      # s += o W'^T (bias in c_mid)
            fused.nat128_ln_mlp_(s, c_mid, self.norm2, self.mlp.fc1, self.mlp.fc2, final_add=c_out if restore else None)
            off = None if restore else c_out
            return s, (fused.layer_norm(s, next_norm, offset=off) if next_norm is not None else None), restore
        if y is None:
            y = fused.layer_norm(s, self.norm1, offset=c_in)
        s2 = s.view(-1, C)
        _accumulate(s2, self.attn.attend(y).view(-1, C), self.attn.proj)                # s += o W'^T  (bias in c_mid)
        y2 = fused.layer_norm(s, self.norm2, offset=c_mid)
        _accumulate(s2, self.mlp.hidden(y2), self.mlp.fc2)                              # s += h W2'^T (bias in c_out)
        return s, (fused.layer_norm(s, next_norm, offset=c_out) if next_norm is not None else None), False

    _ln_packs = None

    def ln_packs(self):
        """What ppn_nat_gemm_bf16 reads for this (folded) layer, rebuilt when a parameter changes: the LayerNorms folded into the
        projections behind them — W' = W diag(gamma) (bfloat16), b' = b + W beta, colsum(W') of the bfloat16 values, float32 —
        for qkv and fc1, and proj / fc2 (LayerScale already folded by NATBlock.fold) with float32 biases:
        (wq, bq, csq, w1, b1, cs1, wp, bp, w2, b2)."""
        if self._ln_packs is None:
            self._ln_packs = fused.WeightCache()
        q, f1, pj, f2, n1, n2 = self.attn.qkv, self.mlp.fc1, self.attn.proj, self.mlp.fc2, self.norm1, self.norm2

        def build():
            out = []
            for lin, ln in ((q, n1), (f1, n2)):
                w32 = lin.weight.detach().float()
                wf = (w32 * ln.weight.detach().float()[None, :]).to(torch.bfloat16).contiguous()
                b = (lin.bias.detach().float() if lin.bias is not None else 0.0) + w32 @ ln.bias.detach().float()
                out += [wf, b.contiguous(), wf.float().sum(1).contiguous()]
            for lin in (pj, f2):
                out += [lin.weight.detach().contiguous(), lin.bias.detach().float().contiguous() if lin.bias is not None
                        else torch.zeros(lin.out_features, dtype=torch.float32, device=lin.weight.device)]
            return tuple(out)
        return self._ln_packs.get((q.weight, q.bias, f1.weight, f1.bias, pj.weight, pj.bias, f2.weight, f2.bias, n1.weight, n1.bias,
                                   n2.weight, n2.bias), build)

    _mlp_packs = None

    def mlp_packs(self):
        """What ppn_nat_mlp_bf16 reads for this (folded) layer's MLP: (packed weights, hb [hidden, 2] = (colsum, folded bias),
        b2) — built from ln_packs() on the device, rebuilt when a parameter changes."""
        if self._mlp_packs is None:
            self._mlp_packs = fused.WeightCache()
        f1, f2, n2 = self.mlp.fc1, self.mlp.fc2, self.norm2

        def build():
            _, _, _, w1, b1, cs1, _, _, w2, b2 = self.ln_packs()
            return fused.nat_mlp_pack(w1, w2), torch.stack([cs1, b1], dim=1).contiguous(), b2
        return self._mlp_packs.get((f1.weight, f1.bias, f2.weight, f2.bias, n2.weight, n2.bias), build)

    def _streams_c128(self, s):
        return (s.shape[-1] == 128 and s.is_cuda and s.dtype == torch.bfloat16 and (s.numel() // 128) % 16 == 0
                and self.mlp.fc1.out_features == 256 and isinstance(self.mlp.act, nn.GELU) and self.mlp.act.approximate == "none"
                and self.attn.qkv.weight.dtype == torch.bfloat16 and not os.environ.get("PPNET_LIBRARY_NAT128"))

    def offsets(self, device):
        """(c_in, c_mid, c_out) as float32 tensors on `device`: plain attributes, not buffers, so that module.to(bfloat16)
        does not round the accumulated biases."""
        if self._c_dev is None or self._c_dev[0].device != device:
            self._c_dev = tuple(t.to(device) for t in self._c)
        return self._c_dev

    def forward(self, x, y=None, next_norm=None, next_pad=None):
        """x: residual stream [B,H,W,C]; y = norm1(x) if the caller already has it. Returns (x', next_norm(x')).
        The attention's zero-padding to kernel*dilation is virtual (na.NeighborhoodAttention2D.forward), so next_pad
        stays None; the argument is kept for a caller that wants the materialised padded grid.
        Residual add, LayerScale and the following LayerNorm are one fused kernel each (DropPath is the identity
        at inference, nat.py:140-153)."""
        if self.folded:
            return self._forward_folded(x, y, next_norm)[:2]
        hw = (x.shape[1], x.shape[2])
        if y is None:
            y = fused.layer_norm(x, self.norm1)
        real = hw if (y.shape[1], y.shape[2]) != hw else None               # a materialised padded y still works
        dp = self.drop_path_rate if self.training else 0.0                  # x + drop_path(gamma * f(.)): the mask commutes with gamma
        x, y2 = fused.residual_layer_norm(x, drop_path(self.attn(y, real), dp, self.training), self.gamma1 if self.layer_scale else None, self.norm2)
        return fused.residual_layer_norm(x, drop_path(self.mlp(y2), dp, self.training), self.gamma2 if self.layer_scale else None, next_norm, next_pad)


def _fold_doc():
    """Folded inference form of a NAT level (SegNet.prepare_inference on the GPU path).  LayerScale is folded into the two
    output projections (W' = diag(gamma) W, b' = gamma * b), so a sub-layer is x' = x + o W'^T + b'.  The residual stream
    is kept WITHOUT the constant part: s = x - c, where c is the sum of the b' seen so far in the level (known when the
    weights are).  Then s' = s + o W'^T is ONE library GEMM accumulating into s (beta = 1: `addmm_`), every LayerNorm is
    LN(s + c) with c added in registers (ppn_layernorm_offset), and the level's end adds c once for the downsampler.
    Per sub-layer the activations cross HBM 4 times (GEMM reads s, writes s'; LN reads s', writes y) instead of 5
    (GEMM writes a; the fused residual+LN kernel reads x and a, writes x' and y)."""


class NATBlock(nn.Module):
    def __init__(self, dim, depth, num_heads, kernel_size, dilations=None, downsample=True, mlp_ratio=4.0, qkv_bias=True,
                 qk_scale=None, drop=0.0, attn_drop=0.0, drop_path=0.0, norm_layer=nn.LayerNorm, layer_scale=None):
        super().__init__()
        self.blocks = nn.ModuleList(
            NATLayer(dim, num_heads, kernel_size, None if dilations is None else dilations[i], mlp_ratio, qkv_bias,
                     qk_scale, drop_path=drop_path[i] if isinstance(drop_path, (list, tuple)) else drop_path,
                     norm_layer=norm_layer, layer_scale=layer_scale) for i in range(depth))
        self.downsample = ConvDownsampler(dim, norm_layer) if downsample else None

    def forward(self, x, out_norm=None, inplace=False):
        """Returns (next level's input, out_norm(x) or x): the level's output norm rides on the last fused kernel.
        The fused kernels update the residual stream in place: inplace=True lets them use the caller's tensor (NAT hands
        over the tokenizer's / downsampler's fresh output), otherwise it is copied first."""
        if not inplace:
            x = x.clone()
        if self._ln_folded_ok(x):
            return self._forward_ln_folded(x, out_norm)
        y = None
        n = len(self.blocks)
        hw = (x.shape[1], x.shape[2])
        for i, blk in enumerate(self.blocks):
            if i + 1 < n:
                nxt = self.blocks[i + 1]
                x, y = blk(x, y, None if (nxt.folded and nxt._streams_c128(x)) else nxt.norm1, None)
            elif blk.folded:
                # x is s = x_true - c: the true stream is read itself by the downsampler (or returned when there is no output norm)
                want = self.downsample is not None or out_norm is None
                x, y, restored = blk._forward_folded(x, y, out_norm, restore=want)
                if want and not restored:
                    x = fused.bias_act_(x.permute(0, 3, 1, 2), blk.offsets(x.device)[2], 1.0).permute(0, 2, 3, 1)
            else:
                x, y = blk(x, y, out_norm, None)
        xo = y if out_norm is not None else x
        return (x, xo) if self.downsample is None else (self.downsample(x), xo)

    def _ln_folded_ok(self, x):
        """The level runs on ppn_nat_gemm_bf16 (csrc/nat_gemm.hip): folded bfloat16 inference, C and the MLP width multiples of
        256, whole 256-token tiles — and at least 64 of them in the narrowest projection (tokens x C): the persistent kernels walk
        256 x 256 tiles one per CU, and a batch of 1-16 problems has a handful (a level-2 projection at batch 1 is TWO tiles, each
        walking K alone: 34 us where the wave-per-block kernel of gemm_small.hip behind a LayerNorm launch takes 12)."""
        b0 = self.blocks[0]
        C = x.shape[-1]
        return (b0.folded and x.is_cuda and x.dtype == torch.bfloat16 and b0.attn.qkv.weight.dtype == torch.bfloat16 and C % 256 == 0
                and (x.numel() // C) % 256 == 0 and (x.numel() // C // 256) * (C // 256) >= int(os.environ.get("PPNET_SMALL_GEMM_TILES", "64")) and b0.mlp.fc1.out_features % 256 == 0 and b0.mlp._erf_gelu() and x.is_contiguous()
                and not torch.is_grad_enabled() and not _library_width(C) and not os.environ.get("PPNET_NO_LN_FOLD"))

    def _forward_ln_folded(self, x, out_norm):
        """The level with the dense half of every layer on the build's own persistent GEMMs (reference SegNet/nat.py:140-153): per
        layer four launches and the attention —
            qkv = GEMM_ln(s)            LayerNorm folded into the projection: the GEMM reads the raw residual stream and the row
                                        sums the previous accumulating GEMM left behind
            a   = NA(qkv)
            s  += a Wp'^T + bp'         in place, residual add in the matrix pipe, row sums of the new s out
            h   = GEMM_ln_gelu(s)       LayerNorm folded in, erf-GELU in the epilogue
            s  += h W2'^T + b2'
        — no LayerNorm kernel, no separate residual / bias / activation pass, no vendor GEMM.  s is the TRUE residual stream (the
        biases are added in the epilogues), so the downsampler and the output norm read it as it is."""
        B, H, W, C = x.shape
        M = B * H * W
        s2 = x.view(M, C)
        st = fused.row_stats(s2)                                            # the level's first stream came from a LayerNorm kernel
        P = fused.nat_partials(C)
        st_mid = torch.empty(P, M, 2, dtype=torch.float32, device=x.device)
        st_out = torch.empty(P, M, 2, dtype=torch.float32, device=x.device)
        fused_mlp = (P == C // 128 and fused.nat_mlp_ok(M, C, self.blocks[0].mlp.fc1.out_features) and not os.environ.get("PPNET_NO_FUSED_MLP"))
        for blk in self.blocks:
            wq, bq, csq, w1, b1, cs1, wp, bp, w2, b2 = blk.ln_packs()
            qkv = torch.empty(B, H, W, 3 * C, dtype=x.dtype, device=x.device)
            fused.nat_gemm(s2, wq, bq, "ln", qkv.view(M, 3 * C), colsum=csq, stats_in=st, eps=blk.norm1.eps)
            a = blk.attn.attend(x, qkv=qkv)
            fused.nat_gemm(a.view(M, C), wp, bp, "acc", s2, stats_out=st_mid)
            if fused_mlp:
                # LN -> fc1 -> GELU -> fc2 -> residual as ONE kernel: the hidden activation never reaches HBM (csrc/nat_mlp.hip)
                wpk, hb, b2v = blk.mlp_packs()
                fused.nat_mlp_(s2, wpk, hb, b2v, w1.shape[0], stats_out=st_out, eps=blk.norm2.eps)
            else:
                h = torch.empty(M, w1.shape[0], dtype=x.dtype, device=x.device)
                fused.nat_gemm(s2, w1, b1, "ln_gelu", h, colsum=cs1, stats_in=st_mid, eps=blk.norm2.eps)
                fused.nat_gemm(h, w2, b2, "acc", s2, stats_out=st_out)
            st = st_out
        xo = fused.layer_norm(x, out_norm) if out_norm is not None else x
        return (x, xo) if self.downsample is None else (self.downsample(x), xo)

    def fold(self):
        """See _fold_doc.  After the checkpoint is loaded; float32 algebra, then back to the parameters' dtype."""
        c = None
        for blk in self.blocks:
            g1 = blk.gamma1.detach().float() if blk.layer_scale else None
            g2 = blk.gamma2.detach().float() if blk.layer_scale else None
            for lin, g in ((blk.attn.proj, g1), (blk.mlp.fc2, g2)):
                if g is not None:
                    lin.weight = nn.Parameter((lin.weight.detach().float() * g[:, None]).to(lin.weight.dtype))
                    lin.bias = nn.Parameter((lin.bias.detach().float() * g).to(lin.bias.dtype))
            dim = blk.attn.proj.bias.shape[0]
            zero = torch.zeros(dim, dtype=torch.float32, device=blk.attn.proj.bias.device)
            c_in = c if c is not None else zero
            c_mid = c_in + blk.attn.proj.bias.detach().float()
            c_out = c_mid + blk.mlp.fc2.bias.detach().float()
            blk._c, blk._c_dev = (c_in.clone().contiguous(), c_mid.clone().contiguous(), c_out.clone().contiguous()), None
            if blk.layer_scale:
                blk.gamma1 = nn.Parameter(torch.ones_like(blk.gamma1)); blk.gamma2 = nn.Parameter(torch.ones_like(blk.gamma2))
            blk.folded = True
            c = c_out
        return self


class NAT(nn.Module):
    def __init__(self, embed_dim, mlp_ratio, depths, num_heads, drop_path_rate=0.2, in_chans=3, kernel_size=7,
                 dilations=None, out_indices=(0, 1, 2, 3), qkv_bias=True, qk_scale=None, drop_rate=0.0,
                 attn_drop_rate=0.0, norm_layer=nn.LayerNorm, frozen_stages=-1, pretrained=None, layer_scale=None,
                 **kwargs):
        super().__init__()
        self.num_levels = len(depths)
        self.embed_dim = embed_dim
        self.num_features = [int(embed_dim * 2 ** i) for i in range(self.num_levels)]
        self.patch_embed = ConvTokenizer(in_chans, embed_dim, norm_layer)
        dpr = [float(v) for v in torch.linspace(0, drop_path_rate, sum(depths))]      # nat.py:247
        self.levels = nn.ModuleList(
            NATBlock(int(embed_dim * 2 ** i), depths[i], num_heads[i], kernel_size,
                     None if dilations is None else dilations[i], downsample=(i < self.num_levels - 1),
                     mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                     drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])], norm_layer=norm_layer,
                     layer_scale=layer_scale) for i in range(self.num_levels))
        self.out_indices = tuple(out_indices)
        self.compute_indices = tuple(out_indices)      # inference may narrow this to the levels the head reads
        for i in out_indices:
            self.add_module(f"norm{i}", norm_layer(self.num_features[i]))
        if isinstance(pretrained, str):
            self.init_weights(pretrained)

    def init_weights(self, pretrained=None):
        if isinstance(pretrained, str):
            sd = torch.load(pretrained, map_location="cpu", weights_only=True)
            sd = sd.get("state_dict", sd.get("model", sd))
            self.load_state_dict(sd, strict=False)

    def forward(self, x):
        """x: the image [B,3,H,W], or (GPU inference) the u8 occupancy codes [B,H,W] it would be rendered from."""
        x = self.patch_embed.forward_codes(x) if x.dtype == torch.uint8 else self.patch_embed(x)
        outs = [None] * len(self.out_indices)          # one slot per out_index (nat.py:326-332); levels nobody reads stay None
        for idx, level in enumerate(self.levels):
            want = idx in self.compute_indices
            x, xo = level(x, getattr(self, f"norm{idx}") if want else None, inplace=True)   # x: fresh LayerNorm output
            if want:
                outs[self.out_indices.index(idx)] = xo.permute(0, 3, 1, 2)    # [B,C,H,W] in channels_last memory format (zero-copy view)
        return outs


class DiNAT(NAT):
    """DiNAT is NAT with per-layer dilations (dinat.py:15-22)."""


class _ConvModule(nn.Sequential):
    """mmcv ConvModule(conv -> bn -> ReLU) with its parameter names `conv.*`, `bn.*` (conv has no bias under a norm)."""

    def __init__(self, cin, cout, k, dilation=1):
        super().__init__()
        self.add_module("conv", nn.Conv2d(cin, cout, k, 1, ((k - 1) // 2) * dilation, dilation, bias=False))
        self.add_module("bn", nn.BatchNorm2d(cout))          # SyncBN reverts to BN outside distributed runs (SegNet/train.py:179-185)
        self.add_module("activate", nn.ReLU(inplace=True))


class _Upsample(nn.Module):
    def __init__(self, scale_factor, align_corners=False):
        super().__init__()
        self.scale_factor, self.align_corners = float(scale_factor), align_corners

    def forward(self, x, relu=False, bias=None):
        if self.scale_factor == 2.0 and not self.align_corners and x.shape[1] % 8 == 0 and x.is_cuda:
            return fused.upsample2x_nhwc(x, relu, bias)                      # HIP kernel, bias + ReLU folded into the loads
        if bias is not None:
            x = x + bias.view(1, -1, 1, 1)
        if relu:
            x = F.relu(x)
        size = [int(t * self.scale_factor) for t in x.shape[-2:]]            # mmseg/ops/wrappers.py:43-51
        return F.interpolate(x, size, None, "bilinear", self.align_corners)


class SETRUPHead(nn.Module):
    def __init__(self, in_channels=1024, channels=512, num_classes=2, num_convs=1, up_scale=4, kernel_size=3,
                 in_index=-1, dropout_ratio=0.1, align_corners=False, norm_layer=None, norm_cfg=None, **kwargs):
        super().__init__()
        assert kernel_size in (1, 3)
        self.in_index, self.align_corners = in_index, align_corners
        self.norm = nn.LayerNorm(in_channels, eps=1e-6)
        self.up_convs = nn.ModuleList()
        cin = in_channels
        for _ in range(num_convs):
            self.up_convs.append(nn.Sequential(_ConvModule(cin, channels, kernel_size), _Upsample(up_scale, align_corners)))
            cin = channels
        self.conv_seg = nn.Conv2d(channels, num_classes, 1)
        self.dropout = nn.Dropout2d(dropout_ratio) if dropout_ratio > 0 else nn.Identity()   # decode_head.py cls_seg; identity in eval

    def forward(self, inputs, lowres=False):
        """lowres=True: the classifier's logits BEFORE the last x2 up-sampling (the caller fuses the rest of the tail)."""
        x = inputs[self.in_index]
        # LayerNorm over channels (setr_up_head.py:73-76) on the NHWC view; stays channels_last for the convolutions
        x = fused.layer_norm(x.permute(0, 2, 3, 1), self.norm).permute(0, 3, 1, 2)
        prepared = all(isinstance(up[0].bn, nn.Identity) and up[0].conv.bias is not None for up in self.up_convs)
        if prepared and all(_use_mfma_conv(x, up[0].conv) for up in self.up_convs) and self.conv_seg.out_channels == 2:
            return self._forward_mfma(x, lowres)
        for up in self.up_convs[:-1]:
            cm = up[0]
            if isinstance(cm.bn, nn.Identity) and cm.conv.bias is not None:  # prepared: the folded-BN bias rides in the upsample kernel
                c = cm.conv
                x = up[1](F.conv2d(x, c.weight, None, c.stride, c.padding), relu=True, bias=c.bias)
            else:
                x = up[1](cm.bn(cm.conv(x)), relu=True)                      # conv -> BN -> ReLU + x2 bilinear in one kernel
        # last stage: conv_seg is a 1x1 convolution and bilinear interpolation is linear with weights summing to 1,
        # so conv_seg(upsample(y)) == upsample(conv_seg(y)): classify at the low resolution and upsample 2 channels
        # instead of `channels` (the reference materialises a [B,512,R/2,R/2] tensor here, setr_up_head.py:78-80)
        conv, up = self.up_convs[-1][0], self.up_convs[-1][1]
        if isinstance(conv.bn, nn.Identity) and conv.conv.bias is not None and x.is_cuda:
            c = conv.conv                                                     # prepared: bias + ReLU in one in-place HIP pass
            y = fused.bias_act_(F.conv2d(x, c.weight, None, c.stride, c.padding).contiguous(memory_format=torch.channels_last), c.bias, 0.0)
        else:
            y = conv(x)
        if self.training:
            # the reference's order (setr_up_head.py:78-80, decode_head.py:232-237): up-sample, channel dropout, classify —
            # the dropout mask does not commute with the interpolation
            return self.conv_seg(self.dropout(up(y)))
        lo = self.conv_seg(y).contiguous()
        return lo if lowres else up(lo)                                      # 2 channels: the library bilinear kernel


    _mfma = None

    def _forward_mfma(self, x, lowres):
        """Prepared bfloat16 inference on the hand-written MFMA kernels: every ConvModule is one implicit-GEMM launch with the
        folded-BatchNorm bias and the ReLU in its epilogue, and the last one also applies the 1x1 classifier (commuted in front
        of the last up-sampling, as in forward()) so the 512-channel activation at the highest resolution is never written."""
        if self._mfma is None:
            self._mfma = fused.WeightCache()
        cs = self.conv_seg
        src = [t for up in self.up_convs for t in (up[0].conv.weight, up[0].conv.bias)] + [cs.weight, cs.bias]
        packs = self._mfma.get(src, lambda: [_mfma_weights(up[0].conv) for up in self.up_convs] +
                               [(cs.weight.detach().float().reshape(cs.out_channels, -1).contiguous(), cs.bias.detach().float().contiguous())])
        # The convolution kernels address their input with 32-bit byte offsets: the last stage reads [B, channels, H 2^(n-1), W 2^(n-1)]
        # bfloat16, which passes 4 GiB at batch 256 of 512 x 512 maps.  Larger batches go through the head in slices.
        B = x.shape[0]
        last_in = self.up_convs[-1][0].conv.in_channels * x.shape[-2] * x.shape[-1] * 4 ** (len(self.up_convs) - 1) * 2
        bmax = max(1, (2 ** 32 - 1) // last_in)
        if B > bmax:
            step = -(-B // (-(-B // bmax)))                                   # equal slices
            return torch.cat([self._forward_mfma(x[i:i + step], lowres) for i in range(0, B, step)], dim=0)
        for i, up in enumerate(self.up_convs[:-1]):
            x = up[1](fused.conv3x3_mfma(x, packs[i][0], packs[i][1], stride=1, relu=True))
        w2, b2 = packs[-1]
        n = len(self.up_convs) - 1
        lo = fused.conv3x3_relu_classify2(x, packs[n][0], packs[n][1], w2, b2).to(x.dtype).contiguous()
        return lo if lowres else self.up_convs[-1][1](lo)


class UPerHead(nn.Module):
    """UPerNet head (SegNet/mmseg/decode_heads/uper_head.py:12-127 + psp_head.py:10-60): pyramid pooling on the last level,
    lateral 1x1 convs, top-down bilinear fusion, 3x3 FPN convs, concatenation, 3x3 bottleneck, 1x1 classifier.  Checkpoint
    keys follow mmseg: `psp_modules.i.1.{conv,bn}`, `bottleneck.{conv,bn}`, `lateral_convs.i.{conv,bn}`,
    `fpn_convs.i.{conv,bn}`, `fpn_bottleneck.{conv,bn}`, `conv_seg`.  The head of the reference's default SegNet config
    (SegNet/test.py:29-32 -> configs/nat/upernet_nat_base.py)."""

    def __init__(self, in_channels=(128, 256, 512, 1024), channels=64, num_classes=2, pool_scales=(1, 2, 3, 6),
                 in_index=(0, 1, 2, 3), dropout_ratio=0.1, align_corners=False, norm_cfg=None, **kwargs):
        super().__init__()
        self.in_index, self.align_corners = tuple(in_index), align_corners
        self.psp_modules = nn.ModuleList(
            nn.Sequential(nn.AdaptiveAvgPool2d(ps), _ConvModule(in_channels[-1], channels, 1)) for ps in pool_scales)
        self.bottleneck = _ConvModule(in_channels[-1] + len(pool_scales) * channels, channels, 3)
        self.lateral_convs = nn.ModuleList(_ConvModule(c, channels, 1) for c in in_channels[:-1])
        self.fpn_convs = nn.ModuleList(_ConvModule(channels, channels, 3) for _ in in_channels[:-1])
        self.fpn_bottleneck = _ConvModule(len(in_channels) * channels, channels, 3)
        self.conv_seg = nn.Conv2d(channels, num_classes, 1)                  # Dropout2d is the identity at inference

    def _resize(self, x, size):
        if (x.is_cuda and not self.align_corners and tuple(size) == (2 * x.shape[2], 2 * x.shape[3]) and x.shape[1] % 8 == 0
                and x.dtype in (torch.float32, torch.bfloat16) and not fused.recording(x)):
            return fused.upsample2x_nhwc(x)                                  # the FPN's x2 steps: the build's NHWC kernel
        return F.interpolate(x, size=size, mode="bilinear", align_corners=self.align_corners)

    _packs = None

    def _prepared_mfma(self, x):
        """Prepared bfloat16 inference on the build's own kernels: BatchNorm folded into every ConvModule (a bias on its conv)."""
        cms = [m for m in self.modules() if isinstance(m, _ConvModule)]
        return (x.is_cuda and x.dtype == torch.bfloat16 and not self.training and all(isinstance(c.bn, nn.Identity) and c.conv.bias is not None for c in cms)
                and self.conv_seg.out_channels == 2 and not fused.recording(x, self.conv_seg.weight) and not os.environ.get("PPNET_LIBRARY_CONV"))

    def _forward_mfma(self, inputs):
        """uper_head.py:76-127 with every convolution on the hand-written kernels: 1x1 ConvModules (laterals, pyramid pooling) are
        ppn_gemm_bf16 over the NHWC tokens with bias + ReLU in the epilogue, 3x3 ConvModules the implicit-GEMM kernel
        (ppn_conv3x3_mfma_bf16), the last one fused with the 1x1 classifier (ppn_conv3x3_relu_classify2_bf16: the 64-channel
        activation at the highest resolution is never written); the FPN's top-down step, the resize + concatenation of its outputs
        and the pyramid pooling module's pools and output assembly on NHWC kernels (ppn_upsample2x_add_nhwc, ppn_resize_concat_nhwc,
        ppn_adaptive_pools_nhwc).  Nothing of the head runs on framework kernels but the 1x1 ConvModule of a pool scale with fewer
        than 256 pooled positions in the batch."""
        if self._packs is None:
            self._packs = fused.WeightCache()
        cms = [m[1] for m in self.psp_modules] + [self.bottleneck] + list(self.lateral_convs) + list(self.fpn_convs) + [self.fpn_bottleneck]
        src = [t for c in cms for t in (c.conv.weight, c.conv.bias)] + [self.conv_seg.weight, self.conv_seg.bias]

        def build():
            pk = {}
            for c in cms:
                cv = c.conv
                if cv.kernel_size == (1, 1):
                    pk[c] = (cv.weight.detach().reshape(cv.out_channels, cv.in_channels).contiguous(), cv.bias.detach().float().contiguous())
                else:
                    pk[c] = _mfma_weights(cv)
            cs = self.conv_seg
            pk["seg"] = (cs.weight.detach().float().reshape(cs.out_channels, -1).contiguous(), cs.bias.detach().float().contiguous())
            return pk
        pk = self._packs.get(src, build)

        def conv1(cm, t):                                                    # 1x1 ConvModule on a channels_last [B,C,H,W] tensor
            Bn, Cc, Hh, Ww = t.shape
            tok = t.permute(0, 2, 3, 1).reshape(-1, Cc)
            w, b = pk[cm]
            if tok.shape[0] >= 256 and Cc % 64 == 0 and Cc >= 128 and tok.is_contiguous():
                y = fused.gemm_bf16(tok, w, b, "bias_relu")
            else:                                                            # the pyramid's 1 .. 36 pooled positions per image: too few rows for a tile
                y = F.relu(F.linear(tok, w, b.to(tok.dtype)))
            return y.view(Bn, Hh, Ww, -1).permute(0, 3, 1, 2)

        def conv3(cm, t, relu=True):
            w, b = pk[cm]
            if _use_mfma_conv(t, cm.conv, narrow=True):
                return fused.conv3x3_mfma(t, w, b, stride=1, relu=relu)
            y = F.conv2d(t, cm.conv.weight, cm.conv.bias, 1, 1)
            return F.relu(y) if relu else y
        inputs = [inputs[i] for i in self.in_index]
        x = inputs[-1]
        own_resize = not self.align_corners and self.conv_seg.in_channels % 8 == 0 and not os.environ.get("PPNET_UPER_UNFUSED_RESIZE")
        scales = [m[0].output_size if isinstance(m[0].output_size, int) else m[0].output_size[0] for m in self.psp_modules]
        if own_resize and len(scales) <= 4 and x.shape[1] % 8 == 0:
            # the pyramid pooling module (psp_head.py:48-60) as 6 launches: every pool in one kernel, a 1x1 ConvModule each on the
            # GEMM kernel (B s^2 rows), the resizes back + the concatenation with x in one kernel
            pooled = fused.adaptive_pools(x, scales)
            psp = fused.resize_concat([x] + [conv1(m[1], t) for m, t in zip(self.psp_modules, pooled)])
        else:
            psp = torch.cat([x] + [self._resize(conv1(m[1], m[0](x)), x.shape[2:]) for m in self.psp_modules], dim=1).contiguous(memory_format=torch.channels_last)
        laterals = [conv1(cm, inputs[i]) for i, cm in enumerate(self.lateral_convs)] + [conv3(self.bottleneck, psp)]
        for i in range(len(laterals) - 1, 0, -1):
            fine, coarse = laterals[i - 1], laterals[i]
            if (own_resize and fine.shape[2] == 2 * coarse.shape[2] and fine.shape[3] == 2 * coarse.shape[3]
                    and fine.permute(0, 2, 3, 1).is_contiguous()):
                fused.upsample2x_add_(fine, coarse)                         # the resize and the sum: one kernel, in place
            else:
                laterals[i - 1] = fine + self._resize(coarse, fine.shape[2:])
        outs = [conv3(self.fpn_convs[i], laterals[i].contiguous(memory_format=torch.channels_last)) for i in range(len(laterals) - 1)] + [laterals[-1]]
        if len(outs) == 4 and own_resize:
            cat = fused.resize_concat(outs)                                 # the three resizes + the concatenation: one kernel
        else:
            outs = [outs[0]] + [self._resize(o, outs[0].shape[2:]) for o in outs[1:]]
            cat = torch.cat(outs, dim=1).contiguous(memory_format=torch.channels_last)
        fb = self.fpn_bottleneck
        if _use_mfma_conv(cat, fb.conv, narrow=True):
            w, b = pk[fb]
            w2, b2 = pk["seg"]
            return fused.conv3x3_relu_classify2(cat, w, b, w2, b2).to(cat.dtype)
        return self.conv_seg(conv3(fb, cat))

    def forward(self, inputs):
        if self._prepared_mfma(inputs[self.in_index[-1]]):
            return self._forward_mfma(inputs)
        inputs = [inputs[i] for i in self.in_index]
        x = inputs[-1]
        psp = torch.cat([x] + [self._resize(m(x), x.shape[2:]) for m in self.psp_modules], dim=1)
        laterals = [conv(inputs[i]) for i, conv in enumerate(self.lateral_convs)] + [self.bottleneck(psp)]
        for i in range(len(laterals) - 1, 0, -1):
            laterals[i - 1] = laterals[i - 1] + self._resize(laterals[i], laterals[i - 1].shape[2:])
        outs = [self.fpn_convs[i](laterals[i]) for i in range(len(laterals) - 1)] + [laterals[-1]]
        outs = [outs[0]] + [self._resize(o, outs[0].shape[2:]) for o in outs[1:]]
        return self.conv_seg(self.fpn_bottleneck(torch.cat(outs, dim=1)))


class FCNHead(nn.Module):
    """mmseg's FCNHead (SegNet/mmseg/decode_heads/fcn_head.py:11-81 over decode_head.py:54-107,224-229) — the auxiliary head of
    the reference's NAT training configs (configs/_base_/models/nat.py:22-35, configs/nat/setr_up_nat_base.py:39-42: level 2,
    512 -> 256 channels, one 3x3 conv-BN-ReLU, Dropout2d(0.1), 1x1 classifier, loss weight 0.4).  Checkpoint keys follow mmseg:
    `convs.i.{conv,bn}`, `conv_cat.{conv,bn}`, `conv_seg`.  Training only: inference never evaluates it (encoder_decoder.py:63-80)."""

    def __init__(self, in_channels=256, channels=256, num_classes=19, num_convs=2, kernel_size=3, concat_input=True, dilation=1,
                 in_index=-1, dropout_ratio=0.1, align_corners=False, norm_cfg=None, loss_decode=None, **kwargs):
        super().__init__()
        assert num_convs >= 0 and dilation > 0
        self.in_index, self.align_corners, self.concat_input = in_index, align_corners, concat_input
        self.loss_weight = float((loss_decode or {}).get("loss_weight", 1.0))
        if num_convs == 0:
            assert in_channels == channels
            self.convs = nn.Identity()
        else:
            self.convs = nn.Sequential(*[_ConvModule(in_channels if i == 0 else channels, channels, kernel_size, dilation)
                                         for i in range(num_convs)])
        if concat_input:
            self.conv_cat = _ConvModule(in_channels + channels, channels, kernel_size)
        self.conv_seg = nn.Conv2d(channels, num_classes, 1)
        self.dropout = nn.Dropout2d(dropout_ratio) if dropout_ratio > 0 else nn.Identity()

    def forward(self, inputs):
        x = inputs[self.in_index]
        y = self.convs(x)
        if self.concat_input:
            y = self.conv_cat(torch.cat([x, y], dim=1))
        return self.conv_seg(self.dropout(y))


NAT_BASE_UPER = dict(   # SegNet/configs/nat/upernet_nat_base.py:6-34 (the default config of SegNet/test.py:29-32)
    backbone=dict(embed_dim=128, mlp_ratio=2.0, depths=[3, 4, 18, 5], num_heads=[4, 8, 16, 32], kernel_size=7,
                  layer_scale=1e-5),
    decode_head=dict(type="UPerHead", in_channels=[128, 256, 512, 1024], in_index=[0, 1, 2, 3], pool_scales=(1, 2, 3, 6),
                     channels=64, num_classes=2))

DINAT_BASE = dict(   # SegNet/configs/dinat/dinat_base.py:5-24 over _base_/models/dinat.py:3-46
    backbone=dict(embed_dim=128, mlp_ratio=2.0, depths=[3, 4, 18, 5], num_heads=[4, 8, 16, 32], kernel_size=7,
                  layer_scale=1e-5,
                  dilations=[[1, 16, 1], [1, 4, 1, 8], [1, 2, 1, 3, 1, 4, 1, 2, 1, 3, 1, 4, 1, 2, 1, 3, 1, 4], [1, 2, 1, 2, 1]]),
    decode_head=dict(in_channels=1024, channels=512, num_convs=4, up_scale=2, num_classes=2, kernel_size=3))

IMG_MEAN = (123.675, 116.28, 103.53)       # _base_/datasets/planning_seg.py:12-13
IMG_STD = (58.395, 57.12, 57.375)


class SegNet(nn.Module):
    """EncoderDecoder(test_cfg=mode 'whole') (SegNet/mmseg/models/segmentors/encoder_decoder.py:16-265, base.py:62-112): the
    constructor takes the reference's config sub-dicts (`from_config` the whole `model=dict(...)`), `forward` the reference
    harness's call — `model(return_loss=False, img=[x], img_metas=[[...]]) -> list[np.ndarray int64 [H,W]]`
    (mmseg/apis/test.py:93) — and `model(img=x, img_metas=[...], gt_semantic_seg=y)` returns the loss dict of forward_train."""

    def __init__(self, backbone=None, decode_head=None, auxiliary_head=None, train_cfg=None, test_cfg=None, pretrained=None):
        super().__init__()
        bb_cfg = dict(backbone or DINAT_BASE["backbone"])
        bb_type = bb_cfg.pop("type", "DiNAT")
        if pretrained is not None and bb_cfg.get("pretrained") is None:
            bb_cfg["pretrained"] = pretrained                                  # encoder_decoder.py:32-36
        self.backbone = {"NAT": NAT, "DiNAT": DiNAT}[bb_type](**bb_cfg)
        head_cfg = dict(decode_head or DINAT_BASE["decode_head"])
        head_type = head_cfg.pop("type", "SETRUPHead")
        self.decode_head = {"SETRUPHead": SETRUPHead, "UPerHead": UPerHead, "FCNHead": FCNHead}[head_type](**head_cfg)
        self.auxiliary_head = None
        if auxiliary_head is not None:                                         # encoder_decoder.py:52-61 (a dict, or a list of them)
            mk = lambda c: FCNHead(**{k: v for k, v in dict(c).items() if k != "type"})
            self.auxiliary_head = nn.ModuleList(mk(c) for c in auxiliary_head) if isinstance(auxiliary_head, (list, tuple)) else mk(auxiliary_head)
        self.align_corners = self.decode_head.align_corners
        self.train_cfg, self.test_cfg = train_cfg, dict(test_cfg or {"mode": "whole"})
        if self.test_cfg.get("mode", "whole") != "whole":
            raise NotImplementedError("test_cfg.mode 'whole' only (every configuration under SegNet/configs)")
        self._narrow_levels()

    def _aux_heads(self):
        a = self.auxiliary_head
        return [] if a is None else (list(a) if isinstance(a, nn.ModuleList) else [a])

    def _narrow_levels(self, inference=False):
        """The backbone evaluates only the levels a head reads (their output norm + the NHWC -> NCHW view): SETR-UP reads the last
        one; the auxiliary head's level joins while it can be trained (not after prepare_inference)."""
        outs = tuple(self.backbone.out_indices)        # a head's in_index selects a SLOT of the backbone's output list

        def levels(ii):
            ii = ii if isinstance(ii, (tuple, list)) else (ii,)
            for i in ii:
                if not -len(outs) <= i < len(outs):
                    raise ValueError(f"head in_index {i} outside the backbone's {len(outs)} outputs (out_indices {outs})")
            return {outs[i] for i in ii}

        need = levels(self.decode_head.in_index)
        if not inference:
            for h in self._aux_heads():
                need |= levels(h.in_index)
        self.backbone.compute_indices = tuple(sorted(need))

    @classmethod
    def from_config(cls, cfg):
        """cfg: the reference's `model = dict(type='EncoderDecoder', pretrained=..., backbone=dict(type='DiNAT', ...),
        decode_head=dict(type='SETRUPHead', ...), auxiliary_head=..., train_cfg=..., test_cfg=dict(mode='whole'))`
        (configs/_base_/models/dinat.py:3-46 merged with configs/dinat/dinat_base.py:5-24), or a whole config holding it under
        'model'.  mmcv-only keys (init_cfg, norm_cfg, loss_decode, conv_cfg, act_cfg, in_patch_size, frozen_stages) are accepted
        and ignored where this build has one fixed choice."""
        cfg = dict(cfg.get("model", cfg))
        typ = cfg.pop("type", "EncoderDecoder")
        if typ != "EncoderDecoder":
            raise NotImplementedError(f"segmentor type {typ!r}: EncoderDecoder only")
        known = ("backbone", "decode_head", "auxiliary_head", "train_cfg", "test_cfg", "pretrained")
        extra = set(cfg) - set(known) - {"neck", "init_cfg"}
        if extra or cfg.get("neck") is not None:
            raise NotImplementedError(f"unsupported model keys: {sorted(extra | ({'neck'} if cfg.get('neck') is not None else set()))}")
        return cls(**{k: cfg[k] for k in known if k in cfg})

    def prepare_inference(self):
        """After the checkpoint is loaded: fold each head BatchNorm into its convolution (exact algebra in float32,
        W' = W * g/sqrt(v+eps), b' = beta - mu * g/sqrt(v+eps)) and put every convolution weight in channels_last, so
        the whole network runs on NHWC tensors with no layout copies.  Changes the state-dict layout: load first."""
        cms = [up[0] for up in self.decode_head.up_convs] if isinstance(self.decode_head, SETRUPHead) else \
              [m for m in self.decode_head.modules() if isinstance(m, _ConvModule)]
        for cm in cms:
            if isinstance(cm.bn, nn.BatchNorm2d):
                bn, conv = cm.bn, cm.conv
                scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach()
                conv.weight = nn.Parameter((conv.weight.detach() * scale.view(-1, 1, 1, 1)))
                conv.bias = nn.Parameter((bn.bias - bn.running_mean * scale).detach())
                cm.bn = nn.Identity()
        import os
        if not os.environ.get("PPNET_NO_FOLD"):          # A/B knob: keep the fused residual+LayerNorm form
            for level in self.backbone.levels:
                level.fold()
        self._narrow_levels(inference=True)              # the auxiliary head is a training-time branch
        self.prepared = True
        self.to(memory_format=torch.channels_last)
        return self

    prepared = False

    def encode_decode(self, img):
        out = self.decode_head(self.backbone(img))
        return F.interpolate(out, img.shape[-2:], mode="bilinear", align_corners=self.align_corners)   # (img may be u8 codes [B,R,R])

    def labels_u8(self, img):
        """argmax labels as u8 [B,R,R].  SETR-UP head with two classes on the GPU: the x2 up-sampling of the logits, the
        resize to the input size, softmax and argmax are one HIP kernel (ppn_seg_labels_2class); otherwise forward()."""
        head = self.decode_head
        if (img.is_cuda and isinstance(head, SETRUPHead) and head.conv_seg.out_channels == 2 and not self.align_corners
                and not head.align_corners and head.up_convs[-1][1].scale_factor == 2.0):
            return fused.seg_labels_2class(head(self.backbone(img), lowres=True), img.shape[-2:])
        if img.dtype == torch.uint8 and not self.backbone.patch_embed.takes_codes(img):
            img = fused.grid_to_image(img, IMG_MEAN, IMG_STD, next(self.parameters()).dtype)   # occupancy codes the tokenizer kernel cannot take: render
        return self.forward(img).to(torch.uint8)                  # (codes go straight to the palette tokenizer: NAT.forward)

    # ------------------------------------------------------------------ the reference's calling convention
    def forward(self, img=None, img_metas=None, return_loss=True, return_logits=False, **kwargs):
        """base.py:99-112.  Three forms:
        * `model(return_loss=False, img=[x], img_metas=[[meta, ...]])` — what single_gpu_test / multi_gpu_test call
          (mmseg/apis/test.py:93,196): one entry per test-time augmentation; returns list[np.ndarray int64 [H,W]], one per image.
        * `model(img=x, img_metas=[meta, ...], gt_semantic_seg=y)` (return_loss=True) — forward_train: the dict of losses.
        * `model(x)` with a tensor and no img_metas (this build's batched form): the argmax labels as a tensor [B,H,W]
          (+ the logits with return_logits=True)."""
        if isinstance(img, (list, tuple)):
            if return_loss:
                raise TypeError("return_loss=True takes img as a Tensor and img_metas as list[dict] (base.py:103-107)")
            return self.forward_test(list(img), img_metas, **kwargs)
        if img_metas is None and "gt_semantic_seg" not in kwargs:
            logits = self.encode_decode(img)
            pred = F.softmax(logits.float(), dim=1).argmax(dim=1)           # encoder_decoder.py:242,257
            return (pred, logits) if return_logits else pred
        if return_loss:
            return self.forward_train(img, img_metas, **kwargs)
        raise TypeError("return_loss=False takes img as list[Tensor] and img_metas as list[list[dict]] (base.py:103-107)")

    def forward_test(self, imgs, img_metas, rescale=True, **kwargs):
        """base.py:64-96 + encoder_decoder.py:254-292: per augmentation softmax of the logits resized to ori_shape, flipped back
        where the augmentation flipped, averaged over the augmentations; argmax -> one int64 array per image."""
        for var, name in ((imgs, "imgs"), (img_metas, "img_metas")):
            if not isinstance(var, list):
                raise TypeError(f"{name} must be a list, but got {type(var)}")
        if len(imgs) != len(img_metas):
            raise ValueError(f"num of augmentations ({len(imgs)}) != num of image meta ({len(img_metas)})")
        if len(imgs) == 1:
            return self.simple_test(imgs[0], img_metas[0], rescale)
        assert rescale                                                       # aug_test, encoder_decoder.py:274-276
        prob = self.inference(imgs[0], img_metas[0], rescale)
        for x, m in zip(imgs[1:], img_metas[1:]):
            prob = prob + self.inference(x, m, rescale)
        return list((prob / len(imgs)).argmax(dim=1).cpu().numpy())

    def simple_test(self, img, img_meta, rescale=True):
        """encoder_decoder.py:254-265.  Unflipped whole-image inference at the input size on the GPU takes the fused tail
        (labels_u8: up-sampling, resize, softmax and argmax in one kernel) — the same labels, as int64 arrays."""
        meta0 = _meta(img_meta)[0] if img_meta else {}
        size = tuple(meta0.get("ori_shape", img.shape[-2:])[:2]) if rescale else tuple(img.shape[-2:])
        if img.is_cuda and not meta0.get("flip", False) and size == tuple(img.shape[-2:]) and not torch.is_grad_enabled():
            return list(self.labels_u8(img).to(torch.int64).cpu().numpy())
        return list(self.inference(img, img_meta, rescale).argmax(dim=1).cpu().numpy())

    def inference(self, img, img_meta, rescale=True):
        """encoder_decoder.py:200-252: whole_inference + softmax + un-flip.  Returns the class probabilities [B,C,H,W]."""
        metas = _meta(img_meta) if img_meta else [{}]
        ori = metas[0].get("ori_shape")
        assert all(m.get("ori_shape") == ori for m in metas)
        logits = self.encode_decode(img)
        if rescale and ori is not None and tuple(ori[:2]) != tuple(logits.shape[-2:]):
            logits = F.interpolate(logits, tuple(ori[:2]), mode="bilinear", align_corners=self.align_corners)
        out = F.softmax(logits.float(), dim=1)
        if metas[0].get("flip", False):
            d = metas[0].get("flip_direction", "horizontal")
            assert d in ("horizontal", "vertical")
            out = out.flip(dims=(3,) if d == "horizontal" else (2,))
        return out

    def forward_train(self, img, img_metas, gt_semantic_seg, **kwargs):
        """encoder_decoder.py:122-152 with decode_head.py:209-237 (losses): {'decode.loss_ce', 'decode.acc_seg'} and, with an
        auxiliary head, {'aux.loss_ce', 'aux.acc_seg'} (loss weights 1.0 / 0.4, ignore_index 255)."""
        if self.prepared:
            raise RuntimeError("SegNet.prepare_inference() folded BatchNorm / LayerScale into the weights: build a fresh SegNet to train")
        feats = self.backbone(img)
        gt = gt_semantic_seg.squeeze(1).long() if gt_semantic_seg.dim() == 4 else gt_semantic_seg.long()
        losses = {}
        heads = [("decode", self.decode_head, 1.0)] + [(f"aux_{i}" if isinstance(self.auxiliary_head, nn.ModuleList) else "aux", h, h.loss_weight)
                                                       for i, h in enumerate(self._aux_heads())]
        for name, head, w in heads:
            logit = F.interpolate(head(feats).float(), gt.shape[-2:], mode="bilinear", align_corners=head.align_corners)
            losses[f"{name}.loss_ce"], losses[f"{name}.acc_seg"] = decode_losses(logit, gt, w)
        return losses


def randomize_neutral_parameters(model, seed=0, gamma=(0.05, 0.3)):
    """Give every parameter that a fresh initialisation leaves NEUTRAL a non-trivial seeded value, in place: LayerScale
    (`layer_scale=1e-5`, dinat_base.py:14, makes every residual branch invisible), biases (0), LayerNorm / BatchNorm affine
    parameters (1 / 0), the relative position bias and the BatchNorm running statistics (0 / 1).  No trained weights ship with
    the reference; a model initialised this way exercises the arithmetic the way a trained checkpoint does, which is what a
    precision comparison (bench.py's `ppnet.parity`, tests/test_ppnet_config3.py) needs.  Returns the model."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("gamma1") or name.endswith("gamma2"):
                p.copy_(torch.empty(p.shape).uniform_(gamma[0], gamma[1], generator=g))
            elif p.dim() == 1 and name.endswith("bias"):
                p.copy_(torch.empty(p.shape).uniform_(-0.2, 0.2, generator=g))
            elif p.dim() == 1 and name.endswith("weight"):
                p.copy_(torch.empty(p.shape).uniform_(0.7, 1.3, generator=g))
            elif name.endswith("rpb"):
                p.copy_(torch.empty(p.shape).normal_(0.0, 0.5, generator=g))
        for mod in model.modules():
            if isinstance(mod, nn.BatchNorm2d):
                mod.running_mean.copy_(torch.empty(mod.running_mean.shape).uniform_(-0.2, 0.2, generator=g))
                mod.running_var.copy_(torch.empty(mod.running_var.shape).uniform_(0.5, 1.5, generator=g))
    return model


@torch.no_grad()
def balance_classifier_bias(segnet, images):
    """Shift the decode head's class-1 bias so that the two classes split the pixels of `images` ([n,3,R,R], normalised)
    evenly: an UNTRAINED network labels every pixel alike, which makes any label-agreement figure trivially 1.  Run on the
    unprepared module (before SegNet.prepare_inference folds anything); returns the shift applied (the median class margin)."""
    lg = segnet.encode_decode(images).float()
    d = (lg[:, 1] - lg[:, 0]).flatten().median()
    segnet.decode_head.conv_seg.bias[1] -= d.to(segnet.decode_head.conv_seg.bias.dtype)
    return float(d)


def decode_losses(logit, gt, loss_weight=1.0, ignore_index=255):
    """(loss_ce, acc_seg) of BaseDecodeHead.losses (decode_head.py:231-265) for resized logits [B,C,H,W] and labels [B,H,W].
    mmseg's CrossEntropyLoss is F.cross_entropy(reduction='none', ignore_index) followed by a mean over ALL pixels — ignored
    ones contribute 0 to the sum and still count in the divisor (losses/cross_entropy_loss.py:20-31, losses/utils.py:66-68);
    accuracy() is called without an ignore index and divides by target.numel() (decode_head.py:264, losses/accuracy.py:39-49)."""
    loss = loss_weight * F.cross_entropy(logit, gt, ignore_index=ignore_index, reduction="none").mean()
    with torch.no_grad():
        acc = (logit.argmax(1) == gt).float().sum() * (100.0 / gt.numel())
    return loss, acc


def _meta(img_meta):
    """img_metas as the data loader hands them: list[dict], or a DataContainer-like object whose `.data[0]` is that list
    (mmseg/apis/test.py:97)."""
    if hasattr(img_meta, "data") and not isinstance(img_meta, (list, tuple)):
        img_meta = img_meta.data[0]
    return list(img_meta)


def normalize_images(rgb_u8):
    """u8 [B,R,R,3] or float [B,3,R,R] in [0,255] -> (x - mean) / std, [B,3,R,R] (planning_seg.py:12-41)."""
    x = rgb_u8.permute(0, 3, 1, 2).float() if rgb_u8.dtype == torch.uint8 else rgb_u8.float()
    mean = torch.tensor(IMG_MEAN, device=x.device).view(1, 3, 1, 1)
    std = torch.tensor(IMG_STD, device=x.device).view(1, 3, 1, 1)
    return (x - mean) / std
