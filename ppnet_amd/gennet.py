"""GenNet's AE-ViT (reference GenNet/networks/ae_vit.py:12-76 + vit.py:71-161) as an inference module with the
reference's constructor `AEViT(img_channels, out_channels, img_resolution, dim)` and state-dict key names
(`conv_first.{0,1}`, `enc_conv.i.{0,1}`, `vit_blocks.i.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,mlp.fc2}`,
`dec_conv.i.{0,1}`, `conv_final`), so `model.load_state_dict(torch.load(w)['model'])` (predict.py:51-52) works.

Structure: conv3x3+BN+LeakyReLU stem, int(log2(R//28)) stride-2 conv stages, 3 pre-LN ViT blocks (3 heads,
MLP x4, GELU, LN eps 1e-6) on the (R/2^n)^2 tokens, mirrored transposed-conv stages, conv3x3 to out_channels.
Dense work runs on the ROCm libraries through PyTorch (bf16 autocast optional); pinned against the reference's
own module by tests/golden/g13_aevit.npz.
"""
import math
from functools import partial

import os

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Attention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias=True):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        q, k, v = self.qkv(x).view(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        o = F.scaled_dot_product_attention(q, k, v, scale=self.scale)        # softmax(q k^T * scale) v, vit.py:103-109
        return self.proj(o.transpose(1, 2).reshape(B, N, C))


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


class _Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4, drop_path=0.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attention(dim, num_heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))
        self.drop_path_rate = float(drop_path)    # stochastic depth, training only (vit.py:150,158-161)

    def forward(self, x):
        from .segnet import drop_path
        x = x + drop_path(self.attn(_ln(x, self.norm1)), self.drop_path_rate, self.training)
        return x + drop_path(self.mlp(_ln(x, self.norm2)), self.drop_path_rate, self.training)


def _stage(conv):
    return nn.Sequential(conv, nn.BatchNorm2d(conv.out_channels), nn.LeakyReLU())


def pack_trunk_params(blocks):
    """The float32 parameter blocks ppn_gennet_trunk_bf16 reads (include/ppnet_hip.h), from the ViT blocks' modules."""
    out = []
    for b in blocks:
        f = lambda t: t.detach().float().reshape(-1)
        out += [f(b.norm1.weight), f(b.norm1.bias), f(b.attn.qkv.weight), f(b.attn.qkv.bias), f(b.attn.proj.weight), f(b.attn.proj.bias),
                f(b.norm2.weight), f(b.norm2.bias), f(b.mlp.fc1.weight), f(b.mlp.fc1.bias), f(b.mlp.fc2.weight.detach().float().t().contiguous()),
                f(b.mlp.fc2.bias)]
    return torch.cat(out).contiguous()


def pack_s2_weights(conv):
    """(weights, bias) in the layout ppn_gennet_conv_s2_bf16 reads, from a 24 -> 24 3x3 stride-2 Conv2d / ConvTranspose2d
    (padding 1, output_padding 1) whose BatchNorm is already folded."""
    C = 24
    w = conv.weight.detach().float()
    dev = w.device
    bias = torch.zeros(32, dtype=torch.float32, device=dev)
    if conv.bias is not None:
        bias[:C] = conv.bias.detach().float()
    if isinstance(conv, nn.ConvTranspose2d):                   # weight [ci][co][ky][kx]; out(2iy+a) <- in(iy+dy) through tap ky
        tap = lambda a, d: (1 if d == 0 else None) if a == 0 else (2 if d == 0 else 0)
        wt = torch.zeros(4, 32, 96, dtype=torch.float32, device=dev)
        for a in range(2):
            for b in range(2):
                for dy in range(2):
                    for dx in range(2):
                        ky, kx = tap(a, dy), tap(b, dx)
                        if ky is None or kx is None:
                            continue
                        p = dy * 2 + dx
                        wt[a * 2 + b, :C, p * C:(p + 1) * C] = w[:, :, ky, kx].t()
        return wt.to(torch.bfloat16).contiguous(), bias
    wk = torch.zeros(32, 224, dtype=torch.float32, device=dev)  # weight [co][ci][ky][kx] -> [co][(ky*3+kx)*24 + ci]
    wk[:C, :216] = w.permute(0, 2, 3, 1).reshape(C, 216)
    return wk.to(torch.bfloat16).contiguous(), bias


def pack_first_enc_weights(conv1, conv2):
    """The four parameter blocks of ppn_gennet_first_enc_bf16 (include/ppnet_hip.h) from the BatchNorm-folded first convolution
    (1 -> 24, 3x3, stride 1) and first encoder convolution (24 -> 24, 3x3, stride 2)."""
    C = 24
    dev = conv1.weight.device
    w = conv1.weight.detach().float().reshape(C, 9)
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    w1 = torch.zeros(32, 32, dtype=torch.bfloat16, device=dev)
    w1[:C, 0:9] = hi
    w1[:C, 16:25] = lo
    b1 = torch.zeros(32, dtype=torch.float32, device=dev)
    b1[:C] = conv1.bias.detach().float()
    bh = b1[:C].to(torch.bfloat16)                                         # the bias rides in the product: columns 9 (hi) and 25 (lo)
    w1[:C, 9] = bh                                                         # against the kernel's constant-1 input slots
    w1[:C, 25] = (b1[:C] - bh.float()).to(torch.bfloat16)
    w2 = conv2.weight.detach().float()                                     # [co][ci][ky][kx]
    slot_ci = [(4 * g + e) if e < 4 else ((16 + 4 * g + e - 4) if g < 2 else -1) for g in range(4) for e in range(8)]
    wk2 = torch.zeros(2, 9, 16, 32, dtype=torch.float32, device=dev)
    for k, ci in enumerate(slot_ci):
        if ci < 0:
            continue
        taps = w2[:, ci].reshape(C, 9)                                     # [co][tap]
        for nt in range(2):
            rows = min(16, C - nt * 16)
            wk2[nt, :, :rows, k] = taps[nt * 16:nt * 16 + rows].t()
    b2 = torch.zeros(32, dtype=torch.float32, device=dev)
    if conv2.bias is not None:
        b2[:C] = conv2.bias.detach().float()
    return w1.reshape(2, 16, 32).contiguous(), b1, wk2.to(torch.bfloat16).contiguous(), b2


def _is_s2_stage(c):
    return (c.in_channels == 24 and c.out_channels == 24 and c.kernel_size == (3, 3) and c.stride == (2, 2) and c.padding == (1, 1)
            and c.dilation == (1, 1) and c.groups == 1 and (not isinstance(c, nn.ConvTranspose2d) or c.output_padding == (1, 1)))


class _FusedStage(nn.Module):
    """A conv stage after prepare_inference(): the (transposed) convolution with the BatchNorm folded in runs WITHOUT its
    bias on the library, and bias + LeakyReLU are one in-place pass of the HIP kernel (ppn_bias_act_nhwc) instead of the
    library's separate add and activation kernels.  C % 8 != 0 or a CPU tensor: plain torch ops."""

    def __init__(self, conv, slope):
        super().__init__()
        self.conv, self.slope = conv, slope
        from .fused import WeightCache
        self._f32 = WeightCache()                         # (weight, bias) as float32 for the direct 1-channel kernel
        self._s2 = WeightCache()                          # (packed weight, bias) for the MFMA stride-2 kernels

    def s2_pack(self):
        return self._s2.get((self.conv.weight, self.conv.bias), lambda: pack_s2_weights(self.conv))

    def forward(self, x):
        c = self.conv
        if not (x.is_cuda and c.out_channels % 8 == 0):
            return F.leaky_relu(c(x), self.slope)
        if (isinstance(c, nn.Conv2d) and c.in_channels == 1 and c.out_channels <= 32 and c.kernel_size == (3, 3) and c.stride == (1, 1)
                and c.padding == (1, 1) and c.dilation == (1, 1)):
            # 1 -> dim at full resolution: the direct HIP kernel (ppn_conv3x3_c1_nhwc), bias + LeakyReLU inside
            from . import fused
            w32, b32 = self._f32.get((c.weight, c.bias), lambda: (c.weight.detach().float().contiguous(), c.bias.detach().float().contiguous()))
            return fused.conv3x3_c1(x, w32, b32, self.slope)
        import os
        if x.dtype == torch.bfloat16 and _is_s2_stage(c) and x.shape[-1] % 2 == 0 and x.shape[-2] % 2 == 0 and not os.environ.get("PPNET_LIBRARY_CONV"):
            # the 24-channel stride-2 stages: MFMA kernel with the weights in registers, bias + LeakyReLU in its epilogue
            from . import fused
            wp, bp = self.s2_pack()
            return fused.gennet_conv_s2(x, wp, bp, self.slope, isinstance(c, nn.ConvTranspose2d))
        if isinstance(c, nn.ConvTranspose2d):
            y = F.conv_transpose2d(x, c.weight, None, c.stride, c.padding, c.output_padding, c.groups, c.dilation)
        else:
            y = F.conv2d(x, c.weight, None, c.stride, c.padding, c.dilation, c.groups)
        from . import fused
        return fused.bias_act_(y.contiguous(memory_format=torch.channels_last), c.bias, self.slope)


def _ln(x, ln):
    # The CPU branches of this module (here and in _FusedStage) are not a product fallback: they let the CPU golden test
    # (tests/test_gennet_golden.py) check the module's wiring against the reference's own outputs without a GPU.  PPNet, the
    # drop-in classes and bench.py run on the GPU only, where every op below is a HIP kernel or a ROCm library call.
    if x.is_cuda:
        from . import fused
        return fused.layer_norm(x, ln)                                       # thread-per-row HIP kernel (C = 24)
    return ln(x)


class AEViT(nn.Module):
    def __init__(self, img_channels, out_channels, img_resolution=256, dim=192):
        super().__init__()
        n_down = int(math.log2(img_resolution // 28))                       # ae_vit.py:23
        self.conv_first = _stage(nn.Conv2d(img_channels, dim, 3, 1, 1))
        self.enc_conv = nn.ModuleList(_stage(nn.Conv2d(dim, dim, 3, 2, 1)) for _ in range(n_down))
        dpr = [float(v) for v in torch.linspace(0, 0.1, 3)]                   # ae_vit.py:35-36: drop_path_rate 0.1 over the depth
        self.vit_blocks = nn.Sequential(*[_Block(dim, 3, 4, dpr[i]) for i in range(3)])
        self.dec_conv = nn.ModuleList(_stage(nn.ConvTranspose2d(dim, dim, 3, 2, 1, output_padding=1)) for _ in range(n_down))
        self.conv_final = nn.Conv2d(dim, out_channels, 3, 1, 1)
        self._final_f32 = None                            # set by prepare_inference(): WeightCache of (float32 weight, float bias)
        from .fused import WeightCache
        self._trunk = WeightCache()                       # float32 parameter blocks of the fused ViT-trunk kernel
        self._first_enc = WeightCache()                   # packed parameters of the fused first convolution + first encoder stage

    def prepare_inference(self):
        """After the checkpoint is loaded: fold every eval-mode BatchNorm into the (transposed) convolution in front of
        it (exact algebra in float32) and switch the convolution weights to channels_last. Load the state dict first."""
        for stage in [self.conv_first, *self.enc_conv, *self.dec_conv]:
            conv, bn = stage[0], stage[1]
            if not isinstance(bn, nn.BatchNorm2d):
                continue
            scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach()
            shape = (1, -1, 1, 1) if isinstance(conv, nn.ConvTranspose2d) else (-1, 1, 1, 1)     # out-channel axis
            conv.weight = nn.Parameter(conv.weight.detach() * scale.view(shape))
            conv.bias = nn.Parameter(((conv.bias.detach() if conv.bias is not None else 0) - bn.running_mean) * scale + bn.bias.detach())
            stage[1] = nn.Identity()
        self.to(memory_format=torch.channels_last)
        self.conv_first = _FusedStage(self.conv_first[0], self.conv_first[2].negative_slope)
        self.enc_conv = nn.ModuleList(_FusedStage(st[0], st[2].negative_slope) for st in self.enc_conv)
        self.dec_conv = nn.ModuleList(_FusedStage(st[0], st[2].negative_slope) for st in self.dec_conv)
        from .fused import WeightCache
        self._final_f32 = WeightCache()
        return self

    def _final_pack(self):
        cf = self.conv_final
        return self._final_f32.get((cf.weight, cf.bias), lambda: (
            cf.weight.detach().float().contiguous(), float(cf.bias.detach().float()[0]) if cf.bias is not None and cf.out_channels == 1 else 0.0))

    def _fused_first_stage(self, x):
        """Prepared bfloat16 inference: conv_first and enc_conv[0] as one kernel (ppn_gennet_first_enc_bf16), or None."""
        import os
        cf = self.conv_first
        if not (isinstance(cf, _FusedStage) and len(self.enc_conv) > 0 and isinstance(self.enc_conv[0], _FusedStage) and x.is_cuda
                and x.dtype == torch.bfloat16 and not os.environ.get("PPNET_GENNET_UNFUSED")):
            return None
        c1, c2 = cf.conv, self.enc_conv[0].conv
        if not (isinstance(c1, nn.Conv2d) and c1.in_channels == 1 and c1.out_channels == 24 and c1.kernel_size == (3, 3) and c1.stride == (1, 1)
                and c1.padding == (1, 1) and c1.bias is not None and isinstance(c2, nn.Conv2d) and _is_s2_stage(c2)
                and x.shape[-1] % 2 == 0 and x.shape[-2] % 2 == 0):
            return None
        from . import fused
        packs = self._first_enc.get((c1.weight, c1.bias, c2.weight, c2.bias), lambda: pack_first_enc_weights(c1, c2))
        return fused.gennet_first_enc(x, *packs, cf.slope, self.enc_conv[0].slope)

    def forward(self, x):
        y = self._fused_first_stage(x)
        if y is not None:
            x = y
            for blk in self.enc_conv[1:]:
                x = blk(x)
        else:
            x = self.conv_first(x)
            for blk in self.enc_conv:
                x = blk(x)
        B, C, H, W = x.shape
        import os
        if (x.is_cuda and x.dtype == torch.bfloat16 and self._final_f32 is not None and C == 24 and H * W <= 1024 and (H * W) % 8 == 0
                and self.vit_blocks[0].attn.num_heads == 3 and not os.environ.get("PPNET_LIBRARY_TRUNK")):
            # prepared bfloat16 inference: the three ViT blocks are one kernel (residual stream in registers, K / V in LDS)
            from . import fused
            trunk = self._trunk.get(list(self.vit_blocks.parameters()), lambda: pack_trunk_params(self.vit_blocks).to(x.device))
            x = fused.gennet_trunk(x.contiguous(memory_format=torch.channels_last), trunk, len(self.vit_blocks))
        else:
            t = self.vit_blocks(x.flatten(2).transpose(1, 2))                # feature2token / token2feature, base.py:43-52
            x = t.transpose(1, 2).reshape(B, C, H, W)
        c = self.conv_final
        last = self.dec_conv[-1] if len(self.dec_conv) else None
        if (isinstance(last, _FusedStage) and x.is_cuda and x.dtype == torch.bfloat16 and self._final_f32 is not None and c.out_channels == 1
                and c.in_channels == 24 and isinstance(last.conv, nn.ConvTranspose2d) and _is_s2_stage(last.conv)
                and not os.environ.get("PPNET_GENNET_UNFUSED_TAIL") and not os.environ.get("PPNET_LIBRARY_CONV")):
            # prepared bfloat16 inference: the last decoder stage and the final convolution are one kernel — the 24-channel tensor
            # at the output resolution never reaches memory (ppn_gennet_dec_final_bf16; bit-identical to the two kernels)
            from . import fused
            for blk in self.dec_conv[:-1]:
                x = blk(x)
            wp, bp = last.s2_pack()
            wf, bf = self._final_pack()
            return fused.gennet_dec_final(x, wp, bp, last.slope, wf, bf)
        for blk in self.dec_conv:
            x = blk(x)
        if x.is_cuda and self._final_f32 is not None and c.out_channels == 1 and c.in_channels % 8 == 0 and c.in_channels <= 32:
            from . import fused                                               # dim -> 1 at full resolution: direct HIP kernel
            wf, bf = self._final_pack()
            return fused.conv3x3_to1(x, wf, bf)
        return c(x)


AE = AEViT      # predict.py:12 `from networks import AEViT as AE`


def normalize_heatmap_u8(y):
    """predict.py:95-102 per sample: (y - min) / (max - min) -> 8-bit 'L' image. y [B,1,R,R] -> u8 [B,R,R]."""
    B = y.shape[0]
    f = y.float().reshape(B, -1)
    lo = f.min(dim=1, keepdim=True).values
    hi = f.max(dim=1, keepdim=True).values
    n = (f - lo) / (hi - lo)
    return (n * 255).to(torch.uint8).reshape(B, y.shape[-2], y.shape[-1])    # ToPILImage: mul(255).byte()


WEIGHTS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "weights")


def trained_checkpoint(resolution):
    """Path of the GenNet checkpoint this build trained itself at `resolution` (tools/train_gennet.py: the build's own training step
    on pairs from the build's own generator; the reference ships no weights), or None.  The file has the reference's layout —
    {'model': state_dict}, float32 (GenNet/train.py:133-141; predict.py:51-52 reads that key)."""
    path = os.path.join(WEIGHTS_DIR, f"gennet_r{int(resolution)}.pth")
    return path if os.path.exists(path) else None


def load_trained(model, resolution):
    """Load trained_checkpoint(resolution) into an (unprepared) AEViT the way predict.py:51-52 does; returns True if there was one."""
    path = trained_checkpoint(resolution)
    if path is None:
        return False
    model.load_state_dict(torch.load(path, map_location="cpu", weights_only=True)["model"], strict=True)
    return True
