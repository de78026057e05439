"""The hand-written bf16 MFMA kernels (ppnet_amd/csrc/mfma_gemm.h) through the C ABI against float32 PyTorch references of the
same op on the same bf16 inputs: implicit-GEMM 3x3 convolution (stride 1 / 2, bias, ReLU), the fused conv + ReLU + 2-class
classifier, and the dense projection with its three epilogues — ragged shapes (one tile per workgroup) and aligned shapes with
more tiles than CUs (the persistent form, where the LDS-DMA stream and the counted waits run across tile boundaries)."""
import pytest

torch = pytest.importorskip("torch")
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _close(got, want, rel=1.5e-2):
    got, want = got.float(), want.float()
    scale = max(1.0, float(want.abs().max()))
    err = float((got - want).abs().max())
    assert err <= rel * scale, (err, scale)


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,relu", [
    (2, 16, 16, 64, 256, 1, True),        # one ragged-free small case
    (3, 9, 7, 128, 264, 1, False),        # ragged M and N, odd image
    (5, 8, 8, 192, 512, 1, True),
    (2, 32, 32, 128, 256, 2, False),      # NAT ConvDownsampler: stride 2, no bias
    (3, 17, 13, 64, 512, 2, True),        # stride 2 on odd sizes
    (70, 32, 32, 64, 512, 1, True),       # 560 tiles > CUs
])
def test_conv3x3_mfma_vs_torch(B, H, W, Cin, Cout, stride, relu):
    from ppnet_amd import fused
    torch.manual_seed(B * 100 + H)
    x = torch.randn(B, Cin, H, W, device="cuda").to(BF).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, 3, 3, device="cuda") * (2.0 / (9 * Cin)) ** 0.5).to(BF)
    b = torch.randn(Cout, device="cuda")
    want = F.conv2d(x.float(), w.float(), b, stride, 1)
    if relu:
        want = F.relu(want)
    got = fused.conv3x3_mfma(x, w.permute(0, 2, 3, 1).contiguous(), b.contiguous(), stride=stride, relu=relu)
    assert got.shape == want.shape and got.dtype == BF
    _close(got, want)


def test_conv3x3_relu_classify2_vs_torch():
    from ppnet_amd import fused
    torch.manual_seed(3)
    for (B, H, W, Cin, Cout) in [(3, 8, 8, 128, 512), (40, 16, 16, 64, 512), (2, 10, 6, 64, 264),
                                 (5, 12, 12, 64, 256), (2, 9, 7, 128, 64)]:       # Cout <= 256: one column block, no workspace, straight onto the logits
        x = torch.randn(B, Cin, H, W, device="cuda").to(BF).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(Cout, Cin, 3, 3, device="cuda") * (2.0 / (9 * Cin)) ** 0.5).to(BF)
        b = torch.randn(Cout, device="cuda") * 0.3
        w2 = torch.randn(2, Cout, device="cuda") * 0.1
        b2 = torch.randn(2, device="cuda")
        want = F.conv2d(F.relu(F.conv2d(x.float(), w.float(), b, 1, 1)), w2.view(2, Cout, 1, 1), b2)
        got = fused.conv3x3_relu_classify2(x, w.permute(0, 2, 3, 1).contiguous(), b.contiguous(), w2.contiguous(), b2)
        assert got.shape == (B, 2, H, W) and got.dtype == torch.float32
        _close(got, want, rel=2e-3)                                           # float32 out: only the bf16 operands round
        for _ in range(3):                                                    # per-slot partial sums added in a fixed order: no atomics
            again = fused.conv3x3_relu_classify2(x, w.permute(0, 2, 3, 1).contiguous(), b.contiguous(), w2.contiguous(), b2)
            assert torch.equal(again, got)


@pytest.mark.parametrize("M,N,K,persistent", [(300, 264, 128, 0), (512, 256, 192, 0), (256 * 40, 768, 192, 256), (256 * 33, 512, 320, 256),
                                              (65536, 512, 512, 0), (65536, 512, 512, 256),
                                              # few rows (batches of 1-16): the wave-per-block kernel of gemm_small.hip
                                              (49, 1024, 4096, 0), (196, 1536, 512, 0), (784, 768, 256, 0), (3136, 512, 2048, 0), (33, 64, 128, 0)])
def test_gemm_bf16_epilogues_vs_torch(M, N, K, persistent):
    from ppnet_amd import fused
    torch.manual_seed(M % 97 + N)
    a = torch.randn(M, K, device="cuda").to(BF)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    b = torch.randn(N, device="cuda")
    lin = a.float() @ w.float().t()
    _close(fused.gemm_bf16(a, w, b, "bias", persistent_blocks=persistent), lin + b)
    _close(fused.gemm_bf16(a, w, b, "bias_gelu", persistent_blocks=persistent), F.gelu(lin + b))
    _close(fused.gemm_bf16(a, w, b, "bias_relu", persistent_blocks=persistent), F.relu(lin + b))
    c0 = torch.randn(M, N, device="cuda").to(BF)
    c = c0.clone()
    out = fused.gemm_bf16(a, w, None, "accum", out=c, persistent_blocks=persistent)
    assert out is c
    _close(c, c0.float() + lin)


@pytest.mark.parametrize("B,H,W", [(2, 8, 8), (3, 32, 32), (5, 64, 48), (4, 256, 256)])
def test_gennet_stride2_stages_vs_torch(B, H, W):
    """ppn_gennet_conv_s2_bf16 (encoder Conv2d(24,24,3,2,1) and decoder ConvTranspose2d(24,24,3,2,1,output_padding=1), bias +
    LeakyReLU fused) against float32 torch on the same bf16 operands."""
    import torch.nn as nn
    from ppnet_amd import fused
    from ppnet_amd.gennet import pack_s2_weights
    torch.manual_seed(B + H)
    for transposed in (False, True):
        conv = (nn.ConvTranspose2d(24, 24, 3, 2, 1, output_padding=1) if transposed else nn.Conv2d(24, 24, 3, 2, 1)).cuda()
        with torch.no_grad():
            conv.weight.copy_(conv.weight.to(BF).float())
        x = torch.randn(B, 24, H, W, device="cuda").to(BF).contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            want = F.leaky_relu(conv(x.float()), 0.01)
        wp, bp = pack_s2_weights(conv)
        got = fused.gennet_conv_s2(x, wp, bp, 0.01, transposed)
        assert got.shape == want.shape and got.dtype == BF
        _close(got, want)


@pytest.mark.parametrize("B,side", [(3, 32), (2, 28), (5, 8)])
def test_gennet_trunk_vs_torch_blocks(B, side):
    """ppn_gennet_trunk_bf16 (three pre-LN ViT blocks in one kernel, vit.py:88-161) against the module's own float32 forward on
    the same (bf16-representable) weights and input.  Inside the kernel q, k, v and the probabilities are bfloat16 (float32
    accumulation), the residual stream float32: tolerance 2 % of the output scale."""
    from ppnet_amd import fused
    from ppnet_amd.gennet import _Block, pack_trunk_params
    torch.manual_seed(side)
    blocks = torch.nn.Sequential(*[_Block(24, 3, 4) for _ in range(3)]).cuda().eval()
    with torch.no_grad():
        for p in blocks.parameters():
            if p.dim() == 1:
                p.uniform_(-0.5, 0.5)
            p.copy_(p.to(BF).float())
        for b in blocks:
            b.norm1.weight.uniform_(0.7, 1.3); b.norm2.weight.uniform_(0.7, 1.3)
    x = torch.randn(B, 24, side, side, device="cuda").to(BF).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        t = blocks.cpu()(x.float().cpu().flatten(2).transpose(1, 2))          # CPU float32: plain torch ops, no HIP LayerNorm kernel
        want = t.transpose(1, 2).reshape(B, 24, side, side)
    got = fused.gennet_trunk(x, pack_trunk_params(blocks).cuda(), 3)
    assert got.shape == want.shape and got.dtype == BF
    _close(got.cpu(), want, rel=2e-2)


@pytest.mark.parametrize("gain", [1.0, 4.0, 8.0])
def test_gennet_trunk_softmax_stabiliser_paths(gain, monkeypatch):
    """The trunk's softmax subtracts either the bound |q| max|k| (no pass over the keys; taken while the bound is <= 40) or the exact
    row maximum (two passes).  gain 1: ordinary logits, the bound path by default — it must agree with the forced exact path
    (PPNET_TRUNK_EXACT_MAX=1) to the bfloat16 rounding of the output.  gain 4: q and k weights x4, logits x16, the bound exceeds
    40 for some (gain 4) or nearly all (gain 8) query tiles and those take the exact path by themselves — against float32 torch
    (a peaked softmax amplifies the bfloat16 rounding of q and k: 5 % of the output scale) and again close to the forced path."""
    from ppnet_amd import fused
    from ppnet_amd.gennet import _Block, pack_trunk_params
    torch.manual_seed(17)
    blocks = torch.nn.Sequential(*[_Block(24, 3, 4) for _ in range(3)]).cuda().eval()
    with torch.no_grad():
        for b in blocks:
            b.attn.qkv.weight[:48] *= gain
            b.attn.qkv.bias[:48] *= gain
        for p in blocks.parameters():
            p.copy_(p.to(BF).float())
    x = torch.randn(3, 24, 32, 32, device="cuda").to(BF).contiguous(memory_format=torch.channels_last)
    params = pack_trunk_params(blocks).cuda()
    got = fused.gennet_trunk(x, params, 3)
    monkeypatch.setenv("PPNET_TRUNK_EXACT_MAX", "1")
    exact = fused.gennet_trunk(x, params, 3)
    monkeypatch.delenv("PPNET_TRUNK_EXACT_MAX")
    d = (got.float() - exact.float()).abs()
    assert float(d.max()) <= 2.0 ** -6 * float(exact.float().abs().max()) and float(d.mean()) < 1e-3 * float(exact.float().abs().mean())
    with torch.no_grad():
        t = blocks.cpu()(x.float().cpu().flatten(2).transpose(1, 2))
        want = t.transpose(1, 2).reshape(3, 24, 32, 32)
    _close(got.cpu(), want, rel={1.0: 2e-2, 4.0: 5e-2, 8.0: 0.12}[gain])     # logits x64 at gain 8: q / k rounding times a one-hot softmax


def _nat128_modules(seed):
    import torch.nn as nn
    g = torch.Generator().manual_seed(seed)
    ln = nn.LayerNorm(128, eps=1e-5)
    qkv, fc1, fc2 = nn.Linear(128, 384), nn.Linear(128, 256), nn.Linear(256, 128)
    with torch.no_grad():
        ln.weight.copy_(1.0 + 0.2 * torch.randn(128, generator=g)); ln.bias.copy_(0.1 * torch.randn(128, generator=g))
        for lin in (qkv, fc1, fc2):
            lin.weight.copy_(torch.randn(lin.weight.shape, generator=g) / lin.in_features ** 0.5)
            lin.bias.copy_(0.3 * torch.randn(lin.bias.shape, generator=g))
    return [m.cuda().to(torch.bfloat16) for m in (ln, qkv, fc1, fc2)]


@pytest.mark.parametrize("tokens", [16, 4096 + 48, 1 << 20])
def test_nat128_streaming_kernels_vs_float64(tokens):
    """ppn_nat128_ln_qkv_bf16 / ppn_nat128_ln_mlp_bf16 (SegNet/nat.py:101-153 at C = 128) against the same algebra in
    float64 on the SAME bfloat16 parameters and input: LN(s + offset) -> qkv, and s + fc2.W gelu(fc1(LN(s + offset))).
    Tolerance: the kernel rounds the LayerNorm output and the hidden activations to bfloat16 once each (as the library path
    does when it stores them), accumulates in float32 and rounds the result to bfloat16: errors are a few bf16 ulps of the
    O(1..4) outputs — 0.06 absolute bounds the worst element, 6e-3 the mean."""
    from ppnet_amd import fused
    ln, qkv, fc1, fc2 = _nat128_modules(5)
    g = torch.Generator().manual_seed(tokens)
    s = (1.5 * torch.randn(tokens, 128, generator=g)).cuda().to(torch.bfloat16)
    off = (0.5 * torch.randn(128, generator=g)).cuda()
    d = lambda t: t.detach().double()
    y = torch.nn.functional.layer_norm(d(s) + d(off), (128,), d(ln.weight), d(ln.bias), ln.eps)
    want_qkv = y @ d(qkv.weight).t() + d(qkv.bias)
    want_s = d(s) + torch.nn.functional.gelu(y @ d(fc1.weight).t() + d(fc1.bias)) @ d(fc2.weight).t()
    got_qkv = fused.nat128_ln_qkv(s, off, ln, qkv)
    s2 = s.clone()
    fused.nat128_ln_mlp_(s2, off, ln, fc1, fc2)
    torch.cuda.synchronize()
    for got, want, name in ((got_qkv, want_qkv, "qkv"), (s2, want_s, "mlp")):
        err = (got.double() - want).abs()
        assert float(err.max()) < 0.06 and float(err.mean()) < 6e-3, (name, float(err.max()), float(err.mean()))
    # the last layer of a level: a per-channel constant added in the epilogue (ppn_nat128_ln_mlp_add_bf16)
    add = (0.7 * torch.randn(128, generator=g)).cuda()
    s3 = s.clone()
    fused.nat128_ln_mlp_(s3, off, ln, fc1, fc2, final_add=add)
    err = (s3.double() - (want_s + d(add))).abs()
    assert float(err.max()) < 0.06 and float(err.mean()) < 6e-3, ("mlp + add", float(err.max()), float(err.mean()))
    # no offset / no bias forms
    got0 = fused.nat128_ln_qkv(s, None, ln, qkv)
    y0 = torch.nn.functional.layer_norm(d(s), (128,), d(ln.weight), d(ln.bias), ln.eps)
    assert float((got0.double() - (y0 @ d(qkv.weight).t() + d(qkv.bias))).abs().max()) < 0.06


def test_nat128_layer_matches_library_form(monkeypatch):
    """A folded 128-channel NAT layer through the streaming kernels against the same layer on the library GEMM path
    (PPNET_LIBRARY_NAT128): both are bf16 pipelines with float32 accumulation, they differ by rounding only."""
    from ppnet_amd.segnet import NATBlock
    torch.manual_seed(3)
    blk = NATBlock(128, 2, 4, 7, dilations=[1, 2], downsample=False, layer_scale=1e-1).cuda().to(torch.bfloat16).eval()
    with torch.no_grad():
        for p in blk.parameters():
            if p.dim() == 1 and p.numel() == 128:
                p.add_(0.05 * torch.randn_like(p))
    blk.fold()
    x = torch.randn(2, 16, 16, 128, device="cuda").to(torch.bfloat16)
    with torch.no_grad():
        a, _ = blk(x)
        monkeypatch.setenv("PPNET_LIBRARY_NAT128", "1")
        b, _ = blk(x)
    err = (a.float() - b.float()).abs()
    assert float(err.max()) < 0.05 and float(err.mean()) < 4e-3, (float(err.max()), float(err.mean()))


def _code_grids(B, R, seed):
    g = torch.Generator().manual_seed(seed)
    grid = torch.full((B, R, R), 255, dtype=torch.uint8)
    grid[torch.rand(B, R, R, generator=g) < 0.3] = 0                      # obstacles
    grid[torch.rand(B, R, R, generator=g) < 0.02] = 128                   # marker pixels
    grid[torch.rand(B, R, R, generator=g) < 0.01] = 77                    # any other code renders black
    grid[:, 0, :] = 0; grid[:, :, -1] = 128                               # borders exercise the zero padding
    return grid.cuda()


@pytest.mark.parametrize("B,R", [(1, 32), (3, 64), (2, 256)])
def test_tokenizer_palette_conv_vs_float64(B, R):
    """ppn_tokenizer_conv1_codes_bf16 (SegNet/nat.py:24-40 first convolution, from the occupancy codes) against
    torch.nn.functional.conv2d in float64 on the rendered bfloat16 palette image (ppn_grid_to_image) with the same
    bfloat16 weights.  The table is hi + lo bfloat16 (16 mantissa bits per entry) and the sum is float32: the result
    differs from the float64 one by the final bfloat16 rounding only — half an ulp, 2^-9 relative, plus 1e-4."""
    from ppnet_amd import fused
    from ppnet_amd.segnet import IMG_MEAN, IMG_STD
    torch.manual_seed(R)
    conv = torch.nn.Conv2d(3, 64, 3, 2, 1).cuda().to(torch.bfloat16)
    grid = _code_grids(B, R, R + B)
    img = fused.grid_to_image(grid, IMG_MEAN, IMG_STD, torch.bfloat16)
    # float64 reference on the host (a float64 convolution on the GPU can send MIOpen into a long search)
    want = torch.nn.functional.conv2d(img.double().cpu(), conv.weight.double().cpu(), conv.bias.double().cpu(), 2, 1).permute(0, 2, 3, 1)
    got = fused.tokenizer_conv1_codes(grid, fused.tokenizer_lut(conv, IMG_MEAN, IMG_STD)).double().cpu()
    assert got.shape == want.shape
    assert bool(((got - want).abs() <= want.abs() * 2.0 ** -8 + 1e-4).all()), float((got - want).abs().max())


def test_tokenizer_codes_path_matches_image_path():
    """ConvTokenizer.forward_codes (palette convolution, bias-free second convolution, bias inside the LayerNorm) against
    ConvTokenizer.forward on the rendered image: same tokens up to bfloat16 rounding of the intermediates."""
    from ppnet_amd import fused
    from ppnet_amd.segnet import ConvTokenizer, IMG_MEAN, IMG_STD
    torch.manual_seed(9)
    tok = ConvTokenizer(3, 128, torch.nn.LayerNorm).cuda().to(torch.bfloat16).eval()
    grid = _code_grids(2, 64, 4)
    assert tok.takes_codes(grid)
    with torch.no_grad():
        a = tok.forward_codes(grid).float()
        b = tok(fused.grid_to_image(grid, IMG_MEAN, IMG_STD, torch.bfloat16)).float()
    err = (a - b).abs()
    assert a.shape == b.shape and float(err.max()) < 0.08 and float(err.mean()) < 6e-3, (float(err.max()), float(err.mean()))


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_heatmap_u8_kernel_is_bit_identical_to_the_torch_composition(dtype):
    """ppn_heatmap_u8 (predict.py:95-102 min-max + ToPILImage's mul(255).byte()) against gennet.normalize_heatmap_u8, the torch
    composition of the same float32 operations in the same order: identical bytes."""
    from ppnet_amd import fused
    from ppnet_amd.gennet import normalize_heatmap_u8
    g = torch.Generator().manual_seed(8)
    y = (torch.randn(5, 1, 96, 160, generator=g) * 3 + 0.7).cuda().to(getattr(torch, dtype))
    got = fused.heatmap_u8(y)
    want = normalize_heatmap_u8(y)
    assert got.shape == want.shape == (5, 96, 160) and got.dtype == torch.uint8
    assert torch.equal(got, want)
    assert int(got.reshape(5, -1).min(dim=1).values.max()) == 0 and int(got.reshape(5, -1).max(dim=1).values.min()) == 255


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (1, 34, 50), (3, 256, 256)])
def test_gennet_fused_first_stage_vs_float64(B, H, W):
    """ppn_gennet_first_enc_bf16 (ae_vit.py:24-36: first convolution + first stride-2 stage in one kernel) against the two
    convolutions in float64 on the same bfloat16 parameters, with the intermediate rounded to bfloat16 where the two-kernel path
    stores it.  Inputs are the {0,1} mask GenNet sees plus a general bfloat16 image.  Tolerance: the float32 accumulations and
    the final bfloat16 rounding — 2^-8 relative + 2e-3."""
    import torch.nn as nn
    import torch.nn.functional as F
    from ppnet_amd import fused
    from ppnet_amd.gennet import pack_first_enc_weights
    torch.manual_seed(H + W)
    c1, c2 = nn.Conv2d(1, 24, 3, 1, 1).cuda().to(torch.bfloat16), nn.Conv2d(24, 24, 3, 2, 1).cuda().to(torch.bfloat16)
    packed = pack_first_enc_weights(c1, c2)
    g = torch.Generator().manual_seed(B)
    for x in ((torch.rand(B, 1, H, W, generator=g) < 0.4).float(), torch.randn(B, 1, H, W, generator=g)):
        x = x.cuda().to(torch.bfloat16)
        got = fused.gennet_first_enc(x, *packed, 0.01, 0.2).double().cpu()
        h = lambda t: t.detach().double().cpu()                            # float64 reference on the host
        y1 = F.leaky_relu(F.conv2d(h(x), h(c1.weight), h(c1.bias), 1, 1), 0.01).to(torch.bfloat16).double()
        want = F.leaky_relu(F.conv2d(y1, h(c2.weight), h(c2.bias), 2, 1), 0.2)
        assert got.shape == want.shape == (B, 24, H // 2, W // 2)
        err = (got - want).abs()
        # a float32-vs-float64 difference can flip the intermediate's bfloat16 rounding (one ulp of one of 216 terms): allow for it
        assert bool((err <= want.abs() * 2.0 ** -7 + 6e-3).all()), float(err.max())
        assert float(err.mean()) < 1e-3


def test_gennet_fused_first_stage_matches_two_kernel_path(monkeypatch):
    """AEViT (prepared, bfloat16) with the fused first stage against the same module with PPNET_GENNET_UNFUSED=1."""
    from ppnet_amd.gennet import AEViT
    torch.manual_seed(11)
    net = AEViT(1, 1, img_resolution=64, dim=24).cuda().eval()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5); m.weight.uniform_(0.8, 1.2); m.bias.normal_(0, 0.1)
        net.prepare_inference().to(torch.bfloat16)
        x = (torch.rand(4, 1, 64, 64, device="cuda") < 0.5).to(torch.bfloat16)
        a = net(x).float()
        monkeypatch.setenv("PPNET_GENNET_UNFUSED", "1")
        b = net(x).float()
    err = (a - b).abs()
    assert float(err.max()) < 0.05 * float(b.abs().max()) + 1e-3 and float(err.mean()) < 5e-3 * float(b.abs().mean()) + 1e-4


@pytest.mark.parametrize("B,R", [(1, 64), (3, 128), (2, 256)])
def test_tokenizer_one_kernel_vs_float64(B, R):
    """ppn_tokenizer_codes_bf16 (SegNet/nat.py:17-46: both convolutions + LayerNorm from the occupancy codes) against float64:
    conv2d on the rendered bfloat16 palette image, the intermediate rounded to bfloat16 (where the two-kernel path stores it),
    second conv2d, layer_norm — all with the module's bfloat16 parameters.  The output is O(1) after the LayerNorm: 0.03 absolute
    covers the final bfloat16 rounding plus the rare one-ulp flips of the intermediate; the mean error is an order below."""
    import torch.nn.functional as F
    from ppnet_amd import fused
    from ppnet_amd.segnet import ConvTokenizer, IMG_MEAN, IMG_STD
    torch.manual_seed(R)
    tok = ConvTokenizer(3, 128, torch.nn.LayerNorm).cuda().to(torch.bfloat16).eval()
    with torch.no_grad():
        tok.norm.weight.uniform_(0.5, 1.5); tok.norm.bias.normal_(0, 0.2)
    grid = _code_grids(B, R, R + B)
    d = lambda t: t.detach().double().cpu()                                 # the float64 reference runs on the host: a float64 convolution
    with torch.no_grad():                                                   # on the GPU sends MIOpen into a minutes-long search
        got = tok.forward_codes(grid).double().cpu()
        img = fused.grid_to_image(grid, IMG_MEAN, IMG_STD, torch.bfloat16).double().cpu()
        y1 = F.conv2d(img, d(tok.proj[0].weight), d(tok.proj[0].bias), 2, 1).to(torch.bfloat16).double()
        y2 = F.conv2d(y1, d(tok.proj[1].weight), d(tok.proj[1].bias), 2, 1).permute(0, 2, 3, 1)
        want = F.layer_norm(y2, (128,), d(tok.norm.weight), d(tok.norm.bias), tok.norm.eps)
    assert got.shape == want.shape == (B, R // 4, R // 4, 128)
    err = (got - want).abs()
    assert float(err.max()) < 0.03 and float(err.mean()) < 3e-3, (float(err.max()), float(err.mean()))


def test_tokenizer_one_kernel_matches_two_kernel_path(monkeypatch):
    from ppnet_amd.segnet import ConvTokenizer
    torch.manual_seed(4)
    tok = ConvTokenizer(3, 128, torch.nn.LayerNorm).cuda().to(torch.bfloat16).eval()
    grid = _code_grids(2, 128, 9)
    with torch.no_grad():
        a = tok.forward_codes(grid).float()
        monkeypatch.setenv("PPNET_TOKENIZER_TWO_KERNELS", "1")
        b = tok.forward_codes(grid).float()
    err = (a - b).abs()
    assert a.shape == b.shape and float(err.max()) < 0.05 and float(err.mean()) < 4e-3, (float(err.max()), float(err.mean()))


@pytest.mark.parametrize("B,H,W", [(2, 8, 8), (3, 9, 21), (2, 128, 128), (1, 40, 24)])
def test_gennet_fused_decoder_tail_is_bit_identical_to_the_two_kernels(B, H, W):
    """ppn_gennet_dec_final_bf16 (last ConvTranspose2d stage + bias + LeakyReLU, then the 24 -> 1 convolution, the 24-channel
    tensor never stored) against ppn_gennet_conv_s2_bf16 followed by ppn_conv3x3_to1_nhwc: the same products in the same order,
    so every bfloat16 output is EQUAL (ragged tiles, image borders and odd sizes included); and against float32 torch."""
    import torch.nn as nn
    from ppnet_amd import fused
    from ppnet_amd.gennet import pack_s2_weights
    torch.manual_seed(3 * B + H)
    dec = nn.ConvTranspose2d(24, 24, 3, 2, 1, output_padding=1).cuda()
    fin = nn.Conv2d(24, 1, 3, 1, 1).cuda()
    with torch.no_grad():
        dec.weight.copy_(dec.weight.to(BF).float())
    x = torch.randn(B, 24, H, W, device="cuda").to(BF).contiguous(memory_format=torch.channels_last)
    wp, bp = pack_s2_weights(dec)
    w1 = fin.weight.detach().float().contiguous()
    b1 = float(fin.bias.detach()[0])
    mid = fused.gennet_conv_s2(x, wp, bp, 0.01, True)
    two = fused.conv3x3_to1(mid, w1, b1)
    one = fused.gennet_dec_final(x, wp, bp, 0.01, w1, b1)
    assert one.shape == two.shape == (B, 1, 2 * H, 2 * W) and one.dtype == BF
    assert torch.equal(one, two)
    with torch.no_grad():
        want = fin(F.leaky_relu(dec(x.float()), 0.01).to(BF).float())
    _close(one, want)


def test_nat128_proj_add_vs_float64():
    """ppn_nat128_proj_add_bf16: s += a Wp^T for 128-channel tokens (SegNet/nat.py:144-146 with LayerScale folded), against the
    float64 product of the same bfloat16 operands: one bfloat16 rounding of the sum (2^-8 relative) + float32 accumulation."""
    import torch
    from ppnet_amd import fused
    torch.manual_seed(11)
    tokens = 16 * 1000 + 48
    s0 = (torch.randn(tokens, 128, device="cuda") * 1.5).to(torch.bfloat16)
    a = torch.randn(tokens, 128, device="cuda").to(torch.bfloat16)
    proj = torch.nn.Linear(128, 128).cuda().to(torch.bfloat16)
    s = s0.clone()
    fused.nat128_proj_add_(s, a, proj)
    ref = s0.double() + a.double() @ proj.weight.double().t()
    err = (s.double() - ref).abs()
    assert bool((err <= 2.0 ** -8 * ref.abs() + 1e-3).all()), float(err.max())
    s2 = s0.clone()
    fused.nat128_proj_add_(s2, a, proj)
    assert torch.equal(s, s2)
