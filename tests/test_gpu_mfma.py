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
    for (B, H, W, Cin, Cout) in [(3, 8, 8, 128, 512), (40, 16, 16, 64, 512), (2, 10, 6, 64, 264)]:
        x = torch.randn(B, Cin, H, W, device="cuda").to(BF).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(Cout, Cin, 3, 3, device="cuda") * (2.0 / (9 * Cin)) ** 0.5).to(BF)
        b = torch.randn(Cout, device="cuda") * 0.3
        w2 = torch.randn(2, Cout, device="cuda") * 0.1
        b2 = torch.randn(2, device="cuda")
        want = F.conv2d(F.relu(F.conv2d(x.float(), w.float(), b, 1, 1)), w2.view(2, Cout, 1, 1), b2)
        got = fused.conv3x3_relu_classify2(x, w.permute(0, 2, 3, 1).contiguous(), b.contiguous(), w2.contiguous(), b2)
        assert got.shape == (B, 2, H, W) and got.dtype == torch.float32
        _close(got, want, rel=2e-3)                                           # float32 out: only the bf16 operands round


@pytest.mark.parametrize("M,N,K,persistent", [(300, 264, 128, 0), (512, 256, 192, 0), (256 * 40, 768, 192, 256), (256 * 33, 512, 320, 256),
                                              (65536, 512, 512, 0), (65536, 512, 512, 256)])
def test_gemm_bf16_epilogues_vs_torch(M, N, K, persistent):
    from ppnet_amd import fused
    torch.manual_seed(M % 97 + N)
    a = torch.randn(M, K, device="cuda").to(BF)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    b = torch.randn(N, device="cuda")
    lin = a.float() @ w.float().t()
    _close(fused.gemm_bf16(a, w, b, "bias", persistent_blocks=persistent), lin + b)
    _close(fused.gemm_bf16(a, w, b, "bias_gelu", persistent_blocks=persistent), F.gelu(lin + b))
    c0 = torch.randn(M, N, device="cuda").to(BF)
    c = c0.clone()
    out = fused.gemm_bf16(a, w, None, "accum", out=c, persistent_blocks=persistent)
    assert out is c
    _close(c, c0.float() + lin)
