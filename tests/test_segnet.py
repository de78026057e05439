"""SegNet (DiNAT + SETR-UP): checkpoint key names / sizes on CPU; on the GPU a small NAT is compared with a
float64 CPU evaluation of the same weights in which the attention is the definition oracle (oracle/na_np.py;
NATTEN itself is absent — parity unpinned), and DiNAT-B + SETR-UP is run end to end at 256x256."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")


def test_dinat_base_checkpoint_layout():
    from ppnet_amd.segnet import SegNet
    m = SegNet()
    sd = m.state_dict()
    for k in ["backbone.patch_embed.proj.0.weight", "backbone.patch_embed.proj.1.bias", "backbone.patch_embed.norm.weight",
              "backbone.levels.0.blocks.0.attn.qkv.weight", "backbone.levels.0.blocks.1.attn.rpb",
              "backbone.levels.2.blocks.17.mlp.fc2.bias", "backbone.levels.2.blocks.0.gamma1",
              "backbone.levels.0.downsample.reduction.weight", "backbone.levels.2.downsample.norm.bias", "backbone.norm3.weight",
              "decode_head.norm.weight", "decode_head.up_convs.0.0.conv.weight", "decode_head.up_convs.3.0.bn.running_var",
              "decode_head.conv_seg.bias"]:
        assert k in sd, k
    assert "backbone.levels.3.downsample.reduction.weight" not in sd
    assert "decode_head.up_convs.0.0.conv.bias" not in sd                 # ConvModule drops the bias under a norm
    assert sd["backbone.levels.1.blocks.3.attn.rpb"].shape == (8, 13, 13)
    assert sd["backbone.levels.3.blocks.0.attn.qkv.weight"].shape == (3072, 1024)
    assert sd["decode_head.up_convs.0.0.conv.weight"].shape == (512, 1024, 3, 3)
    assert [b.attn.dilation for b in m.backbone.levels[1].blocks] == [1, 4, 1, 8]
    n = sum(p.numel() for p in m.backbone.parameters())
    assert 85e6 < n < 95e6                                                # DiNAT-Base ~ 90 M parameters


@pytest.mark.gpu
def test_small_nat_gpu_vs_oracle_composition():
    from oracle import na_np as NA
    from ppnet_amd.segnet import NAT
    assert torch.cuda.is_available()
    torch.manual_seed(0)
    cfg = dict(embed_dim=64, mlp_ratio=2.0, depths=[2, 2], num_heads=[2, 4], kernel_size=7,
               dilations=[[1, 2], [1, 1]], layer_scale=0.5, out_indices=(0, 1))
    m = NAT(**cfg).eval()
    x = torch.randn(2, 3, 64, 64)
    with torch.no_grad():
        got = [o.cpu().double().numpy() for o in m.cuda()(x.cuda())]
    # float64 CPU evaluation of the same weights with the oracle attention
    m = m.cpu().double()

    def oracle_attn(mod):
        def f(t):
            sd = {k: v.detach().numpy() for k, v in mod.state_dict().items()}
            return torch.tensor(NA.neighborhood_attention_2d(t.numpy(), sd["qkv.weight"], sd["qkv.bias"], sd["rpb"],
                                                             sd["proj.weight"], sd["proj.bias"], mod.num_heads, 7, mod.dilation))
        return f
    for lvl in m.levels:
        for blk in lvl.blocks:
            blk.attn.forward = oracle_attn(blk.attn)
    with torch.no_grad():
        want = [o.numpy() for o in m(x.double())]
    assert [g.shape for g in got] == [(2, 64, 16, 16), (2, 128, 8, 8)]
    for g, w in zip(got, want):
        assert np.abs(g - w).max() < 2e-3                                  # float32 GPU vs float64 reference


@pytest.mark.gpu
def test_dinat_base_end_to_end_256():
    from ppnet_amd.segnet import SegNet, normalize_images
    torch.manual_seed(0)
    m = SegNet().cuda().eval()
    img = torch.randint(0, 256, (2, 256, 256, 3), dtype=torch.uint8, device="cuda")
    with torch.no_grad():
        pred, logits = m(normalize_images(img), return_logits=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred16, logits16 = m(normalize_images(img), return_logits=True)
    assert pred.shape == (2, 256, 256) and pred.dtype == torch.int64 and logits.shape == (2, 2, 256, 256)
    assert set(torch.unique(pred).tolist()) <= {0, 1}
    assert torch.isfinite(logits).all()
    # bf16 projections / convolutions (fp32 accumulate): the 2-class mask agrees on almost every pixel
    assert float((pred == pred16).float().mean()) > 0.97
