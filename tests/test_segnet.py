"""SegNet (DiNAT + SETR-UP): checkpoint key names / sizes on CPU; on the GPU a small NAT is compared with a
float64 CPU evaluation of the same weights in which the attention is the definition oracle (oracle/na_np.py;
NATTEN itself is absent — parity unpinned), and DiNAT-B + SETR-UP is run end to end at 256x256."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")


@pytest.fixture(autouse=True)
def _inference_mode():
    """Everything in this file is inference: under autograd recording the fused host functions would step aside for
    differentiable torch ops (ppnet_amd/fused.py recording()) and the HIP kernels would not be the thing tested."""
    with torch.no_grad():
        yield


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


def test_nat_upernet_default_config_layout():
    """The reference's default SegNet (SegNet/test.py:29-32): NAT-Base + UPerHead(channels=64)."""
    from ppnet_amd.segnet import NAT_BASE_UPER, SegNet
    m = SegNet(**NAT_BASE_UPER)
    sd = m.state_dict()
    for k in ["decode_head.psp_modules.0.1.conv.weight", "decode_head.psp_modules.3.1.bn.running_mean",
              "decode_head.bottleneck.conv.weight", "decode_head.lateral_convs.2.bn.weight", "decode_head.fpn_convs.0.conv.weight",
              "decode_head.fpn_bottleneck.conv.weight", "decode_head.conv_seg.weight"]:
        assert k in sd, k
    assert sd["decode_head.bottleneck.conv.weight"].shape == (64, 1024 + 4 * 64, 3, 3)
    assert sd["decode_head.fpn_bottleneck.conv.weight"].shape == (64, 256, 3, 3)
    assert all(b.attn.dilation == 1 for lvl in m.backbone.levels for b in lvl.blocks)       # NAT: no dilations
    assert m.backbone.compute_indices == (0, 1, 2, 3)                                        # UPerNet reads every level


def _reference_nat_forward(m, x):
    """float64 CPU evaluation of a ppnet_amd.segnet.NAT's weights, written out op by op from SegNet/nat.py:41-59,
    140-153,204-209,316-324 with the definition oracle as the attention (test-side reference; the product module
    itself is GPU-only)."""
    import torch.nn.functional as F
    from oracle import na_np as NA
    sd = {k: v.detach().cpu().double() for k, v in m.state_dict().items()}
    ln = lambda t, p: F.layer_norm(t, (t.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)
    x = F.conv2d(F.conv2d(x, sd["patch_embed.proj.0.weight"], sd["patch_embed.proj.0.bias"], 2, 1),
                 sd["patch_embed.proj.1.weight"], sd["patch_embed.proj.1.bias"], 2, 1).permute(0, 2, 3, 1)
    x = ln(x, "patch_embed.norm")
    outs = []
    for li, lvl in enumerate(m.levels):
        for bi, blk in enumerate(lvl.blocks):
            p = f"levels.{li}.blocks.{bi}"
            a = NA.neighborhood_attention_2d(ln(x, p + ".norm1").numpy(), sd[p + ".attn.qkv.weight"].numpy(),
                                             sd[p + ".attn.qkv.bias"].numpy(), sd[p + ".attn.rpb"].numpy(),
                                             sd[p + ".attn.proj.weight"].numpy(), sd[p + ".attn.proj.bias"].numpy(),
                                             blk.attn.num_heads, 7, blk.attn.dilation)
            x = x + sd[p + ".gamma1"] * torch.tensor(a)
            h = F.gelu(F.linear(ln(x, p + ".norm2"), sd[p + ".mlp.fc1.weight"], sd[p + ".mlp.fc1.bias"]))
            x = x + sd[p + ".gamma2"] * F.linear(h, sd[p + ".mlp.fc2.weight"], sd[p + ".mlp.fc2.bias"])
        outs.append(ln(x, f"norm{li}").permute(0, 3, 1, 2).numpy())
        if lvl.downsample is not None:
            x = ln(F.conv2d(x.permute(0, 3, 1, 2), sd[f"levels.{li}.downsample.reduction.weight"], None, 2, 1).permute(0, 2, 3, 1),
                   f"levels.{li}.downsample.norm")
    return outs


@pytest.mark.gpu
def test_small_nat_gpu_vs_oracle_composition():
    from ppnet_amd.segnet import NAT
    assert torch.cuda.is_available()
    torch.manual_seed(0)
    cfg = dict(embed_dim=64, mlp_ratio=2.0, depths=[2, 2], num_heads=[2, 4], kernel_size=7,
               dilations=[[1, 4], [2, 1]], layer_scale=0.5, out_indices=(0, 1))   # 16x16 d=4 and 8x8 d=2: padded grids
    m = NAT(**cfg).eval()
    x = torch.randn(2, 3, 64, 64)
    want = _reference_nat_forward(m, x.double())
    with torch.no_grad():
        got = [o.cpu().double().numpy() for o in m.cuda()(x.cuda())]
    assert [g.shape for g in got] == [(2, 64, 16, 16), (2, 128, 8, 8)]
    for g, w in zip(got, want):
        assert np.abs(g - w).max() < 2e-3                                  # float32 GPU vs float64 reference


@pytest.mark.gpu
def test_folded_levels_equal_unfolded():
    """NATBlock.fold(): LayerScale in the projection weights, biases carried as a float32 offset, residual adds inside the
    GEMMs (addmm_) — the same function as the unfolded level, to float32 rounding."""
    import copy
    from ppnet_amd.segnet import NAT
    torch.manual_seed(3)
    cfg = dict(embed_dim=64, mlp_ratio=2.0, depths=[3, 2], num_heads=[2, 4], kernel_size=7,
               dilations=[[1, 4, 1], [2, 1]], layer_scale=0.5, out_indices=(0, 1))
    m = NAT(**cfg).eval()
    with torch.no_grad():
        for p in m.parameters():                                  # non-trivial biases and scales everywhere
            if p.dim() == 1:
                p.uniform_(-0.5, 0.5)
        for lvl in m.levels:
            for blk in lvl.blocks:
                blk.gamma1.uniform_(0.2, 1.5); blk.gamma2.uniform_(0.2, 1.5)
    x = torch.randn(2, 3, 64, 64)
    f = copy.deepcopy(m)
    for lvl in f.levels:
        lvl.fold()
    with torch.no_grad():
        want = [o.float().cpu() for o in m.cuda()(x.cuda())]
        got = [o.float().cpu() for o in f.cuda()(x.cuda())]
    for g, w in zip(got, want):
        assert g.shape == w.shape and (g - w).abs().max() < 2e-4 * max(1.0, float(w.abs().max()))
    # bf16: the folded form keeps the offsets in float32 (not rounded by module.to(bfloat16))
    with torch.no_grad():
        gb = [o.float().cpu() for o in copy.deepcopy(f).to(torch.bfloat16)(x.cuda().to(torch.bfloat16))]
    for g, w in zip(gb, want):
        assert (g - w).abs().max() < 0.15 * max(1.0, float(w.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("C,dtype", [(128, "float32"), (256, "bfloat16"), (512, "bfloat16"), (1024, "bfloat16"), (64, "float32"),
                                     (24, "float32"), (24, "bfloat16"), (8, "float32"), (56, "bfloat16")])   # C <= 64: thread-per-row kernel
def test_fused_residual_layernorm_vs_torch(C, dtype):
    from ppnet_amd import fused
    dt = getattr(torch, dtype)
    torch.manual_seed(1)
    ln = torch.nn.LayerNorm(C, eps=1e-5).cuda().to(dt)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5); ln.bias.uniform_(-0.5, 0.5)
    x = torch.randn(3, 7, 5, C, device="cuda", dtype=dt)
    a = torch.randn(3, 7, 5, C, device="cuda", dtype=dt)
    g = (torch.rand(C, device="cuda") * 2).to(dt)
    tol = 1e-5 if dt == torch.float32 else 4e-2
    y = fused.layer_norm(x, ln)
    assert (y.float() - torch.nn.functional.layer_norm(x.float(), (C,), ln.weight.float(), ln.bias.float(), 1e-5)).abs().max() < tol
    want_x = x.float() + g.float() * a.float()
    x2, y2 = fused.residual_layer_norm(x.clone(), a, g, ln)
    assert (x2.float() - want_x).abs().max() < tol
    assert (y2.float() - torch.nn.functional.layer_norm(x2.float(), (C,), ln.weight.float(), ln.bias.float(), 1e-5)).abs().max() < tol
    x3, y3 = fused.residual_layer_norm(x.clone(), a, None, None)
    assert y3 is None and (x3.float() - (x.float() + a.float())).abs().max() < tol


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_upsample2x_relu_vs_torch(dtype):
    from ppnet_amd import fused
    dt = getattr(torch, dtype)
    torch.manual_seed(2)
    for (h, w) in ((9, 13), (1, 1), (1, 5), (2, 2), (16, 16)):            # the kernel works on 2 x 2 output blocks: borders and 1-wide images
        x = torch.randn(3, 64, h, w, device="cuda", dtype=dt).contiguous(memory_format=torch.channels_last)
        for relu in (False, True):
            got = fused.upsample2x_nhwc(x, relu)
            ref = torch.nn.functional.interpolate(torch.relu(x.float()) if relu else x.float(), scale_factor=2, mode="bilinear",
                                                  align_corners=False)
            assert got.shape == ref.shape
            assert (got.float() - ref).abs().max() < (1e-5 if dt == torch.float32 else 2e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_bias_act_and_biased_upsample_vs_torch(dtype):
    from ppnet_amd import fused
    dt = getattr(torch, dtype)
    torch.manual_seed(4)
    tol = 1e-5 if dt == torch.float32 else 3e-2
    for C in (24, 64):
        x = torch.randn(3, C, 9, 13, device="cuda", dtype=dt).contiguous(memory_format=torch.channels_last)
        b = torch.randn(C, device="cuda", dtype=dt)
        for slope in (0.01, 0.0, 1.0):
            ref = torch.nn.functional.leaky_relu(x.float() + b.float().view(1, -1, 1, 1), slope)
            got = fused.bias_act_(x.clone(memory_format=torch.channels_last), b, slope)
            assert (got.float() - ref).abs().max() < tol
        ref = torch.nn.functional.interpolate(torch.relu(x.float() + b.float().view(1, -1, 1, 1)), scale_factor=2, mode="bilinear",
                                              align_corners=False)
        assert (fused.upsample2x_nhwc(x, True, b).float() - ref).abs().max() < tol


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_fpn_resize_kernels_vs_the_framework_sequence(dtype):
    """UPerHead's top-down step and output assembly (uper_head.py:103-108, 117-127) as single kernels against the framework calls they
    replace, in the tensor's own dtype: interpolate(size=..., bilinear, align_corners=False) then add / cat.  Same arithmetic per
    output (PyTorch's source index and weights, float32 taps, one rounding per framework kernel); what is left is the compilers'
    choice of fused multiply-adds inside the four-tap sum: units in the last place of the taps (measured: 4.8e-7 in float32).
    The levels are the network's x2 / x4 / x8 pyramid and ragged ones (sizes that are no multiples of each other, a 1-wide level)."""
    from ppnet_amd import fused
    dt = getattr(torch, dtype)
    torch.manual_seed(6)
    F_ = torch.nn.functional
    tol = 2e-6 if dt == torch.float32 else 2.0 ** -6                       # a few ulp of the taps' magnitude (unit normal draws, |x| < 5)

    def close(got, want):
        return bool(((got.float() - want.float()).abs() <= tol).all())
    for B, C, sizes in ((3, 64, ((16, 16), (8, 8), (4, 4), (2, 2))), (2, 24, ((13, 9), (7, 5), (3, 3), (1, 2))), (1, 8, ((5, 5), (5, 5), (1, 1), (2, 4)))):
        lv = [torch.randn(B, C, h, w, device="cuda", dtype=dt).contiguous(memory_format=torch.channels_last) for (h, w) in sizes]
        want = torch.cat([lv[0]] + [F_.interpolate(t, size=sizes[0], mode="bilinear", align_corners=False) for t in lv[1:]], dim=1)
        got = fused.resize_concat4(lv)
        assert got.shape == want.shape and got.permute(0, 2, 3, 1).is_contiguous()
        assert close(got, want), (B, C, sizes, (got.float() - want.float()).abs().max().item())
        assert torch.equal(got[:, :C], lv[0])                              # the finest level is a copy
    # the pyramid pooling module's two kernels: every pool scale in one launch, the input + the pooled maps resized back in one
    x = torch.randn(3, 64, 8, 8, device="cuda", dtype=dt).contiguous(memory_format=torch.channels_last)
    for xx, scales in ((x, (1, 2, 3, 6)), (x[:, :, :7, :5].contiguous(memory_format=torch.channels_last), (2, 4, 7))):
        got = fused.adaptive_pools(xx, scales)
        for s_, gt in zip(scales, got):
            want = F_.adaptive_avg_pool2d(xx, s_)
            assert gt.shape == want.shape and close(gt, want), (s_, (gt.float() - want.float()).abs().max().item())
        small = [torch.randn(3, 24, s_, s_, device="cuda", dtype=dt).contiguous(memory_format=torch.channels_last) for s_ in scales]
        want = torch.cat([xx] + [F_.interpolate(t, size=xx.shape[2:], mode="bilinear", align_corners=False) for t in small], dim=1)
        got = fused.resize_concat([xx] + small)
        assert got.shape == want.shape and close(got, want), (scales, (got.float() - want.float()).abs().max().item())
    for (h, w) in ((8, 8), (1, 1), (5, 3)):
        coarse = torch.randn(2, 64, h, w, device="cuda", dtype=dt).contiguous(memory_format=torch.channels_last)
        fine = torch.randn(2, 64, 2 * h, 2 * w, device="cuda", dtype=dt).contiguous(memory_format=torch.channels_last)
        want = fine + F_.interpolate(coarse, size=(2 * h, 2 * w), mode="bilinear", align_corners=False)
        got = fused.upsample2x_add_(fine, coarse)
        assert got is fine and close(got, want), (h, w, (got.float() - want.float()).abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_single_channel_convs_vs_torch(dtype):
    from ppnet_amd import fused
    dt = getattr(torch, dtype)
    torch.manual_seed(5)
    tol = 2e-5 if dt == torch.float32 else 3e-2
    for C, H, W in ((24, 37, 50), (8, 16, 16), (32, 5, 70)):
        x1 = (torch.rand(3, 1, H, W, device="cuda") > 0.5).to(dt)
        w = torch.randn(C, 1, 3, 3, device="cuda") * 0.3
        b = torch.randn(C, device="cuda")
        ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(x1.float(), w, b, 1, 1), 0.01)
        got = fused.conv3x3_c1(x1, w.contiguous(), b.contiguous(), 0.01)
        assert got.shape == ref.shape and got.is_contiguous(memory_format=torch.channels_last)
        assert (got.float() - ref).abs().max() < tol
        xc = torch.randn(3, C, H, W, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
        w1 = torch.randn(1, C, 3, 3, device="cuda") * 0.2
        ref1 = torch.nn.functional.conv2d(xc.float(), w1, torch.tensor([0.37], device="cuda"), 1, 1)
        got1 = fused.conv3x3_to1(xc, w1.contiguous(), 0.37)
        assert got1.shape == ref1.shape and (got1.float() - ref1).abs().max() < tol * max(1.0, float(ref1.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_fused_segmentation_tail_vs_torch(dtype):
    """ppn_seg_labels_2class against the three library ops it replaces (interpolate x2, interpolate to size, softmax+argmax)."""
    from ppnet_amd import fused
    dt = getattr(torch, dtype)
    torch.manual_seed(6)
    lo = (torch.randn(5, 2, 16, 16, device="cuda") * 2).to(dt)
    mid = torch.nn.functional.interpolate(lo, scale_factor=2, mode="bilinear", align_corners=False)
    full = torch.nn.functional.interpolate(mid, (64, 64), mode="bilinear", align_corners=False)
    want = torch.nn.functional.softmax(full.float(), dim=1).argmax(dim=1)
    got = fused.seg_labels_2class(lo, (64, 64))
    assert got.shape == want.shape and got.dtype == torch.uint8
    # identical up to exact ties / last-ulp differences of the interpolation (library build contracts FMAs)
    assert float((got.long() == want).float().mean()) > 0.9995
    # non-square, other ratio (R = 224: 28 -> 56 -> 224 is x4 on the second stage)
    lo2 = (torch.randn(2, 2, 7, 9, device="cuda")).to(dt)
    full2 = torch.nn.functional.interpolate(torch.nn.functional.interpolate(lo2, scale_factor=2, mode="bilinear", align_corners=False),
                                            (56, 72), mode="bilinear", align_corners=False)
    want2 = torch.nn.functional.softmax(full2.float(), dim=1).argmax(dim=1)
    assert float((fused.seg_labels_2class(lo2, (56, 72)).long() == want2).float().mean()) > 0.999


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_grid_to_image_equals_render_then_normalise(dtype):
    from ppnet_amd import edage, fused
    from ppnet_amd.segnet import IMG_MEAN, IMG_STD, normalize_images
    dt = getattr(torch, dtype)
    torch.manual_seed(7)
    codes = torch.tensor([0, 255, 128], dtype=torch.uint8, device="cuda")
    grid = codes[torch.randint(0, 3, (3, 64, 96), device="cuda")]
    want = normalize_images(edage.grid_to_rgb(grid) * 255.0).to(dt)
    got = fused.grid_to_image(grid, IMG_MEAN, IMG_STD, dt)
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, want)


@pytest.mark.gpu
def test_dinat_base_end_to_end_256():
    from ppnet_amd.segnet import SegNet, normalize_images
    torch.manual_seed(0)
    m = SegNet().cuda().eval()
    img = torch.randint(0, 256, (2, 256, 256, 3), dtype=torch.uint8, device="cuda")
    import copy
    m16 = copy.deepcopy(m).to(torch.bfloat16)                   # what PPNet runs: bf16 weights and activations
    with torch.no_grad():
        pred, logits = m(normalize_images(img), return_logits=True)
        pred16, logits16 = m16(normalize_images(img).to(torch.bfloat16), return_logits=True)
    assert pred.shape == (2, 256, 256) and pred.dtype == torch.int64 and logits.shape == (2, 2, 256, 256)
    assert set(torch.unique(pred).tolist()) <= {0, 1}
    assert torch.isfinite(logits).all()
    # bf16 projections / convolutions (fp32 accumulate): the 2-class mask agrees on almost every pixel
    assert float((pred == pred16).float().mean()) > 0.97
    # the fused output tail (labels_u8) gives the labels of forward()
    with torch.no_grad():
        lab = m.labels_u8(normalize_images(img))
    assert lab.dtype == torch.uint8 and float((lab.long() == pred).float().mean()) > 0.9995


@pytest.mark.gpu
def test_ppnet_pipeline_runs_at_512_and_224():
    """BASELINE config 5 resolution (512: AE-ViT with down_time = 4, 128 x 128 SegNet tokens) and the reference default (224):
    the whole plan() path on generator output, and config 5's PPNet column — success rate and plan length / target length
    (process_map.py:496-503; updated_geometric_planner.py:260-277) — on ridge heat maps along the labels, where plans exist
    (no trained weights ship with the reference: the networks' own heat map is noise)."""
    from ppnet_amd import edage, evaluate
    from ppnet_amd.ppnet import PPNet
    dev = torch.device("cuda:0")
    for R in (512, 224):
        pb = edage.generate_paths(4, R, 50, 3, seed=3, device=dev)
        mb = edage.generate_maps(pb, 4, 5, 20, seed=3)
        torch.manual_seed(0)
        model = PPNet(R).to(dev).eval()
        init, end = mb.segpoint[:, 0].contiguous(), mb.segpoint[:, 10].contiguous()
        out = model.plan(mb.grid, init, end, mb.obstacles, mb.n_obstacles[:, 0].contiguous())
        torch.cuda.synchronize()
        assert out["ok"].shape == (16,) and out["waypoints"].shape[0] == 16 and out["collision"].dtype == torch.bool
        mask = model.segment(mb.grid)
        assert mask.shape == (16, R, R)
        heat = model.heatmap(mask)
        assert heat.shape == (16, R, R) and heat.dtype == torch.uint8
        assert int(heat.reshape(16, -1).max(dim=1).values.min()) == 255         # per-sample min-max normalisation
        ridge = evaluate.label_heatmaps(pb, mb, 4, sigma=2.0)                    # a narrow ridge: the greedy walk meanders on a plateau
        res = model.plan_tail(ridge, init, end, mb.obstacles, mb.n_obstacles[:, 0].contiguous())
        ev = evaluate.evaluate_plans(res, pb.length.repeat_interleave(4) * R / 50)
        print(f"R={R}: {ev}")
        assert ev["extract_ok"] >= 0.9 and ev["success"] >= 0.75 and 0.9 < ev["length_ratio"] < 1.3
        # a waypoint beyond column 224 is not a collision on a larger map (the reference's constant is its map size)
        if R == 512:
            assert bool((res["waypoints"][res["ok"]][:, :, 1].max() > 224)) and bool(res["success"].any())


@pytest.mark.gpu
def test_nat_upernet_end_to_end_and_bn_folding():
    import copy
    from ppnet_amd.segnet import NAT_BASE_UPER, SegNet, normalize_images
    torch.manual_seed(1)
    m = SegNet(**NAT_BASE_UPER).cuda().eval()
    with torch.no_grad():
        for mod in m.modules():                                   # non-trivial BN statistics so folding is exercised
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.uniform_(-0.2, 0.2); mod.running_var.uniform_(0.5, 1.5)
        img = normalize_images(torch.randint(0, 256, (2, 224, 224, 3), dtype=torch.uint8, device="cuda"))
        pred, logits = m(img, return_logits=True)
        f = copy.deepcopy(m).prepare_inference()
        _, logits_f = f(img, return_logits=True)
    assert pred.shape == (2, 224, 224) and logits.shape == (2, 2, 224, 224) and torch.isfinite(logits).all()
    assert (logits - logits_f).abs().max() < 1e-3 * max(1.0, float(logits.abs().max()))


@pytest.mark.gpu
def test_nat_upernet_vs_fp64_composition():
    """NAT-Base + UPerHead (the reference's default SegNet config) against the float64 op-by-op
    composition of uper_head.py:76-127 + psp_head.py:48-60 + nat.py (oracle/segnet_ref.py), unprepared and prepared
    (BN folded, levels folded).  Attention semantics: the definition oracle (parity unpinned)."""
    import copy
    from oracle import segnet_ref as SR
    from ppnet_amd.segnet import NAT_BASE_UPER, SegNet, normalize_images
    torch.manual_seed(2)
    m = SR.randomize(SegNet(**NAT_BASE_UPER).eval(), seed=3).cuda()
    img = normalize_images(torch.randint(0, 256, (2, 128, 128, 3), dtype=torch.uint8, device="cuda"))
    with torch.no_grad():
        want = SR.segnet_logits_fp64(m, NAT_BASE_UPER, img, head="uper")
        got = m.encode_decode(img).double()
        got_f = copy.deepcopy(m).prepare_inference().encode_decode(img).double()
    scale = max(1.0, float(want.abs().max()))
    assert want.shape == (2, 2, 128, 128) and float(want.std()) > 1e-3
    assert float((got - want).abs().max()) < 2e-3 * scale
    assert float((got_f - want).abs().max()) < 2e-3 * scale


# ------------------------------------------------------------------ the reference's calling convention (mmseg)
def _reference_dinat_base_cfg():
    """configs/_base_/models/dinat.py:2-46 merged with configs/dinat/dinat_base.py:5-24 the way mmcv merges `_base_` files
    (dict values update key by key), as plain data."""
    norm_cfg = dict(type='SyncBN', requires_grad=True)
    return dict(
        type='EncoderDecoder', pretrained=None,
        backbone=dict(type='DiNAT', embed_dim=128, mlp_ratio=2.0, depths=[3, 4, 18, 5], num_heads=[4, 8, 16, 32], drop_path_rate=0.5,
                      kernel_size=7, layer_scale=1e-5,
                      dilations=[[1, 16, 1], [1, 4, 1, 8], [1, 2, 1, 3, 1, 4, 1, 2, 1, 3, 1, 4, 1, 2, 1, 3, 1, 4], [1, 2, 1, 2, 1]],
                      out_indices=(0, 1, 2, 3), qkv_bias=True, qk_scale=None, drop_rate=0., attn_drop_rate=0., in_patch_size=4,
                      frozen_stages=-1),
        decode_head=dict(type='SETRUPHead', norm_layer=dict(type='LN', eps=1e-6, requires_grad=True), num_convs=4, up_scale=2,
                         kernel_size=3,
                         init_cfg=[dict(type='Constant', val=1.0, bias=0, layer='LayerNorm'),
                                   dict(type='Normal', std=0.01, override=dict(name='conv_seg'))],
                         in_channels=1024, channels=512, in_index=-1, num_classes=2, norm_cfg=norm_cfg, align_corners=False,
                         loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0)),
        train_cfg=dict(), test_cfg=dict(mode='whole'))


def test_from_config_builds_the_reference_models():
    from ppnet_amd.segnet import FCNHead, SegNet
    m = SegNet.from_config(_reference_dinat_base_cfg())
    assert sorted(m.state_dict()) == sorted(SegNet().state_dict())                # same modules as the built-in DINAT_BASE
    assert m.backbone.compute_indices == (3,) and m.auxiliary_head is None and m.test_cfg == {"mode": "whole"}
    assert [b.drop_path_rate for b in m.backbone.levels[0].blocks][0] == 0.0 and m.backbone.levels[3].blocks[-1].drop_path_rate == 0.5
    # configs/nat/setr_up_nat_base.py:6-42 over _base_/models/nat.py:3-38: NAT backbone, five SETR-UP stages, FCN auxiliary head
    norm_cfg = dict(type='SyncBN', requires_grad=True)
    cfg = dict(model=dict(
        type='EncoderDecoder', pretrained=None,
        backbone=dict(type='NAT', embed_dim=128, mlp_ratio=2.0, depths=[3, 4, 18, 5], num_heads=[4, 8, 16, 32], drop_path_rate=0.5,
                      kernel_size=7, layer_scale=1e-5, out_indices=(0, 1, 2, 3), qkv_bias=True, qk_scale=None, drop_rate=0.,
                      attn_drop_rate=0., in_patch_size=4, frozen_stages=-1),
        decode_head=dict(type='SETRUPHead', norm_layer=dict(type='LN', eps=1e-6, requires_grad=True), num_convs=5, up_scale=2,
                         kernel_size=3, in_channels=1024, channels=512, in_index=-1, num_classes=2, norm_cfg=norm_cfg,
                         align_corners=False, loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0)),
        auxiliary_head=dict(type='FCNHead', in_channels=512, in_index=2, channels=256, num_convs=1, concat_input=False,
                            dropout_ratio=0.1, num_classes=2, norm_cfg=norm_cfg, align_corners=False,
                            loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=0.4)),
        train_cfg=dict(), test_cfg=dict(mode='whole')))
    m = SegNet.from_config(cfg)
    assert isinstance(m.auxiliary_head, FCNHead) and m.auxiliary_head.loss_weight == 0.4
    assert m.backbone.compute_indices == (2, 3)                                    # the auxiliary head reads level 2 while it trains
    sd = m.state_dict()
    assert sd["auxiliary_head.convs.0.conv.weight"].shape == (256, 512, 3, 3) and "auxiliary_head.convs.0.bn.running_mean" in sd
    assert sd["auxiliary_head.conv_seg.weight"].shape == (2, 256, 1, 1) and "auxiliary_head.conv_cat.conv.weight" not in sd
    assert len(m.decode_head.up_convs) == 5 and all(b.attn.dilation == 1 for lvl in m.backbone.levels for b in lvl.blocks)
    m.prepare_inference()
    assert m.backbone.compute_indices == (3,)                                      # a training-time branch: not evaluated at inference
    with pytest.raises(NotImplementedError):
        SegNet.from_config(dict(type='CascadeEncoderDecoder', backbone={}, decode_head={}))
    with pytest.raises(NotImplementedError):
        SegNet.from_config(dict(_reference_dinat_base_cfg(), test_cfg=dict(mode='slide', crop_size=(64, 64), stride=(32, 32))))


class _Meta:
    """Stand-in for mmcv's DataContainer around img_metas (mmseg/apis/test.py:97: `data['img_metas'][0].data[0]`)."""

    def __init__(self, metas):
        self.data = [metas]


def test_forward_is_the_mmseg_harness_call_on_cpu():
    """`result = model(return_loss=False, **data)` with data = {'img': [x], 'img_metas': [[meta, ...]]} exactly as
    single_gpu_test calls it (mmseg/apis/test.py:89-95), on the CPU: the network itself is GPU-only, so encode_decode is replaced
    by a fixed logit function of the image and the harness logic — nesting, rescale to ori_shape, flip, softmax / argmax, the
    list of int64 arrays, the type errors of base.py:76-84 — is what is checked."""
    import torch.nn.functional as F
    from ppnet_amd.segnet import SegNet
    m = SegNet(backbone=dict(embed_dim=16, mlp_ratio=2.0, depths=[1, 1, 1, 1], num_heads=[1, 1, 2, 4], kernel_size=7),
               decode_head=dict(in_channels=128, channels=16, num_convs=2, up_scale=2, num_classes=2)).eval()

    def fake_logits(img):
        s = F.avg_pool2d(img.float().mean(1, keepdim=True), 3, 1, 1)
        return torch.cat([s, -s], dim=1)
    m.encode_decode = fake_logits
    torch.manual_seed(0)
    x = torch.randn(3, 3, 32, 32)
    meta = dict(ori_shape=(32, 32, 3), img_shape=(32, 32, 3), pad_shape=(32, 32, 3), scale_factor=1.0, flip=False)
    data = dict(img=[x], img_metas=[[meta] * 3])
    result = m(return_loss=False, **data)
    assert isinstance(result, list) and len(result) == 3
    want = fake_logits(x).softmax(1).argmax(1).numpy()
    for r, w in zip(result, want):
        assert isinstance(r, np.ndarray) and r.dtype == np.int64 and r.shape == (32, 32) and np.array_equal(r, w)
    assert np.array_equal(np.stack(m(return_loss=False, img=[x], img_metas=[_Meta([meta] * 3)])), want)     # DataContainer form
    # rescale to ori_shape (whole_inference, encoder_decoder.py:200-217)
    meta48 = dict(meta, ori_shape=(48, 40, 3))
    r48 = m(return_loss=False, img=[x], img_metas=[[meta48] * 3])
    assert r48[0].shape == (48, 40)
    w48 = F.interpolate(fake_logits(x), (48, 40), mode="bilinear", align_corners=False).softmax(1).argmax(1).numpy()
    assert np.array_equal(np.stack(r48), w48)
    # flip augmentation: probabilities flipped back and averaged with the plain pass (aug_test, encoder_decoder.py:267-285)
    metaf = dict(meta, flip=True, flip_direction="horizontal")
    ra = m(return_loss=False, img=[x, x.flip(3)], img_metas=[[meta] * 3, [metaf] * 3])
    p = (fake_logits(x).softmax(1) + fake_logits(x.flip(3)).softmax(1).flip(3)) / 2
    assert np.array_equal(np.stack(ra), p.argmax(1).numpy())
    # the batched tensor form of this build is unchanged
    assert torch.equal(m(x), torch.from_numpy(want))
    with pytest.raises(TypeError):
        m(return_loss=False, img=x, img_metas=[[meta] * 3])
    with pytest.raises(TypeError):
        m.forward_test(x, [[meta]])
    with pytest.raises(ValueError):
        m(return_loss=False, img=[x, x], img_metas=[[meta] * 3])


@pytest.mark.gpu
def test_mmseg_call_equals_labels_u8_on_gpu():
    """DiNAT + SETR-UP on the GPU under the harness call: the list of int64 label maps equals labels_u8 (the fused tail), and the
    generic path (logits -> softmax -> argmax) agrees with it wherever the two classes are not tied to the last bf16 bit."""
    from ppnet_amd.segnet import SegNet
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    m = SegNet(backbone=dict(embed_dim=32, mlp_ratio=2.0, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], kernel_size=7,
                             dilations=[[1, 2], [1, 2], [1, 2], [1, 1]], layer_scale=1e-5),
               decode_head=dict(in_channels=256, channels=32, num_convs=4, up_scale=2, num_classes=2)).to(dev).eval()
    x = torch.randn(4, 3, 64, 64, device=dev)
    meta = dict(ori_shape=(64, 64, 3), img_shape=(64, 64, 3), pad_shape=(64, 64, 3), scale_factor=1.0, flip=False)
    res = m(return_loss=False, img=[x], img_metas=[[meta] * 4])
    assert isinstance(res, list) and len(res) == 4 and all(r.dtype == np.int64 and r.shape == (64, 64) for r in res)
    assert np.array_equal(np.stack(res), m.labels_u8(x).cpu().numpy().astype(np.int64))
    prob = m.inference(x, [meta] * 4, True)
    assert (torch.from_numpy(np.stack(res)).to(dev) == prob.argmax(1)).float().mean().item() > 0.995
    # forward_train: the loss dict of encoder_decoder.py:122-152
    with torch.enable_grad():
        m.train()
        losses = m(img=x, img_metas=[meta] * 4, gt_semantic_seg=torch.randint(0, 2, (4, 1, 64, 64), device=dev))
        assert set(losses) == {"decode.loss_ce", "decode.acc_seg"} and losses["decode.loss_ce"].requires_grad
        losses["decode.loss_ce"].backward()


@pytest.mark.gpu
def test_nat_upernet_prepared_bf16_on_own_kernels_vs_fp64_composition():
    """The reference's default SegNet config (NAT-Base + UPerHead, SegNet/test.py:29-32) as bench.py --segnet nat_uper runs it:
    prepared (BatchNorm and LayerScale folded), bfloat16, every convolution / projection on the build's own kernels
    (UPerHead._forward_mfma, the LN-folded persistent GEMMs, the MFMA attention) — against the float64 op-by-op composition of
    uper_head.py:76-127 + psp_head.py:48-60 + nat.py.  Tolerance as for the SETR-UP configuration (tests/test_ppnet_config3.py): rms
    logit error below 1 % of the logit rms, labels equal wherever the float64 margin exceeds 6 x the rms error."""
    import copy
    from oracle import segnet_ref as SR
    from ppnet_amd.segnet import NAT_BASE_UPER, SegNet, UPerHead, normalize_images
    torch.manual_seed(2)
    m = SR.randomize(SegNet(**NAT_BASE_UPER).eval(), seed=3).cuda()
    img = normalize_images(torch.randint(0, 256, (4, 256, 256, 3), dtype=torch.uint8, device="cuda"))
    want = SR.segnet_logits_fp64(m, NAT_BASE_UPER, img, head="uper")
    f = copy.deepcopy(m).prepare_inference().to(torch.bfloat16)
    calls = []
    orig = UPerHead._forward_mfma
    UPerHead._forward_mfma = lambda self, inputs: (calls.append(1), orig(self, inputs))[1]
    try:
        got = f.encode_decode(img.to(torch.bfloat16)).double()
    finally:
        UPerHead._forward_mfma = orig
    assert calls, "the prepared bf16 head did not take the own-kernel path"
    err = got - want
    rms = float(err.pow(2).mean().sqrt())
    assert rms < 0.01 * float(want.pow(2).mean().sqrt()), (rms, float(want.pow(2).mean().sqrt()))
    margin = (want[:, 0] - want[:, 1]).abs()
    sure = margin > 6 * rms
    assert float(sure.float().mean()) > 0.5
    assert bool((got.argmax(1) == want.argmax(1))[sure].all())


@pytest.mark.gpu
def test_nat_upernet_labels_from_occupancy_codes():
    """labels_u8 on the u8 occupancy codes (palette tokenizer, no rendered image) equals labels_u8 on the rendered, normalised image
    for the NAT + UPerNet configuration (the generic tail: resize -> softmax -> argmax), up to bf16 ties."""
    from ppnet_amd import fused
    from ppnet_amd.segnet import IMG_MEAN, IMG_STD, NAT_BASE_UPER, SegNet
    torch.manual_seed(4)
    net = SegNet(**NAT_BASE_UPER).cuda().eval().prepare_inference().to(torch.bfloat16)
    g = torch.full((2, 128, 128), 255, dtype=torch.uint8, device="cuda")
    g[:, 30:60, 40:90] = 0
    g[:, 100:107, 10:17] = 128
    a = net.labels_u8(g)
    b = net.labels_u8(fused.grid_to_image(g, IMG_MEAN, IMG_STD, torch.bfloat16))
    assert a.shape == (2, 128, 128) and a.dtype == torch.uint8
    assert float((a == b).float().mean()) > 0.98
