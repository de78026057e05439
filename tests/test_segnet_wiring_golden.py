"""SegNet's backbone wiring pinned to the REFERENCE's own code.

tests/golden/g16_nat_wiring.npz was written by tests/golden/make_fixtures.py, which imports SegNet/nat.py, SegNet/dinat.py and
SegNet/mmseg/ops/wrappers.py unmodified (sys.modules stubs for timm's DropPath, the mmcv / mmseg registry names and natten — the
attention op is the build's own statement of NATTEN's semantics, oracle/na_np.py, and stays "parity unpinned"; everything around
it is the reference's: ConvTokenizer nat.py:17-45, ConvDownsampler :48-59, Mlp :62-85, NATLayer :88-153 with and without
LayerScale, NATBlock :156-209, NAT.forward_tokens :316-324, resize / Upsample wrappers.py:8-51).

CPU: oracle/segnet_ref.py (the checker of every SegNet GPU test and the CPU baseline of bench.py) reproduces the reference's
per-level outputs in float64 to 1e-10.  GPU: ppnet_amd.segnet.NAT / DiNAT in float32, through the HIP kernels, to 2e-3; the x2
up-sampling kernel and the label kernel's resize against wrappers.py's outputs."""
import json
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.nn.functional as F  # noqa: E402

from tests._oracle_util import wiring_weights  # noqa: E402


@pytest.fixture(scope="module")
def g16(golden_dir):
    return np.load(os.path.join(golden_dir, "g16_nat_wiring.npz"))


def _case(g, name):
    cfg = json.loads(str(g[f"{name}/cfg"]))
    keys = [str(k) for k in g[f"{name}/keys"]]
    shapes = [tuple(json.loads(str(s))) for s in g[f"{name}/shapes"]]
    w = wiring_weights(keys, shapes, int(g[f"{name}/seed"][0]))
    # the regenerated weights ARE the ones the reference module was loaded with
    chk = np.array([[w[k].sum(), (w[k] ** 2).sum()] for k in keys])
    assert np.allclose(chk, g[f"{name}/checksum"], rtol=1e-13, atol=1e-13)
    ys = {i: g[f"{name}/y{i}"] for i in cfg["out_indices"]}
    return cfg, keys, w, g[f"{name}/x"], ys


def test_fixture_holds_both_networks(g16):
    assert sorted(str(c) for c in g16["cases"]) == ["dinat", "nat"]
    cfg, keys, w, x, ys = _case(g16, "dinat")
    assert x.shape == (2, 3, 96, 128) and [ys[i].shape for i in range(4)] == [(2, 32, 24, 32), (2, 64, 12, 16), (2, 128, 6, 8), (2, 256, 3, 4)]
    assert "levels.1.blocks.1.gamma2" in keys and "levels.2.downsample.reduction.weight" in keys and "norm3.bias" in keys
    assert "levels.3.downsample.reduction.weight" not in keys                 # nat.py:262: the last level has no downsampler
    cfg2, keys2, _, x2, ys2 = _case(g16, "nat")
    assert not any(k.endswith("gamma1") for k in keys2) and sorted(ys2) == [0, 2] and "norm1.weight" not in keys2


@pytest.mark.parametrize("name", ["dinat", "nat"])
def test_oracle_backbone_is_the_references_wiring(g16, name):
    """oracle/segnet_ref.py backbone_fp64 == the reference's NAT.forward on the same weights and input (float64)."""
    from oracle import segnet_ref as SR
    cfg, keys, w, x, ys = _case(g16, name)
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    outs = SR.backbone_fp64(sd, torch.from_numpy(x).double(), cfg["depths"], cfg["num_heads"], cfg["dilations"],
                            layer_scale=cfg["layer_scale"] is not None, prefix="")
    for i, want in ys.items():
        got = outs[i].numpy()
        assert got.shape == want.shape
        assert np.abs(got - want).max() < 1e-10 * max(1.0, np.abs(want).max()), (name, i)
    for i in range(len(cfg["depths"])):
        if i not in ys:
            assert outs[i] is None                                            # no norm{i}: the level has no output (nat.py:318-323)


def test_oracle_resizes_are_the_references_wrappers(g16):
    """What oracle/segnet_ref.py calls between the head's stages and at the end of encode_decode — F.interpolate with
    scale_factor=2 resp. size= — equals wrappers.py's Upsample (which turns the scale into a size first, :43-51) and resize."""
    a = torch.from_numpy(g16["resize/in"])
    up = F.interpolate(a, scale_factor=2, mode="bilinear", align_corners=False)
    assert np.array_equal(up.numpy(), g16["resize/upsample2x"])
    for key, size, ac in (("to_24x36", (24, 36), False), ("to_13x7", (13, 7), False), ("to_11x17_ac", (11, 17), True)):
        assert np.array_equal(F.interpolate(a, size=size, mode="bilinear", align_corners=ac).numpy(), g16["resize/" + key])


def test_cpu_product_modules_keep_the_references_keys(g16):
    """ppnet_amd.segnet.NAT / DiNAT take the reference's constructor arguments and load its state dict strictly (same key names,
    same shapes, same order of registration) — what lets a reference checkpoint drop in."""
    from ppnet_amd.segnet import NAT, DiNAT
    for name, cls in (("dinat", DiNAT), ("nat", NAT)):
        cfg, keys, w, _, _ = _case(g16, name)
        m = cls(**cfg)
        assert list(m.state_dict().keys()) == keys
        m.load_state_dict({k: torch.from_numpy(v).float() for k, v in w.items()}, strict=True)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["dinat", "nat"])
def test_gpu_nat_float32_vs_reference_outputs(g16, name):
    """The product backbone on the HIP kernels (float32 path: ppn_na2d_fwd, fused residual / LayerNorm kernels, library
    convolutions) against the reference module's float64 outputs."""
    from ppnet_amd.segnet import NAT, DiNAT
    cfg, keys, w, x, ys = _case(g16, name)
    m = (DiNAT if name == "dinat" else NAT)(**cfg)
    m.load_state_dict({k: torch.from_numpy(v).float() for k, v in w.items()}, strict=True)
    m = m.eval().cuda()
    with torch.no_grad():
        outs = m(torch.from_numpy(x).cuda())
    assert len(outs) == len(cfg["out_indices"])
    for i, o in zip(cfg["out_indices"], outs):
        got = o.float().cpu().double().numpy()
        want = ys[i]
        assert got.shape == want.shape
        err = np.abs(got - want).max()
        assert err < 2e-3 * max(1.0, np.abs(want).max()), (name, i, err)


@pytest.mark.gpu
def test_gpu_upsample_kernel_vs_reference_wrapper(g16):
    """ppn_upsample2x_nhwc (the SETR-UP head's Upsample stages, setr_up_head.py:62-66) against wrappers.py's Upsample output;
    the 5 fixture channels are padded to the kernel's 8."""
    from ppnet_amd import fused
    a = torch.from_numpy(g16["resize/in"]).float()
    a8 = torch.cat([a, torch.zeros(2, 3, 6, 9)], 1).cuda().contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        got = fused.upsample2x_nhwc(a8, False, None)
    assert np.abs(got.float().cpu().numpy()[:, :5] - g16["resize/upsample2x"]).max() < 1e-5
