"""Loop B's drop-in import names (CPU): with ppnet_amd/dropin first on sys.path the import lines of the reference's
scripts — GenNet/predict.py:13 `from networks import AEViT as AE`, SegNet/nat.py:14 `from natten import
NeighborhoodAttention2D as NeighborhoodAttention` — resolve to the build's modules, with the constructor signatures and
state-dict key names the reference's call sites rely on (no compute: the kernels need a GPU)."""
import inspect
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "ppnet_amd", "dropin")
NAMES = ("networks", "natten")


@pytest.fixture()
def dropin_path(monkeypatch):
    monkeypatch.syspath_prepend(DROPIN)
    for m in NAMES:
        sys.modules.pop(m, None)
    yield
    for m in NAMES:
        sys.modules.pop(m, None)


def test_networks_exports_aevit(dropin_path):
    from networks import AEViT as AE           # predict.py:13, verbatim
    import networks
    import ppnet_amd.gennet as G
    assert AE is G.AEViT and networks.AE is G.AEViT
    assert list(inspect.signature(AE.__init__).parameters)[1:5] == ["img_channels", "out_channels", "img_resolution", "dim"]
    m = AE(1, 1, 224, 24)                       # predict.py:46
    keys = list(m.state_dict().keys())
    assert len(keys) == 87                      # 87 state-dict entries
    for k in ("conv_first.0.weight", "enc_conv.0.0.weight", "vit_blocks.0.attn.qkv.weight", "vit_blocks.2.mlp.fc2.bias",
              "dec_conv.0.0.weight", "conv_final.weight"):
        assert k in keys, k
    m.load_state_dict({"model": m.state_dict()}["model"])       # predict.py:51-53's call shape
    import torch
    t = torch.arange(2 * 6 * 4, dtype=torch.float32).reshape(2, 6, 4)
    f = networks.token2feature(t, 2, 3)         # base.py:43-46
    assert f.shape == (2, 4, 2, 3) and torch.equal(networks.feature2token(f), t)
    with pytest.raises(NotImplementedError):
        networks.AESwin


def test_natten_exports_neighborhood_attention(dropin_path):
    from natten import NeighborhoodAttention2D as NeighborhoodAttention      # nat.py:14, verbatim
    import ppnet_amd.na as NA
    assert NeighborhoodAttention is NA.NeighborhoodAttention2D
    attn = NeighborhoodAttention(128, kernel_size=7, dilation=2, num_heads=4, qkv_bias=True, qk_scale=None,
                                 attn_drop=0.0, proj_drop=0.0)              # nat.py:111-120
    assert sorted(attn.state_dict().keys()) == ["proj.bias", "proj.weight", "qkv.bias", "qkv.weight", "rpb"]
    assert tuple(attn.rpb.shape) == (4, 13, 13)
    import torch
    with pytest.raises(RuntimeError):           # no CPU fallback: the op fails loudly off the GPU
        attn(torch.zeros(1, 14, 14, 128))
