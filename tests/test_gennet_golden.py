"""Pins ppnet_amd.gennet.AEViT (own module, reference key names) against outputs of the reference's
GenNet/networks/ae_vit.py captured in tests/golden/g13_aevit.npz.  CPU, float32: the same PyTorch runs both,
differences come only from fused attention vs explicit softmax (<= 2e-5 on O(1) activations)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")


def _model_and_data(golden_dir, R):
    from ppnet_amd.gennet import AEViT
    g = np.load(os.path.join(golden_dir, "g13_aevit.npz"))
    m = AEViT(1, 1, R, 24).eval()
    sd = {k[len(f"R{R}/w/"):]: torch.tensor(g[k]) for k in g.files if k.startswith(f"R{R}/w/")}
    assert set(sd.keys()) == set(m.state_dict().keys())           # same 87 (R=224) checkpoint entries
    m.load_state_dict(sd, strict=True)
    return m, torch.tensor(g[f"R{R}/x"]).float(), g[f"R{R}/y"]


@pytest.mark.parametrize("R", [64, 224, 256, 512])
def test_aevit_cpu_matches_reference(golden_dir, R):
    """R = 256 is what bench.py runs (config 3); R = 512 (config 5) has down_time = 4: one conv / deconv stage more."""
    m, x, y = _model_and_data(golden_dir, R)
    if R == 224:
        assert len(m.state_dict()) == 87 and sum(p.numel() for p in m.parameters()) == 53713
    assert len(m.enc_conv) == len(m.dec_conv) == {64: 1, 224: 3, 256: 3, 512: 4}[R]
    with torch.no_grad():
        got = m(x).numpy()
    assert got.shape == y.shape
    assert np.abs(got - y).max() < 2e-5 * max(1.0, np.abs(y).max())


@pytest.mark.gpu
@pytest.mark.parametrize("R", [64, 224, 256, 512])
def test_aevit_gpu_matches_reference(golden_dir, R):
    assert torch.cuda.is_available()
    m, x, y = _model_and_data(golden_dir, R)
    m = m.cuda()
    with torch.no_grad():
        got = m(x.cuda()).float().cpu().numpy()
    assert np.abs(got - y).max() < 1e-3 * max(1.0, np.abs(y).max())          # float32 on MIOpen / rocBLAS


@pytest.mark.gpu
@pytest.mark.parametrize("R", [64, 224, 256, 512])
def test_aevit_prepared_gpu_matches_reference(golden_dir, R):
    """prepare_inference(): BN folded, bias + LeakyReLU and the ViT LayerNorms on the HIP kernels."""
    m, x, y = _model_and_data(golden_dir, R)
    m = m.prepare_inference().cuda()
    with torch.no_grad():
        got = m(x.cuda().contiguous(memory_format=torch.channels_last)).float().cpu().numpy()
    assert np.abs(got - y).max() < 1e-3 * max(1.0, np.abs(y).max())


@pytest.mark.parametrize("R", [64])
def test_bn_folding_is_exact(golden_dir, R):
    import copy
    m, x, y = _model_and_data(golden_dir, R)
    f = copy.deepcopy(m).prepare_inference()
    with torch.no_grad():
        assert np.abs(f(x).numpy() - y).max() < 5e-5 * max(1.0, np.abs(y).max())
    assert not any(isinstance(mod, torch.nn.BatchNorm2d) for mod in f.modules())


def test_heatmap_normalisation():
    from ppnet_amd.gennet import normalize_heatmap_u8
    y = torch.tensor([[[[0.0, 1.0], [2.0, 4.0]]], [[[-1.0, -1.0], [0.0, 1.0]]]])
    u = normalize_heatmap_u8(y)
    assert u.tolist() == [[[0, 63], [127, 255]], [[0, 0], [127, 255]]]        # per sample, not per batch
