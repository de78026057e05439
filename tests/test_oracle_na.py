"""Self-consistency of the neighbourhood-attention definition oracle (no reference fixtures exist: parity
unpinned).  d=1 interior queries must equal a dense unfold-based formulation; windows must be centred when
they can be, shifted inward otherwise, and dilated attention must equal d=1 attention on each sub-grid."""
import numpy as np
import pytest

from oracle import na_np as NA


def test_window_rules():
    pos, bias = NA.window(10, 32, 7, 1)
    assert pos.tolist() == list(range(7, 14)) and bias.tolist() == list(range(3, 10))
    pos, bias = NA.window(0, 32, 7, 1)
    assert pos.tolist() == list(range(0, 7)) and bias.tolist() == list(range(6, 13))      # key - query + 6
    pos, bias = NA.window(31, 32, 7, 1)
    assert pos.tolist() == list(range(25, 32)) and bias.tolist() == list(range(0, 7))
    pos, bias = NA.window(9, 30, 7, 4)            # group 1: 1,5,...,29 (8 members), member index 2
    assert pos.tolist() == [1, 5, 9, 13, 17, 21, 25] and bias.tolist() == [4, 5, 6, 7, 8, 9, 10]
    pos, bias = NA.window(29, 30, 7, 4)
    assert pos.tolist() == [5, 9, 13, 17, 21, 25, 29] and bias.tolist() == list(range(0, 7))
    for L, d in [(28, 4), (30, 4), (17, 2), (7, 1), (21, 3)]:
        for i in range(L):
            pos, bias = NA.window(i, L, 7, d)
            assert (pos >= 0).all() and (pos < L).all() and i in pos and ((pos - i) % d == 0).all()
            assert ((pos - i) // d + 6 == bias).all()


def test_interior_equals_dense_unfold():
    torch = pytest.importorskip("torch")
    rng = np.random.RandomState(0)
    B, nh, H, W, hd, K = 2, 2, 12, 11, 8, 7
    q, k, v = (rng.standard_normal((B, nh, H, W, hd)) for _ in range(3))
    rpb = rng.standard_normal((nh, 13, 13))
    got = NA.na2d_core(q, k, v, rpb, K, 1)
    tk = torch.tensor(k).permute(0, 1, 4, 2, 3).reshape(B * nh, hd, H, W)
    tv = torch.tensor(v).permute(0, 1, 4, 2, 3).reshape(B * nh, hd, H, W)
    uk = torch.nn.functional.unfold(tk, K).reshape(B, nh, hd, K * K, H - 6, W - 6)      # windows centred at i+3
    uv = torch.nn.functional.unfold(tv, K).reshape(B, nh, hd, K * K, H - 6, W - 6)
    qi = torch.tensor(q)[:, :, 3:H - 3, 3:W - 3]
    logit = torch.einsum("bhijc,bhcnij->bhijn", qi, uk) + torch.tensor(rpb)[:, 3:10, 3:10].reshape(1, nh, 1, 1, 49)
    want = torch.einsum("bhijn,bhcnij->bhijc", logit.softmax(-1), uv).numpy()
    assert np.abs(got[:, :, 3:H - 3, 3:W - 3] - want).max() < 1e-12


def test_dilated_equals_subgrid_attention():
    rng = np.random.RandomState(1)
    B, nh, H, W, hd, K, d = 1, 2, 16, 15, 4, 7, 2
    q, k, v = (rng.standard_normal((B, nh, H, W, hd)) for _ in range(3))
    rpb = rng.standard_normal((nh, 13, 13))
    got = NA.na2d_core(q, k, v, rpb, K, d)
    for gi in range(d):
        for gj in range(d):
            sub = NA.na2d_core(q[:, :, gi::d, gj::d], k[:, :, gi::d, gj::d], v[:, :, gi::d, gj::d], rpb, K, 1)
            assert np.abs(got[:, :, gi::d, gj::d] - sub).max() < 1e-13
