"""GPU parity for the planner tail (rows B9/B10) and the Philox draw kernel: the HIP path against golden
vectors from the reference's process_map.py, against Pillow (resize), and against the CPU oracle."""
import os

import numpy as np
import pytest

from oracle import philox_np as px
from oracle import plan_np as PN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_philox_device_matches_oracle(dev):
    from ppnet_amd import philox
    for seed, stream, inst, first, n in [(0, 1, 0, 0, 10021), (123456789012345, 3, 77, 5, 999), (7, 4, 2 ** 33 + 5, 0, 61)]:
        got = philox.doubles_device(seed, stream, inst, first, n, dev).cpu().numpy()
        assert np.array_equal(got, px.doubles(seed, stream, inst, first, n))          # bit-exact


def test_collision_segments_vs_reference_golden(dev, golden_dir):
    import torch
    from ppnet_amd import plan
    g = np.load(os.path.join(golden_dir, "g11_collision.npz"))
    n = len(g["hit"])
    off = np.concatenate([[0], np.cumsum(g["n_obs"])]).astype(np.int32)
    hit = plan.collision_segments(torch.tensor(g["s"], device=dev), torch.tensor(g["e"], device=dev),
                                  torch.arange(n, dtype=torch.int32, device=dev),
                                  torch.tensor(g["obs"].astype(np.float32), device=dev), torch.tensor(off, device=dev),
                                  float(g["clearance"][0]))
    assert hit.cpu().numpy().astype(np.int8).tolist() == g["hit"].tolist()              # bit-exact mask


def test_collision_bound_follows_resolution(dev):
    """The reference's 224 is its map size (process_map.py:384-387): at R=256 a waypoint beyond column 224 with no obstacle
    near it is free with bound=R and a hit with the reference's constant; random segments agree with the oracle at bound=R."""
    import torch
    from ppnet_amd import plan
    s = torch.tensor([[100.0, 230.0], [10.0, 10.0]], device=dev)
    e = torch.tensor([[120.0, 250.0], [20.0, 30.0]], device=dev)
    prob = torch.zeros(2, dtype=torch.int32, device=dev)
    obs = torch.tensor([[30.0, 200.0, 4.0]], device=dev)
    off = torch.tensor([0, 1], dtype=torch.int32, device=dev)
    assert plan.collision_segments(s, e, prob, obs, off, 256 / 50, bound=256).cpu().tolist() == [False, False]
    assert plan.collision_segments(s, e, prob, obs, off, 256 / 50).cpu().tolist() == [True, False]
    rng = np.random.RandomState(3)
    n, P, R = 600, 12, 256
    sv = (rng.random_sample((n, 2)) * (R + 20) - 10).astype(np.float32)
    ev = (sv + rng.standard_normal((n, 2)) * 25).astype(np.float32)
    pr = rng.randint(0, P, n).astype(np.int32)
    cnt = rng.randint(0, 30, P)
    offs = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    ob = np.concatenate([rng.random_sample((offs[-1], 2)) * R, rng.random_sample((offs[-1], 1)) * 12], axis=1).astype(np.float32)
    hit = plan.collision_segments(torch.tensor(sv, device=dev), torch.tensor(ev, device=dev), torch.tensor(pr, device=dev),
                                  torch.tensor(ob, device=dev), torch.tensor(offs, device=dev), R / 50, bound=R).cpu().numpy()
    want = [PN.collision_check_circle_edge(sv[i], ev[i], ob[offs[pr[i]]:offs[pr[i] + 1]], R / 50, bound=R) for i in range(n)]
    assert hit.tolist() == want


def test_extract_paths_rejects_non_square(dev):
    import torch
    from ppnet_amd import _lib, plan
    heat = torch.zeros(1, 64, 96, dtype=torch.uint8, device=dev)
    z = torch.zeros(1, 2, dtype=torch.float64, device=dev)
    with pytest.raises(_lib.PpnError):
        plan.extract_paths(heat, z, z, down_sample_rate=2)


def test_resize_matches_pillow(dev):
    import torch
    from PIL import Image
    from ppnet_amd import plan
    rng = np.random.RandomState(1)
    # rate 2 (the pipeline's), 3, 4 (9 taps: the kernel's per-output fallback), 8 (the reference's default), up-sampling, and a batch
    # longer than the 16 lines / images a thread walks with one set of weights
    for (h, w, oh, ow, n) in [(224, 224, 112, 112, 3), (256, 256, 128, 128, 3), (64, 96, 16, 48, 3), (50, 70, 25, 35, 3), (33, 47, 11, 15, 3),
                              (256, 256, 32, 32, 2), (24, 40, 48, 60, 2), (32, 32, 16, 16, 37)]:
        a = rng.randint(0, 256, size=(n, h, w)).astype(np.uint8)
        got = plan.resize_bilinear_u8(torch.tensor(a, device=dev), oh, ow).cpu().numpy()
        for i in range(n):
            want = np.asarray(Image.fromarray(a[i], mode="L").resize((ow, oh), Image.BILINEAR))
            assert np.array_equal(got[i], want)


def test_extract_paths_vs_reference_golden(dev, golden_dir):
    import torch
    from ppnet_amd import plan
    g = np.load(os.path.join(golden_dir, "g11_extract_path.npz"))
    for R in (224, 256):
        cases = [c for c in range(int(g["ncase"][0])) if g[f"c{c}_img"].shape[0] == R]
        heat = torch.tensor(np.stack([g[f"c{c}_img"] for c in cases]), device=dev)
        init = torch.tensor(np.stack([g[f"c{c}_init"] for c in cases]), device=dev)
        end = torch.tensor(np.stack([g[f"c{c}_end"] for c in cases]), device=dev)
        ok, full, cnt = plan.extract_paths(heat, init, end, down_sample_rate=2)
        for k, c in enumerate(cases):
            assert int(ok[k]) == int(g[f"c{c}_ok"][0])
            if int(ok[k]):
                want = g[f"c{c}_path"]
                assert int(cnt[k]) == len(want)                                          # same waypoint count
                assert np.abs(full[k, :len(want)].cpu().numpy() - want).max() < 1e-5     # fixture passed through float32


def test_extract_paths_vs_oracle_random_heat(dev):
    import torch
    from ppnet_amd import plan
    rng = np.random.RandomState(4)
    R, n = 128, 6
    yy, xx = np.mgrid[0:R, 0:R].astype(np.float64)
    heats, inits, ends = [], [], []
    for t in range(n):
        init = rng.random_sample(2) * 30 + 10
        end = rng.random_sample(2) * 30 + 85
        ts = np.linspace(0, 1, 300)[:, None]
        ctrl = rng.random_sample(2) * 90 + 20
        curve = (1 - ts) ** 2 * init + 2 * ts * (1 - ts) * ctrl + ts ** 2 * end
        d2 = np.min((yy[None] - curve[:, 0, None, None]) ** 2 + (xx[None] - curve[:, 1, None, None]) ** 2, axis=0)
        h = np.exp(-d2 / (2 * 3.0 ** 2)) + rng.random_sample((R, R)) * 0.02
        heats.append((np.clip(h, 0, 1) * 255).astype(np.uint8)); inits.append(init); ends.append(end)
    ok, full, cnt = plan.extract_paths(torch.tensor(np.stack(heats), device=dev), torch.tensor(np.stack(inits), device=dev),
                                       torch.tensor(np.stack(ends), device=dev), down_sample_rate=2)
    for t in range(n):
        o_ok, o_path = PN.extract_path(heats[t], inits[t], ends[t], down_sample_rate=2)
        assert bool(ok[t]) == bool(o_ok)
        if o_ok:
            assert int(cnt[t]) == len(o_path)
            assert np.abs(full[t, :len(o_path)].cpu().numpy() - o_path).max() < 1e-9     # waypoints are exact lattice sums


def test_batched_plan_collision_equals_per_segment_kernel(dev):
    """ppn_plan_collision (one launch per batch: segments from the waypoint array, each problem's own obstacle rows) against the
    per-segment kernel ppn_collision_segments_bound driven by the explicit segment / CSR lists — same flags, float32 and float64
    obstacle rows, ragged counts (0, 1, 2 and many points) and ragged obstacle counts."""
    import torch
    from ppnet_amd import plan
    g = torch.Generator().manual_seed(12)
    B, M, S, R = 37, 20, 9, 256
    wp = (torch.rand(B, M, 2, generator=g, dtype=torch.float64) * R).to(dev)
    counts = torch.randint(0, M + 1, (B,), generator=g, dtype=torch.int32).to(dev)
    counts[:4] = torch.tensor([0, 1, 2, M], dtype=torch.int32)
    obs64 = torch.cat([torch.rand(B, S, 2, generator=g, dtype=torch.float64) * R, torch.rand(B, S, 1, generator=g, dtype=torch.float64) * 12 + 1], dim=2).to(dev)
    n_obs = torch.randint(0, S + 1, (B,), generator=g, dtype=torch.int32).to(dev)
    n_obs[:2] = torch.tensor([0, S], dtype=torch.int32)
    clearance = R / 50
    # the explicit composition: every consecutive pair below the count, CSR of each problem's first n_obs rows
    s = wp[:, :-1].reshape(-1, 2).float(); e = wp[:, 1:].reshape(-1, 2).float()
    prob = torch.arange(B, device=dev, dtype=torch.int32).repeat_interleave(M - 1)
    valid = (torch.arange(M - 1, device=dev)[None, :] < (counts[:, None] - 1)).reshape(-1)
    rows = torch.cat([obs64[b, :int(n_obs[b])] for b in range(B)]).float()
    off = torch.zeros(B + 1, dtype=torch.int32, device=dev); off[1:] = torch.cumsum(n_obs, 0)
    if rows.numel() == 0:
        rows = torch.zeros(1, 3, device=dev)
    hit = plan.collision_segments(s, e, prob, rows, off, clearance, bound=R)
    want = (hit & valid).reshape(B, M - 1).any(dim=1)
    for obs in (obs64, obs64.float()):
        got = plan.plan_collision(wp, counts, obs, n_obs, clearance, bound=R)
        assert got.dtype == torch.bool and torch.equal(got, want)
    assert 0 < int(want.sum()) < B
