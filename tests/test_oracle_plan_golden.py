"""Pins oracle/plan_np.py against golden vectors captured from the reference's process_map.py
(tests/golden/make_fixtures.py) and its Pillow-resize restatement against Pillow itself.  CPU only."""
import os

import numpy as np
import pytest

from oracle import plan_np as PN


def test_collision_check_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "g11_collision.npz"))
    off = np.concatenate([[0], np.cumsum(g["n_obs"])])
    got = [int(PN.collision_check_circle_edge(g["s"][i], g["e"][i], g["obs"][off[i]:off[i + 1]], float(g["clearance"][0])))
           for i in range(len(g["hit"]))]
    assert got == g["hit"].tolist()
    assert 0 < sum(got) < len(got)


def test_resize_matches_pillow():
    from PIL import Image
    rng = np.random.RandomState(0)
    for (h, w, oh, ow) in [(224, 224, 112, 112), (256, 256, 128, 128), (64, 96, 16, 48), (50, 70, 25, 35), (33, 47, 11, 15)]:
        a = rng.randint(0, 256, size=(h, w)).astype(np.uint8)
        want = np.asarray(Image.fromarray(a, mode="L").resize((ow, oh), Image.BILINEAR))
        assert np.array_equal(PN.resize_bilinear_u8(a, oh, ow), want)


def test_extract_path_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "g11_extract_path.npz"))
    n_ok = 0
    for c in range(int(g["ncase"][0])):
        ok, path = PN.extract_path(g[f"c{c}_img"], g[f"c{c}_init"], g[f"c{c}_end"], down_sample_rate=2)
        assert int(ok) == int(g[f"c{c}_ok"][0]), c
        if ok:
            n_ok += 1
            assert path.shape == g[f"c{c}_path"].shape
            assert np.abs(path - g[f"c{c}_path"]).max() < 1e-5      # the fixture went through float32 tensors
    assert n_ok >= 8
