"""The drop-in modules (ppnet_amd/dropin) driven the way the reference's own __main__ drives them
(EDaGe-PP/MapGenerate.py:154-176), in both random-stream modes."""
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "ppnet_amd", "dropin")


@pytest.fixture()
def dropin(monkeypatch, tmp_path):
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available()
    monkeypatch.syspath_prepend(DROPIN)
    for m in ("PathSeg", "Path", "PathGenerate", "MapGenerate", "process_map"):
        sys.modules.pop(m, None)
    monkeypatch.chdir(tmp_path)
    yield tmp_path
    for m in ("PathSeg", "Path", "PathGenerate", "MapGenerate", "process_map"):
        sys.modules.pop(m, None)


def test_config1_mt19937_replay_matches_reference(dropin, golden_dir):
    """BASELINE config 1 through the reference's own call sequence: np.random.seed(0); torch.manual_seed(0);
    MapGenerate(10, 64, 50, 5, 20, 3).generate(100).  Labels, accepted placements, the pocket obstacles of every target path,
    the FULL obstacle list of all 100 problems and the positions of both global streams (numpy's and torch's) match the reference:
    in replay mode Path computes Qhull's first hull vertex with scipy and the kernel walks the isles in that order.
    One documented exception ("Ties"): where the reference's arg-max over lattice points (Path.py:475) is a
    tie decided by BLAS rounding noise, the kernel — like the oracle in its plain-arithmetic mode — takes the first index; such an
    obstacle sits at another lattice point of the same chord (same radius, same number of draws).  The test allows a row to
    differ from the golden only where the plain-arithmetic oracle differs from it too, and then requires the oracle's value."""
    import torch
    from oracle import edage_np as E
    from ppnet_amd import rng
    rng.set_mode("mt19937")
    import MapGenerate as MG
    g = np.load(os.path.join(golden_dir, "g10_config1_R64.npz"))
    np.random.seed(0)
    torch.manual_seed(0)
    plain = E.generate_paths(E.MTSource(), 10, 64, 50, 3, hull_order="scipy")      # plain arithmetic: ties -> first index
    tie_rows = {}                                                                    # path -> rows of its obstacle list that are ties
    for j, p in enumerate(plain):
        ref_o = g[f"p{j}/obstacles"].reshape(-1, 3)
        assert len(p["obstacles"]) == len(ref_o)
        bad = np.where(np.abs(p["obstacles"].reshape(-1, 3) - ref_o).max(axis=1, initial=0) > 1e-6)[0] if len(ref_o) else []
        if len(bad):
            tie_rows[j] = set(int(b) for b in bad)
    assert sum(len(v) for v in tie_rows.values()) <= 2                              # one or two ties in 110 paths' worth of isles
    np.random.seed(0)
    torch.manual_seed(0)
    MG.cnt = 0
    mg = MG.MapGenerate(path_num=10, resolution=64, map_size=50, obstacles_size=5, obstacles_num=20, clearance=3)
    for j, tp in enumerate(mg.PathGroup.TargetPaths):
        assert abs(tp.Length - g[f"p{j}/length"][0]) < 1e-7
        assert np.abs(np.asarray(tp.PathPoint) - g[f"p{j}/pathpoint_image"]).max() < 1e-7
        assert np.abs(tp.SegPointImage - g[f"p{j}/segpoint_image"]).max() < 1e-7
        assert abs(tp.Rotation - g[f"p{j}/rotation"][0]) < 1e-9
        # pocket obstacles: torch.rand consumed isle by isle in Qhull's vertex order (ppn_edage_paths_ex2's hull_start) -> the
        # reference's values (they pass through float32: 1e-4 px)
        assert np.abs(np.asarray(tp.ConvexHull) - g[f"p{j}/hull_norm"]).max() < 1e-6
        ref_o = g[f"p{j}/obstacles"].reshape(-1, 3).copy()
        assert len(tp.obstacles) == len(ref_o)
        for row in tie_rows.get(j, ()):
            ref_o[row] = plain[j]["obstacles"].reshape(-1, 3)[row]
        assert np.abs(np.array(tp.obstacles).reshape(-1, 3) - ref_o).max(initial=0) < 1e-4
    mg.generate(map_num=100, folder_path=str(dropin / "out"), round_index=0)
    assert len(mg.MapLabel) == 100
    assert np.abs(np.array([np.ravel(l[1])[0] for l in mg.MapLabel]) - g["angle"]).max() < 1e-12
    assert np.array_equal(np.array([l[2] for l in mg.MapLabel]), g["translation"])
    assert np.abs(np.array([l[3] for l in mg.MapLabel]) - g["segpoint"]).max() < 1e-7
    assert np.abs(np.array([l[4] for l in mg.MapLabel]) - g["pathpoint"]).max() < 1e-7
    assert np.array_equal(np.random.random(4), g["np_next_draws"])           # numpy stream fully in step
    problems = [json.loads(l) for l in open("unsolved_problems.txt")]
    assert [p["Index"] for p in problems] == g["problem_index"].tolist()
    assert np.abs(np.array([p["Length"] for p in problems]) - g["problem_length"]).max() < 1e-7
    assert np.array_equal(np.array([torch.rand(1).item() for _ in range(4)], np.float32), g["torch_next_draws"])   # and torch's
    # the full obstacle list of every problem: kept random obstacles (same draws, same filter: 1e-7), then the placed pocket ones
    n_ref = g["n_obs"]
    off = np.concatenate([[0], np.cumsum(n_ref)])
    for m, p in enumerate(problems):
        j = m // 10
        n_pocket = len(g[f"p{j}/obstacles"])
        ref = g["obstacles"][off[m]:off[m + 1]]
        got = np.array(p["Obstacles"]).reshape(-1, 3)
        n_rand = len(ref) - n_pocket
        assert len(got) == len(ref), (m, len(got), len(ref))
        assert np.abs(got[:n_rand] - ref[:n_rand]).max(initial=0) < 1e-7
        ok_rows = [r for r in range(n_pocket) if r not in tie_rows.get(j, ())]
        assert np.abs(got[n_rand:][ok_rows] - ref[n_rand:][ok_rows]).max(initial=0) < 1e-4
        assert np.abs(got[n_rand:, 2] - ref[n_rand:, 2]).max(initial=0) < 1e-4          # a tie moves the centre, never the radius
    assert os.path.exists(dropin / "out" / "0.jpg") and os.path.exists(dropin / "out" / "99.jpg")
    assert os.path.exists(dropin / "out" / "data" / "9.jpg")


def test_philox_mode_runs_reference_main_sequence(dropin):
    import torch
    from ppnet_amd import rng, _lib
    rng.set_mode("philox", seed=5)
    import MapGenerate as MG
    MG.cnt = 0
    mg = MG.MapGenerate(path_num=6, resolution=128, map_size=50, obstacles_num=20, clearance=3)
    mg.generate(map_num=36, folder_path=str(dropin / "r0"), round_index=0, save_images=False)
    assert len(mg.MapLabel) == 36
    grid = mg.last_batch.grid
    assert set(torch.unique(grid).tolist()) <= {0, 128, 255}
    tp = mg.PathGroup.TargetPaths[0]
    ok, hull = tp.boundary_check(0.0, [0, 0])
    assert ok and hull.shape[1] == 2                                      # normalised hull sits inside the image
    assert tp.Space.shape == (3, 128, 128)
    # same seed -> same maps; stream ids advanced -> a second round differs
    rng.set_mode("philox", seed=5)
    MG.cnt = 0
    mg2 = MG.MapGenerate(path_num=6, resolution=128, map_size=50, obstacles_num=20, clearance=3)
    mg2.generate(map_num=36, folder_path=str(dropin / "r1"), round_index=0, save_images=False)
    assert torch.equal(mg2.last_batch.grid, grid)
    rng.set_mode("mt19937")


def test_process_map_functions(dropin, golden_dir):
    import torch
    from ppnet_amd import rng
    rng.set_mode("mt19937")
    import process_map as PM
    g = np.load(os.path.join(golden_dir, "g12_init_end.npz"))
    out = PM.add_init_end_single(torch.zeros([3, 64, 64]), g["init"], g["end"])
    assert np.array_equal(out.numpy(), g["out"])
    c = np.load(os.path.join(golden_dir, "g11_collision.npz"))
    off = np.concatenate([[0], np.cumsum(c["n_obs"])])
    for i in range(0, 40):
        hit = PM.collision_check_circle_edge(torch.tensor(c["s"][i]), torch.tensor(c["e"][i]),
                                             [list(o) for o in c["obs"][off[i]:off[i + 1]]], float(c["clearance"][0]))
        assert int(hit) == int(c["hit"][i])
    from PIL import Image
    e = np.load(os.path.join(golden_dir, "g11_extract_path.npz"))
    ok, path = PM.extract_path(Image.fromarray(e["c0_img"], mode="L"), e["c0_init"], e["c0_end"], down_sample_rate=2)
    assert ok == bool(e["c0_ok"][0])
    if ok:
        assert np.abs(path.numpy() - e["c0_path"]).max() < 1e-5
