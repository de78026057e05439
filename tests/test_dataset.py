"""Dataset layout and problem / solution records (reference process_map.py:30-72,148-191,
236-274, MapGenerate.py:144-149, updated_geometric_planner.py:500-569)."""
import json
import os
import types

import numpy as np
import pytest


def test_voc_colormap_known_entries():
    from ppnet_amd.dataset import voc_colormap
    c = voc_colormap()
    assert c.shape == (256, 3) and c.dtype == np.uint8
    # PASCAL VOC palette: background, aeroplane, bicycle, bird, ..., person (15), tv/monitor (20), void (255)
    assert c[0].tolist() == [0, 0, 0] and c[1].tolist() == [128, 0, 0] and c[2].tolist() == [0, 128, 0]
    assert c[3].tolist() == [128, 128, 0] and c[15].tolist() == [192, 128, 128] and c[20].tolist() == [0, 64, 128]
    assert c[255].tolist() == [224, 224, 192]


@pytest.fixture(scope="module")
def dev():
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return torch.device("cuda:0")


def _fake_maps(n=4, K=3):
    rs = np.random.RandomState(0)
    return types.SimpleNamespace(segpoint=rs.rand(n, 11, 2) * 64, obstacles=rs.rand(n, K + 64, 3) * 64,
                                 n_obstacles=np.stack([np.array([2, 0, 3, 1]), np.array([1, 0, 2, 1])], axis=1), n=n)


def test_problem_and_solution_records_follow_the_reference_schema(tmp_path):
    from ppnet_amd import dataset
    maps = _fake_maps()
    probs = dataset.problem_records(maps, np.array([10.0, 20.0]), placements=2, first_index=100)
    assert [p["Index"] for p in probs] == [100, 101, 102, 103]
    assert [p["Length"] for p in probs] == [10.0, 10.0, 20.0, 20.0]
    assert [len(p["Obstacles"]) for p in probs] == [2, 0, 3, 1]
    assert probs[2]["Init"] == maps.segpoint[2, 0].tolist() and probs[2]["End"] == maps.segpoint[2, 10].tolist()
    assert set(probs[0]) == {"Index", "Init", "End", "Length", "Obstacles"}          # MapGenerate.py:144-146
    wp = np.zeros((4, 6, 2)); wp[0, :3] = [[0, 0], [3, 4], [3, 10]]
    sols = dataset.solution_records(probs, np.array([True, False, False, False]), wp, np.array([3, 0, 0, 0]), 0.002)
    s0 = sols[0]["Solution"][0]
    assert s0["Planner"] == "PPNet" and s0["Waypoint"] == [[0.0, 0.0], [4.0, 3.0], [10.0, 3.0]]    # [x, y] = [col, row]
    assert abs(s0["Length"] - 11.0) < 1e-12 and s0["Time"] == 0.002
    assert sols[1]["Solution"][0]["Waypoint"] is None
    f = tmp_path / "solved_problems_comparison.txt"
    dataset.append_json_lines(str(f), sols)
    back = [json.loads(l) for l in open(f)]
    assert back == sols and set(back[0]["Solution"][0]) == {"Planner", "Waypoint", "Length", "Time"}   # harness :563


@pytest.mark.gpu
def test_write_dataset_layout_matches_device_batch(dev, tmp_path):
    import torch
    from PIL import Image
    from ppnet_amd import dataset, edage
    P, placements, R, K = 3, 4, 64, 10
    pb = edage.generate_paths(P, R, 50, 3, seed=3, device=dev)
    mb = edage.generate_maps(pb, placements, 5, K, seed=3)
    root = str(tmp_path / "ds")
    stems = dataset.write_dataset(root, pb, mb, placements, first_index=200)
    assert stems == [str(200 + i) for i in range(P * placements)]
    mp, ms = edage.label_masks(pb, mb, placements)
    grid = mb.grid.cpu().numpy()
    for i in range(P * placements):
        assert np.array_equal(np.asarray(Image.open(f"{root}/mask_path/{200 + i}.png")), mp[i].cpu().numpy())
        sp = Image.open(f"{root}/mask_space/{200 + i}.png")
        assert sp.mode == "P" and np.array_equal(np.asarray(sp), ms[i].cpu().numpy())
        assert sp.getpalette()[:6] == [0, 0, 0, 128, 0, 0]
        im = np.asarray(Image.open(f"{root}/map/{200 + i}.jpg")).astype(int)
        assert im.shape == (R, R, 3)
        # JPEG is lossy: free cells stay bright, obstacle cells dark (away from edges the error is a few levels)
        free, occ = grid[i] == 255, grid[i] == 0
        assert im[free].mean() > 235 and (occ.sum() == 0 or im[occ].mean() < 40)
    lines = [json.loads(l) for l in open(f"{root}/unsolved_problems.txt")]
    assert len(lines) == P * placements and lines[5]["Index"] == 205
    assert abs(lines[5]["Length"] - float(pb.length[1])) < 1e-12
    assert len(lines[5]["Obstacles"]) == int(mb.n_obstacles[5, 0])
    assert open(f"{root}/ImageSets/Segmentation/test.txt").read().split() == sorted(stems)
    assert len(open(f"{root}/init_end.txt").readlines()) == P * placements
    assert torch.is_tensor(mb.grid)
