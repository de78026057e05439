"""On-disk dataset layout and problem / solution records of the reference, written from GPU batches.

Reference: EDaGe-PP/process_map.py:30-72 (process_map), :148-191 (generate_gen_path / generate_seg_space),
:236-274 (record_init_end, generate_txt); EDaGe-PP/MapGenerate.py:144-149 (unsolved_problems.txt lines);
experiments/ompl_experiments/updated_geometric_planner.py:500-569 (solved_problems_comparison.txt lines).

The reference builds the layout in three passes over image files (MapGenerate writes folders of 100 JPEGs +
MapLabel, process_map re-reads them, rotates every corridor image again and writes the masks).  Here the masks come
from the same device batch that produced the maps (ppn_label_masks), so one call writes

    root/map/{i}.jpg                     occupancy image, start / goal painted red
    root/mask_path/{i}.png               'L', 255 on every 5th label point          (generate_gen_path)
    root/mask_space/{i}.png              'P' + VOC palette, 1 inside the corridor   (generate_seg_space)
    root/init_end.txt                    one dict per image                          (record_init_end)
    root/ImageSets/Segmentation/*.txt    test (+ train / val above 50000 images)     (generate_txt)
    root/unsolved_problems.txt           JSON lines {"Index","Init","End","Length","Obstacles"}

File encoding (JPEG / PNG via Pillow) is host work and not part of the measured path.
"""
import json
import os

import numpy as np
import torch
from PIL import Image

from . import edage

SUPPORTED_MASKS = ("Gen_path", "Seg_space", "All")


def voc_colormap(n=256):
    """imgviz.label_colormap(): the PASCAL VOC palette (bit-interleaved label index), u8 [n,3]."""
    cmap = np.zeros((n, 3), dtype=np.uint8)
    for i in range(n):
        c, r, g, b = i, 0, 0, 0
        for j in range(8):
            r |= ((c >> 0) & 1) << (7 - j)
            g |= ((c >> 1) & 1) << (7 - j)
            b |= ((c >> 2) & 1) << (7 - j)
            c >>= 3
        cmap[i] = (r, g, b)
    return cmap


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def problem_records(maps, path_lengths, placements, first_index=0, limit=None):
    """The dicts MapGenerate.generate_map_randomly appends to unsolved_problems.txt (MapGenerate.py:144-149):
    Init / End are segpoint[0] / segpoint[10] as (row, col), Obstacles rows are [col, row, r] (kept + pocket)."""
    seg = _np(maps.segpoint)
    obs = _np(maps.obstacles)
    n_obs = _np(maps.n_obstacles)[:, 0]
    lengths = _np(path_lengths)
    n = seg.shape[0] if limit is None else min(seg.shape[0], limit)
    out = []
    for m in range(n):
        out.append({"Index": int(first_index + m), "Init": [float(v) for v in seg[m, 0]], "End": [float(v) for v in seg[m, 10]],
                    "Length": float(lengths[m // placements]),
                    "Obstacles": [[float(v) for v in o] for o in obs[m, :n_obs[m]]]})
    return out


def solution_records(problems, ok, waypoints, counts, seconds_per_problem, planner="PPNet"):
    """Adds the harness's "Solution" list to problem dicts (updated_geometric_planner.py:553-566): one entry per
    planner with the waypoints as [x, y] = [col, row] (the harness swaps Init / End the same way, :520,531), the
    polyline length and the planning time; a failed extraction is Waypoint None, as for an OMPL planner without a path."""
    okh, wph, cnth = _np(ok), _np(waypoints), _np(counts)
    out = []
    for i, p in enumerate(problems):
        q = dict(p)
        if okh[i]:
            w = wph[i, :cnth[i]]
            length = float(np.sqrt(((w[1:] - w[:-1]) ** 2).sum(axis=1)).sum())
            sol = {"Planner": planner, "Waypoint": [[float(c), float(r)] for r, c in w], "Length": length,
                   "Time": float(seconds_per_problem)}
        else:
            sol = {"Planner": planner, "Waypoint": None, "Length": None, "Time": float(seconds_per_problem)}
        q["Solution"] = list(p.get("Solution", [])) + [sol]
        out.append(q)
    return out


def append_json_lines(path, records):
    with open(path, "a") as f:
        for r in records:
            f.write(json.dumps(r) + "\n")


def write_split_lists(root, n_images_total=None):
    """generate_txt (process_map.py:251-274): sorted image stems of root/map into ImageSets/Segmentation."""
    txt_root = os.path.join(root, "ImageSets", "Segmentation")
    os.makedirs(txt_root, exist_ok=True)
    stems = sorted(os.path.splitext(f)[0] for f in os.listdir(os.path.join(root, "map"))
                   if os.path.splitext(f)[-1] in (".jpg", ".JPG", ".png", ".PNG"))
    with open(os.path.join(txt_root, "test.txt"), "w") as f:
        f.writelines(s + "\n" for s in stems)
    if len(stems) > 50000:
        with open(os.path.join(txt_root, "train.txt"), "w") as f:
            f.writelines(s + "\n" for s in stems)
        with open(os.path.join(txt_root, "val.txt"), "w") as f:
            f.writelines(s + "\n" for s in stems[-5000:])
    return stems


def write_dataset(root, paths, maps, placements, first_index=0, mode="All", record_problems=True, label_bound=None):
    """Writes one device batch (PathsBatch + MapsBatch of paths.n * placements maps) in the layout above; image i of
    the batch gets the number first_index + i (the reference's folder_index * NUM_PER_FOLDER + i).  Returns the stems."""
    assert mode in SUPPORTED_MASKS
    for d in ("map", "mask_path", "mask_space"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    want_path, want_space = mode in ("Gen_path", "All"), mode in ("Seg_space", "All")
    mask_path, mask_space = edage.label_masks(paths, maps, placements, bound=label_bound, want_path=want_path, want_space=want_space)
    rgb = (edage.grid_to_rgb(maps.grid) * 255.0).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
    rgb_h = rgb.cpu().numpy()
    mp_h = mask_path.cpu().numpy() if want_path else None
    ms_h = mask_space.cpu().numpy() if want_space else None
    seg = _np(maps.segpoint)
    palette = voc_colormap().flatten().tolist()
    stems = []
    with open(os.path.join(root, "init_end.txt"), "a") as ie:
        for m in range(maps.n):
            idx = first_index + m
            Image.fromarray(rgb_h[m], mode="RGB").save(os.path.join(root, "map", f"{idx}.jpg"))
            if want_path:
                Image.fromarray(mp_h[m], mode="L").save(os.path.join(root, "mask_path", f"{idx}.png"))
            if want_space:
                im = Image.fromarray(ms_h[m], mode="P")
                im.putpalette(palette)
                im.save(os.path.join(root, "mask_space", f"{idx}.png"))
            # record_init_end (process_map.py:236-248): str() of a dict, coordinates as str(array)
            ie.write(str({"image": f"{root}/{idx}.jpg", "init": str(seg[m, 0]), "end": str(seg[m, 10])}) + "\n")
            stems.append(str(idx))
    if record_problems:
        append_json_lines(os.path.join(root, "unsolved_problems.txt"),
                          problem_records(maps, paths.length, placements, first_index))
    write_split_lists(root)
    return stems
