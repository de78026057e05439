"""TEST INFRASTRUCTURE — CPU oracle for the planner tail (loop B, rows B9/B10), not product code.

Restates EDaGe-PP/process_map.py: extract_path (:293-365) incl. Pillow's BILINEAR down-sampling,
and collision_check_circle_edge (:383-425) with the float widths of the reference's torch tensors.
Pinned by tests/test_oracle_plan_golden.py against tests/golden/g11_*.npz (captured from the
reference) and, for the resize, against Pillow itself.
"""
import math

import numpy as np

f32 = np.float32
MOTIONS = [[0, 1], [0, -1], [1, 0], [-1, 0], [1, 1], [1, -1], [-1, 1], [-1, -1]]   # process_map.py:294-297
MAX_WAYPOINTS = 2048


def collision_check_circle_edge(s, e, obs, clearance, bound=224):
    """process_map.py:383-425. s, e: float32 pairs; obs: rows (ox, oy, size). Quirks kept: bounds test on
    s[0], s[1] only against 0 / 224; the vertex test uses e twice and never s; `dir` flips persist.
    bound: the reference's constant 224 (its map size); the resolution for maps of another size."""
    s = np.asarray(s, dtype=f32)
    e = np.asarray(e, dtype=f32)
    if s[0] < 0 or s[1] > bound:
        return True
    if e[0] < 0 or e[1] > bound:
        return True
    sx, sy, ex, ey = s[1], s[0], e[1], e[0]
    dx, dy = f32(ex - sx), f32(ey - sy)
    nrm = f32(np.sqrt(f32(f32(dx * dx) + f32(dy * dy))))
    dirx, diry = f32(dy / nrm), f32(f32(-dx) / nrm)
    lim_add = clearance / 2
    for ox, oy, size in obs:
        ox32, oy32 = f32(ox), f32(oy)
        lim = float(size) + lim_add
        ddx, ddy = f32(ex - ox32), f32(ey - oy32)                     # scipy euclidean keeps float32
        if float(np.sqrt(f32(f32(ddx * ddx) + f32(ddy * ddy)))) < lim:
            return True
        qx, qy = f32(ox32 - sx), f32(oy32 - sy)
        dis = f32(f32(dirx * qx) + f32(diry * qy))
        if dis > 0:
            dirx, diry = f32(-dirx), f32(-diry)
        dis = f32(abs(dis))
        px, py = f32(ox32 + f32(dis * dirx)), f32(oy32 + f32(dis * diry))
        ax, ay = f32(px - sx), f32(py - sy)
        an = f32(np.sqrt(f32(f32(ax * ax) + f32(ay * ay))))
        ax, ay = f32(ax / an), f32(ay / an)
        bx, by = f32(px - ex), f32(py - ey)
        bn = f32(np.sqrt(f32(f32(bx * bx) + f32(by * by))))
        bx, by = f32(bx / bn), f32(by / bn)
        if float(dis) < lim and f32(f32(ax * bx) + f32(ay * by)) < 0:
            return True
    return False


def _resize_pass(img, out_size, axis):
    """One pass of Pillow's ImagingResample (8 bpc, bilinear filter, PRECISION_BITS = 22)."""
    in_size = img.shape[axis]
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ss = 1.0 / filterscale
    src = np.moveaxis(img, axis, -1).astype(np.int64)
    out = np.zeros(src.shape[:-1] + (out_size,), dtype=np.uint8)
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.array([max(0.0, 1.0 - abs((x + xmin - center + 0.5) * ss)) for x in range(xmax)])
        ww = w.sum() if xmax else 0.0                                  # Pillow adds sequentially; <= 5 terms
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = w / ww
        kq = np.array([int(-0.5 + v * 4194304.0) if v < 0 else int(0.5 + v * 4194304.0) for v in w], dtype=np.int64)
        acc = (1 << 21) + (src[..., xmin:xmin + xmax] * kq).sum(-1)
        out[..., xx] = np.clip(acc >> 22, 0, 255).astype(np.uint8)
    return np.moveaxis(out, -1, axis)


def resize_bilinear_u8(img, out_h, out_w):
    """PIL.Image.resize((out_w, out_h), Image.BILINEAR) for mode 'L': horizontal pass, then vertical."""
    return _resize_pass(_resize_pass(np.asarray(img, dtype=np.uint8), out_w, 1), out_h, 0)


def extract_path(mask_u8, init_state, end_state, down_sample_rate=8, max_wp=MAX_WAYPOINTS):
    """process_map.py:293-365 with the 1 s wall-clock timeout replaced by the max_wp step cap.
    mask_u8: [H,W] uint8. Returns (ok, waypoints float64 [n+2,2] | None)."""
    H, W = mask_u8.shape
    init = np.asarray(init_state, dtype=np.float64) / down_sample_rate
    end = np.asarray(end_state, dtype=np.float64) / down_sample_rate
    w2, h2 = int(W / down_sample_rate), int(H / down_sample_rate)       # PIL size = (width, height)
    mask = resize_bilinear_u8(mask_u8, h2, w2).astype(np.float32) / np.float32(255.0)
    path = []
    nxt = init
    while len(path) < max_wp:
        cand = [np.array(m, dtype=np.float64) + nxt for m in MOTIONS]
        vals = []
        for c in cand:
            r0, c0 = int(np.round(c[0])), int(np.round(c[1]))
            vals.append(float(mask[r0, c0]) if (0 <= r0 < w2 and 0 <= c0 < h2) else 0.0)    # :318 (size[0] bounds c[0])
        chosen = None
        while max(vals) > 0:
            ci = vals.index(max(vals))
            nxt = cand[ci]
            ok = True
            for i, p in enumerate(path):
                d = math.sqrt((nxt[0] - p[0]) ** 2 + (nxt[1] - p[1]) ** 2)
                if (nxt == p).all() or (d <= 1.5 and i < len(path) - 2):
                    vals[ci] = 0
                    ok = False
                    break
            if ok:
                chosen = ci
                break
        if chosen is None:
            return False, None
        path.append(nxt)
        if math.sqrt((nxt[0] - end[0]) ** 2 + (nxt[1] - end[1]) ** 2) <= 2.5:
            pts = [np.asarray(init_state, dtype=np.float64)] + [p * down_sample_rate for p in path] + \
                  [np.asarray(end_state, dtype=np.float64)]
            return True, np.array(pts)
    return False, None
