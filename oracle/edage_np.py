"""TEST INFRASTRUCTURE — CPU oracle for loop A (EDaGe-PP), not product code.

A NumPy restatement of the reference's map+path generator, function by function, each
citing the reference file:line it follows (paths relative to /root/reference/EDaGe-PP).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Pinned: tests/test_oracle_golden.py checks every function here against tests/golden/*.npz,
which were captured from the reference's own modules (tests/golden/make_fixtures.py).

Pinned to the primitive: torchvision rotate/affine of the corridor mask (Path.py:160-161,175; MapGenerate.py:103-106) —
torchvision is absent, but its tensor path is a published composition of torch primitives that ARE importable;
`rotate_nearest` / `translate_nearest` restate it in float32 and are bit-equal to that composition (see "rasters").

NOT pinned ("parity unpinned"): the raster the reference delegates to
third-party code that is not integer-reproducible —
  * matplotlib->JPEG->PIL('1')->crop->Resize obstacle raster (Path.py:36-49)
    -> restated as "the data point the pixel centre shows lies inside the inked ellipse" (`disc_raster`, `raster_geometry`).

Conventions: a point is (row, col) = (world x, world y) as in the reference (Path.py:401
`space[index[0], index[1]]`). Obstacles are [col, row, radius] (Path.py:495, MapGenerate.py:143).
"""
import math

import numpy as np

from . import philox_np as px

PATHSEGNUM = 10          # MapGenerate.py:22
ORDER = 4                # MapGenerate.py:21
SEG_LEN_RANGE = 7        # PathSeg.py:5
MIN_LEN = 0              # PathSeg.py:6
N_FIT = 1000             # PathSeg.py:22-23
DRAWS_PER_SEG = 1 + N_FIT + 1           # straight flag, 1000 samples, end abscissa
DRAWS_PER_PATH = 1 + PATHSEGNUM * DRAWS_PER_SEG   # + path-level straight flag (PathGenerate.py:36)

GRID_OBST = 0
GRID_FREE = 255
GRID_MARK = 128

FLAG_POCKET_CAP = 1      # set_obstacles try cap hit (the reference would loop forever, Path.py:478)
FLAG_PLACE_CAP = 2       # placement retry cap hit (MapGenerate.py:60-62)
FLAG_EMPTY_ISLE = 4      # search_isle produced an empty slice (reference raises IndexError, Path.py:529)

POCKET_TRY_CAP = 256     # per-isle cap on set_obstacles iterations
PLACE_TRY_CAP = 4096     # per-map cap on placement attempts (reference: 1e6 per target path)

# Arithmetic / tie rule.  The reference calls np.dot on 2-vectors (Path.py:274,473,532); BLAS
# may fuse that into fma(a1*b1 + rn(a0*b0)) or not depending on the host CPU kernel OpenBLAS
# picks, and two reductions in the reference are decided by that rounding noise whenever
# lattice points tie exactly (argmax at Path.py:475 — points displaced along the integer chord
# direction are mathematically equidistant; `dis > int` at Path.py:533 for axis-aligned chords).
#   "blas"  : literal np.dot + exact comparisons  -> reproduces the golden fixtures captured
#             on this container's OpenBLAS, tie noise included (used only for pinning).
#   "plain" : rn(a0*b0) + rn(a1*b1), no fusion, and ties resolved by TIE_EPS (first index
#             within TIE_EPS of the maximum; "exceeds" means by more than TIE_EPS) -> platform
#             independent; this is the rule the HIP kernels implement.
ARITH = "plain"
TIE_EPS = 1e-9


class arith:
    """Context manager: with arith("blas"): ..."""

    def __init__(self, mode):
        assert mode in ("plain", "blas")
        self.mode = mode

    def __enter__(self):
        global ARITH
        self.old, ARITH = ARITH, self.mode

    def __exit__(self, *a):
        global ARITH
        ARITH = self.old


def dot2(dx, dy, n):
    """|row-wise dot| helper: (dx, dy) . n for arrays dx, dy and a 2-vector n."""
    if ARITH == "blas":
        return np.array([np.dot(np.array([a, b]), n) for a, b in zip(np.atleast_1d(dx), np.atleast_1d(dy))])
    return dx * n[0] + dy * n[1]


# --------------------------------------------------------------------------- draw sources
class PhiloxSource:
    """Throughput-mode draws: every draw keyed by (seed, stream, instance, index)."""

    def __init__(self, seed):
        self.seed = seed

    def path_draws(self, path_id):
        return px.doubles(self.seed, px.STREAM_PATH, path_id, 0, DRAWS_PER_PATH)

    def pocket_floats(self, path_id, n):
        return px.floats(self.seed, px.STREAM_POCKET, path_id, 0, n)

    def place_draws(self, map_id, attempt):
        return px.doubles(self.seed, px.STREAM_PLACE, map_id, 3 * attempt, 3)

    def obst_draws(self, map_id, K):
        return px.doubles(self.seed, px.STREAM_OBST, map_id, 0, 3 * K)


class MTSource:
    """Reference-order draws from the *global* numpy / torch generators, consumed exactly as
    the reference consumes them (so the caller must invoke stages in the reference's order)."""

    def path_draws(self, path_id):
        d = np.ones(DRAWS_PER_PATH)
        d[0] = np.random.random(1)[0]                    # PathGenerate.py:36
        path_straight = not (d[0] > 0.01)
        for s in range(PATHSEGNUM):
            b = 1 + s * DRAWS_PER_SEG
            if not path_straight:                        # PathSeg.py:19 short-circuit
                d[b] = np.random.random(1)[0]
            d[b + 1:b + 1 + N_FIT] = np.random.random(N_FIT)   # PathSeg.py:23
            d[b + 1 + N_FIT] = np.random.random(1)[0]    # PathSeg.py:32
        return d

    class _TorchStream:
        def __getitem__(self, i):
            raise TypeError

    def pocket_floats(self, path_id, n):
        return None   # sentinel: set_obstacles pulls torch.rand(1) lazily

    def place_draws(self, map_id, attempt):
        a = np.random.random([1])                        # MapGenerate.py:63
        t = np.random.random([2])                        # MapGenerate.py:64
        return np.array([a[0], t[0], t[1]])

    def obst_draws(self, map_id, K):
        return np.concatenate([np.random.random(K), np.random.random(K), np.random.random(K)])  # MapGenerate.py:128-130

    def map_rotation_draw(self):
        """MapGenerate.py:103-104: torchvision's RandomRotation draws its angle from the global torch generator
        (`torch.empty(1).uniform_(a, a)`: one draw) for every placed map."""
        import torch
        torch.rand(1)


class _FloatFeed:
    """Sequential float32 feed: from an array, or lazily from torch.rand(1) (MT mode)."""

    def __init__(self, arr, rotation_draws=0):
        """rotation_draws: how many leading entries of `arr` belong to space_normalization's RandomRotation
        (a fixture that recorded the torch stream from the start of path_obstacles)."""
        self.arr = arr
        self.i = 0
        self.rotation_draws = rotation_draws

    def rotation_draw(self):
        """Path.py:160-161: torchvision 0.12's RandomRotation draws its angle (`torch.empty(1).uniform_(a, a)`) from the
        global torch generator — one draw per path, before set_obstacles' torch.rand.  Keyed (Philox) feeds have no
        global stream to keep in step."""
        if self.arr is None:
            import torch
            torch.rand(1)
        else:
            self.i += self.rotation_draws

    def next(self):
        if self.arr is None:
            import torch
            v = np.float32(torch.rand(1).item())
        else:
            v = np.float32(self.arr[self.i])
        self.i += 1
        return v


# --------------------------------------------------------------------------- small helpers
def coord_rotation(x, radians):
    """Path.py:271-274."""
    c, s = np.cos(radians), np.sin(radians)
    if ARITH == "blas":
        rot = np.reshape([[c, -s], [s, c]], [2, 2])
        return np.dot(rot, x)
    c, s = float(np.ravel(c)[0]), float(np.ravel(s)[0])
    x = np.asarray(x, dtype=np.float64)
    return np.stack([c * x[0] - s * x[1], s * x[0] + c * x[1]], axis=0)


def polyval4(p, x):
    """np.polyval for a degree-4 coefficient vector (Horner, highest power first)."""
    y = np.zeros_like(np.asarray(x, dtype=np.float64))
    for c in p:
        y = y * x + c
    return y


def euclid(a, b):
    """scipy.spatial.distance.euclidean on 2-vectors == sqrt(dx*dx + dy*dy)."""
    dx = a[..., 0] - b[..., 0]
    dy = a[..., 1] - b[..., 1]
    return np.sqrt(dx * dx + dy * dy)


def euclidean2image(x, step_len, mapoffset):
    """Path.py:378-386: int(np.round(p/step_len + mapoffset)) per coordinate (half-to-even)."""
    x = np.reshape(x, [-1, 2])
    return np.round(x / step_len + mapoffset).astype(np.int64)


# --------------------------------------------------------------------------- A1  PathSeg
def pathseg_random(draws, path_straight):
    """PathSeg.__init__ + random (PathSeg.py:10-36,38-58).

    draws: float64[1002] = [straight flag draw, 1000 sample draws, end draw] (flag slot is
    ignored when path_straight, mirroring the short-circuit at PathSeg.py:19)."""
    is_straight = True if path_straight or draws[0] < 0.2 else False
    x = np.arange(0, N_FIT) / 100
    y = draws[1:1 + N_FIT] * 10 - 5
    poly = np.polyfit(x, y, ORDER)
    poly[ORDER] = 0
    if is_straight:
        poly[0:ORDER - 1] = 0
    endpoint = draws[1 + N_FIT] * (SEG_LEN_RANGE - MIN_LEN) + MIN_LEN
    y_end = polyval4(poly, endpoint)
    translation = np.array([endpoint, y_end])
    p_d = np.polyder(poly)
    grad_st = np.polyval(p_d, 0)
    grad_end = np.polyval(p_d, endpoint)
    xs = np.arange(0, 100) / 100 * (endpoint - 0)
    ys = polyval4(poly, xs)
    length = 0.0
    for i in range(99):                                  # PathSeg.py:52-53 (sequential sum)
        length = length + math.sqrt((xs[i + 1] - xs[i]) ** 2 + (ys[i + 1] - ys[i]) ** 2)
    length = length + math.sqrt((endpoint - xs[99]) ** 2 + (y_end - ys[99]) ** 2)
    return dict(poly=poly, endpoint=endpoint, translation=translation, grad_st=grad_st,
                grad_end=grad_end, length=length, straight=is_straight)


# --------------------------------------------------------------------------- A2  Path.generate
def path_generate(draws):
    """PathGroup.generate's straight draw (PathGenerate.py:36) + Path.generate (Path.py:78-98).

    draws: float64[DRAWS_PER_PATH] in the fixed layout documented at DRAWS_PER_PATH."""
    path_straight = False if draws[0] > 0.01 else True
    segs = [pathseg_random(draws[1 + s * DRAWS_PER_SEG:1 + (s + 1) * DRAWS_PER_SEG], path_straight)
            for s in range(PATHSEGNUM)]
    return path_from_segs(segs, path_straight)


def path_from_segs(segs, path_straight):
    n = len(segs)
    # angle_abs (Path.py:276-289): cumulative sum of atan(GradEnd_k) - atan(GradSt_{k+1})
    ang = np.zeros(n)
    for i in range(1, n):
        a = 0.0
        for k in range(i):
            a = a + (math.atan(segs[k]["grad_end"]) - math.atan(segs[k + 1]["grad_st"]))
        ang[i] = a
    # translation_seg (Path.py:291-299)
    trans = np.zeros([n, 2])
    for idx in range(n):
        t = np.zeros(2)
        for i in range(idx):
            if i != 0:
                t = t + coord_rotation(segs[i]["translation"], ang[i])
            else:
                t = t + segs[i]["translation"]
        trans[idx] = t

    def point_transform(pt, i):                          # Path.py:224-233
        pt = np.asarray(pt, dtype=np.float64)
        if i != 0:
            pt = coord_rotation(pt, ang[i])
        if pt.ndim == 2:
            return pt + trans[i][:, None]
        return pt + trans[i]

    segpoint = [np.zeros(2)]
    for i in range(n):                                   # Path.py:86-90
        e = segs[i]["endpoint"]
        segpoint.append(point_transform(np.array([e, polyval4(segs[i]["poly"], e)]), i))
    segpoint = np.reshape(segpoint, [-1, 2])
    pts = []
    for i in range(n):                                   # Path.plot, Path.py:256-260
        x = np.arange(0, 100) / 100 * segs[i]["endpoint"]
        y = polyval4(segs[i]["poly"], x)
        pts.append(point_transform(np.array([x, y]), i).T)
    pathpoint = np.reshape(pts, [-1, 2])
    d = euclid(pathpoint[:-1], pathpoint[1:])
    length = 0.0
    for v in d:                                          # Path.py:93-94 (python sum, left to right)
        length = length + float(v)
    return dict(segs=segs, straight=path_straight, seg_rot=ang, seg_trans=trans,
                segpoint=segpoint, pathpoint=pathpoint, length=length, endpoint=segpoint[n])


# --------------------------------------------------------------------------- A3  draw_boundary
def draw_boundary(path, clearance):
    """Path.draw_boundary (Path.py:318-356)."""
    segs, ang, trans = path["segs"], path["seg_rot"], path["seg_trans"]
    n = len(segs)
    up_p, up_d, dn_p, dn_d = [], [], [], []
    for i in range(n):
        x = np.arange(0, 50) / 50 * segs[i]["endpoint"]
        y = polyval4(segs[i]["poly"], x)
        y_der = np.polyval(np.polyder(segs[i]["poly"]), x)
        pt = np.array([x, y])
        if i != 0:
            pt = coord_rotation(pt, ang[i])
        pt = (pt + trans[i][:, None]).T
        norm = coord_rotation(np.array([y_der, -np.ones(50)]), ang[i]).T
        norm = norm / np.sqrt(norm[:, 0] * norm[:, 0] + norm[:, 1] * norm[:, 1])[:, None]
        up_p.append(pt - 0.5 * clearance * norm)
        up_d.append(norm)
        dn_p.append(pt + 0.5 * clearance * norm)
        dn_d.append(-1 * norm)
    up_p, up_d, dn_p, dn_d = map(np.array, (up_p, up_d, dn_p, dn_d))
    end = np.reshape(path["endpoint"], [2])
    init_b = np.array([coord_rotation(up_p[0][0], np.pi / 50 * (i + 1)) for i in range(50)])
    end_b = np.array([coord_rotation(up_p[n - 1][49] - end, -np.pi / 50 * (i + 1)) + end for i in range(50)])
    boundary = np.concatenate([init_b[::-1], up_p.reshape(-1, 2), end_b, dn_p.reshape(-1, 2)[::-1]], axis=0)
    return dict(up_point=up_p, up_dir=up_d, down_point=dn_p, down_dir=dn_d,
                init=init_b, end=end_b, boundarypoint=boundary)


# --------------------------------------------------------------------------- A4  corridor canvas
def corridor_canvas(path, bnd, R, map_size, clearance):
    """Path.path_space rays (Path.py:113-134) + free_space_bydirection (Path.py:397-404).

    Returns a bool [2R,2R] canvas (True where the reference stores 255)."""
    step_len = 1 / R * map_size                           # Path.py:118
    step_c2i = map_size / R                               # Path.py:380 (a different expression!)
    dis = 0.8 * clearance / step_len
    n_steps = int(np.round(dis))
    end = np.reshape(path["endpoint"], [2])
    starts = np.concatenate([bnd["init"], bnd["end"], bnd["up_point"].reshape(-1, 2),
                             bnd["down_point"].reshape(-1, 2)], axis=0)
    d_init = -step_len * bnd["init"] / np.sqrt((bnd["init"] ** 2).sum(1))[:, None]
    v = end[None, :] - bnd["end"]
    d_end = step_len * v / np.sqrt((v ** 2).sum(1))[:, None]
    dirs = np.concatenate([d_init, d_end, step_len * bnd["up_dir"].reshape(-1, 2),
                           step_len * bnd["down_dir"].reshape(-1, 2)], axis=0)
    canvas = np.zeros([2 * R, 2 * R], dtype=bool)
    alive = np.ones(len(starts), dtype=bool)
    for k in range(n_steps):
        p = starts + k * dirs
        idx = np.round(p / step_c2i + R).astype(np.int64)
        inb = (idx[:, 0] > 0) & (idx[:, 0] < 2 * R) & (idx[:, 1] > 0) & (idx[:, 1] < 2 * R)  # strict, Path.py:400
        alive &= inb                                       # a ray stops at its first miss
        canvas[idx[alive, 0], idx[alive, 1]] = True
    return canvas


# --------------------------------------------------------------------------- A5  hull
def monotone_chain(points):
    """Exact convex hull of integer points, counter-clockwise in (first, second) coordinates,
    collinear points dropped, starting at the lexicographically smallest vertex. Replaces the
    Qhull call at Path.py:392 (same vertex set; Qhull's start vertex is implementation-defined)."""
    pts = sorted(set((int(a), int(b)) for a, b in points))
    if len(pts) <= 2:
        return np.array(pts, dtype=np.int64).reshape(-1, 2)

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])

    lower = []
    for p in pts:
        while len(lower) >= 2 and cross(lower[-2], lower[-1], p) <= 0:
            lower.pop()
        lower.append(p)
    upper = []
    for p in reversed(pts):
        while len(upper) >= 2 and cross(upper[-2], upper[-1], p) <= 0:
            upper.pop()
        upper.append(p)
    return np.array(lower[:-1] + upper[:-1], dtype=np.int64)


def convexhull(pathpoint_world, R, map_size, order="canonical"):
    """Path.convexhull (Path.py:388-395). order="scipy" reproduces Qhull's vertex order (needed
    only to replay the reference's torch.rand stream isle by isle); "canonical" = monotone_chain."""
    pts = euclidean2image(pathpoint_world, map_size / R, R)
    hull = monotone_chain(pts)
    if order == "scipy":
        from scipy.spatial import ConvexHull
        h = ConvexHull(pts)
        sp = pts[h.vertices]
        # same cyclic sequence, different start
        k = np.where((hull == sp[0]).all(1))[0]
        assert len(k) == 1 and len(sp) == len(hull)
        rolled = np.roll(hull, -int(k[0]), axis=0)
        assert (rolled == sp).all(), "monotone chain and Qhull disagree"
        return sp.astype(np.float64)
    return hull.astype(np.float64)


# --------------------------------------------------------------------------- rasters
# Corridor resample: torchvision 0.12's TENSOR path, restated from its published algorithm and pinned bit for bit to the
# primitives it calls (tests/test_oracle_golden.py::test_resample_rule_is_torchvisions_tensor_path builds that path from
# torch.linspace / bmm / torch.nn.functional.grid_sample, which are importable, and compares source indices):
#   F.rotate(img, angle)       -> matrix = _get_inverse_affine_matrix([0, 0], -angle, [0, 0], 1, [0, 0])
#   F.affine(img, 0, [tx, ty]) -> matrix = _get_inverse_affine_matrix([0, 0], 0, [tx, ty], 1, [0, 0])
#   theta = float32(matrix) [2,3];  base grid x_j = j + 0.5 - w/2, y_i = i + 0.5 - h/2 (torch.linspace, exact in float32)
#   rescaled theta: row 0 / (0.5 w), row 1 / (0.5 h)  (float32 divisions)
#   grid = base.bmm(rescaled theta^T)  -- sgemm with K = 3, an FMA chain in k order:  g = fl(fma(y, b, fl(x a)) + c)
#   grid_sample(nearest, zeros, align_corners=False):  u = fl(fl((g + 1) (size / 2)) - 0.5);  index = nearbyint(u)
# The float64 form of the same map (rounds 1-3) differs from it on ~5 pixels per million of a dense random image (near-ties
# of the rounding) and on none of the 11 golden corridor canvases.
F32 = np.float32


def _nearbyint(x):
    return np.rint(x)      # round half to even, like std::nearbyint in grid_sample's nearest mode


def _fma32(y, b, p):
    """fl32(y * b + p) for float32 arrays, exactly: the product of two float32 is exact in float64; the float64 sum s
    carries the TwoSum error e, and rounding s to float32 equals rounding s + e unless s sits exactly on a float32
    rounding boundary, where e decides the direction."""
    yb = y.astype(np.float64) * np.float64(b)
    p64 = p.astype(np.float64)
    s = yb + p64
    bb = s - yb
    e = (yb - (s - bb)) + (p64 - bb)
    r = s.astype(F32)                                  # round-half-even of s
    d = s - r.astype(np.float64)                       # exact
    cand = (d != 0) & (e != 0)
    if cand.any():
        # s = r + d; if s is exactly the midpoint of r and its neighbour nxt = r + 2d, the true value s + e lies on the
        # side e points to: nxt when e and d have the same sign, r otherwise
        nxt = np.nextafter(r, np.where(d > 0, F32(np.inf), F32(-np.inf)).astype(F32))
        half = np.abs(nxt.astype(np.float64) - r.astype(np.float64)) * 0.5
        r = np.where(cand & (np.abs(d) == half) & (np.sign(e) == np.sign(d)), nxt, r)
    return r.astype(F32)


def inverse_affine_matrix(angle_deg, translate):
    """torchvision.transforms.functional._get_inverse_affine_matrix(center=[0, 0], angle, translate, scale=1, shear=[0, 0])
    in Python floats, as published (torchvision 0.12 functional.py)."""
    rot = math.radians(angle_deg)
    tx, ty = float(translate[0]), float(translate[1])
    a = math.cos(rot - 0.0) / math.cos(0.0)
    b = -math.cos(rot - 0.0) * math.tan(0.0) / math.cos(0.0) - math.sin(rot)
    c = math.sin(rot - 0.0) / math.cos(0.0)
    d = -math.sin(rot - 0.0) * math.tan(0.0) / math.cos(0.0) + math.cos(rot)
    m = [d, -b, 0.0, -c, a, 0.0]
    m[2] += m[0] * (-0.0 - tx) + m[1] * (-0.0 - ty)
    m[5] += m[3] * (-0.0 - tx) + m[4] * (-0.0 - ty)
    m[2] += 0.0
    m[5] += 0.0
    return m


def affine_source_index(h, w, matrix):
    """Source (row, col) index arrays [h,w] of torchvision's tensor-path affine grid + nearest grid_sample for an h x w
    image and a 2x3 inverse matrix (see the block comment above).  Out-of-image indices are returned as they are."""
    th = np.asarray(matrix, dtype=np.float64).astype(F32).reshape(2, 3)
    x = (np.arange(w, dtype=np.float64) + (0.5 - w * 0.5)).astype(F32)[None, :] + np.zeros([h, 1], F32)
    y = (np.arange(h, dtype=np.float64) + (0.5 - h * 0.5)).astype(F32)[:, None] + np.zeros([1, w], F32)
    out = []
    for k, size in ((0, w), (1, h)):
        den = F32(0.5 * size)
        a, b, c = th[k, 0] / den, th[k, 1] / den, th[k, 2] / den
        g = _fma32(y, b, x * a) + c
        u = (g + F32(1)) * F32(size / 2) - F32(0.5)
        out.append(_nearbyint(u).astype(np.int64))
    return out[1], out[0]


def _gather(img, ii, jj, out_h, out_w):
    h, w = img.shape
    ii, jj = ii[:out_h, :out_w], jj[:out_h, :out_w]
    ok = (ii >= 0) & (ii < h) & (jj >= 0) & (jj < w)
    out = np.zeros([out_h, out_w], dtype=img.dtype)
    out[ok] = img[ii[ok], jj[ok]]
    return out


def rotate_nearest(img, angle_deg):
    """torchvision.transforms.functional.rotate(img, angle_deg, NEAREST) on a tensor (what T.RandomRotation(degrees=
    (a, a)) applies, Path.py:160-161, MapGenerate.py:103-104): counter-clockwise about the image centre, zero fill.
    RandomRotation.get_params returns float(torch.empty(1).uniform_(a, a).item()) — the angle ROUNDED TO FLOAT32 — and that
    is what F.rotate builds its matrix from; the points of the label rotate with the float64 angle (MapGenerate.py:70-89)."""
    h, w = img.shape
    angle32 = float(np.float32(angle_deg))
    ii, jj = affine_source_index(h, w, inverse_affine_matrix(-angle32, [0.0, 0.0]))
    return _gather(img, ii, jj, h, w)


def translate_nearest(img, tx, ty, out_h, out_w):
    """torchvision affine(img, angle=0, translate=[tx, ty], scale=1, shear=0) on a tensor, then the crop
    [0:out_h, 0:out_w] (Path.py:175,178; MapGenerate.py:105-106): out[i, j] = img[~(i - ty), ~(j - tx)], zero fill."""
    h, w = img.shape
    ii, jj = affine_source_index(h, w, inverse_affine_matrix(0.0, [tx, ty]))
    return _gather(img, ii, jj, out_h, out_w)


def raster_geometry(R):
    """Geometry of Path.plot_obstacles' raster (Path.py:36-49), in data units = pixels of the R x R map (the function is
    called with size = (R, R), MapGenerate.py:151).  matplotlib's default figure is 6.4 x 4.8 in at dpi 90 = 576 x 432 px
    with the axes at left 0.125, right 0.9, bottom 0.11, top 0.88: the axes box spans figure columns [72.0, 518.4]
    (446.4 px) and rows [51.84, 384.48] (332.64 px) and shows data x in [0, R] left to right, data y in [0, R] top to
    bottom.  The reference crops rows [53, 383) and columns [73, 517) — 330 x 444 px, slightly inside the axes box —
    and resizes the crop to R x R, so the centre of output pixel (i, j) shows the data point
        X = (j + 0.5) * AX + BX,   Y = (i + 0.5) * AY + BY
    with AX = 444 / 446.4, BX = (73 - 72) / 446.4 * R, AY = 330 / 332.64, BY = (53 - 51.84) / 332.64 * R.  Each circle
    is filled AND stroked in black with the default 1 pt line (1.25 figure px at dpi 90, half of it outside the rim):
    the ink reaches SX = 0.625 / 446.4 * R data units beyond the radius horizontally and SY = 0.625 / 332.64 * R
    vertically.  Returns (IAX, IAY, BX, BY, SX, SY) with IAX = 1 / AX = 446.4 / 444, IAY = 332.64 / 330; every value is
    computed with exactly these double operations in this order here and in csrc/ppn_device.h."""
    Rd = float(R)
    return (446.4 / 444.0, 332.64 / 330.0, (73.0 - 72.0) / 446.4 * Rd, (53.0 - 51.84) / 332.64 * Rd,
            0.625 / 446.4 * Rd, 0.625 / 332.64 * Rd)


def raster_ellipse(cx, cy, r, R):
    """A circle [cx = col, cy = row, r] as the reference's raster shows it, carried into the pixel-centre frame: the
    axis-aligned ellipse centred (cxp, cyp) with semi-axes (ex, ey); also (ex * ey)^2."""
    IAX, IAY, BX, BY, SX, SY = raster_geometry(R)
    cxp, cyp = (cx - BX) * IAX, (cy - BY) * IAY
    ex, ey = (r + SX) * IAX, (r + SY) * IAY
    q = ex * ey
    return cxp, cyp, ex, ey, q * q


def disc_raster(obstacles, R):
    """Obstacle raster rule standing in for Path.plot_obstacles (Path.py:36-49; matplotlib -> JPEG -> PIL '1' -> crop ->
    Resize is not integer-reproducible): pixel (i, j) is an obstacle iff its centre lies in the closed ellipse that a
    stroked circle [cx = col, cy = row, r] covers in the pixel-centre frame (raster_geometry, raster_ellipse):
        (((j + 0.5) - cxp) * ey)^2 + (((i + 0.5) - cyp) * ex)^2 <= (ex * ey)^2     (unfused IEEE double, in this order).
    Against the reference's own run (tests/golden/g15): IoU 0.984-0.994, 0.3 % of the pixels differ, of both signs
    (antialiasing, JPEG ringing and the dither of convert('1') are what is left).  Returns bool [R,R], True = obstacle."""
    occ = np.zeros([R, R], dtype=bool)
    yc = (np.arange(R) + 0.5)[:, None]
    xc = (np.arange(R) + 0.5)[None, :]
    for cx, cy, r in np.asarray(obstacles, dtype=np.float64).reshape(-1, 3):
        cxp, cyp, ex, ey, rhs = raster_ellipse(cx, cy, r, R)
        a = (xc - cxp) * ey
        b = (yc - cyp) * ex
        occ |= (a * a + b * b) <= rhs
    return occ


# --------------------------------------------------------------------------- A6  normalisation
def space_normalization(path, bnd, canvas, hull_raw, R, map_size):
    """Path.space_normalization with point_trans=True, check_free=False (Path.py:157-189)."""
    E = path["endpoint"]
    rotation = math.atan(E[1] / E[0]) / np.pi * 180 + (-135)
    rad = -rotation / 180 * np.pi
    hull = coord_rotation((hull_raw - np.array([R, R])).T, rad).T + np.array([R, R])
    hull_center = hull.mean(axis=0)                        # torch.mean over vertices (Path.py:167)
    t = np.array([R / 2, R / 2]) - hull_center             # (t_row, t_col)
    # Path.py:171 swaps to [t_col, t_row] for torchvision's (tx, ty); Path.py:176 swaps back.
    hull_n = hull + t
    step = map_size / R

    def to_image(p):                                       # Path.py:180-188
        q = coord_rotation(np.asarray(p).T, rad).T
        return euclidean2image(q, step, R).astype(np.float64) + t

    space = translate_nearest(rotate_nearest(canvas, -rotation), t[1], t[0], R, R)
    return dict(rotation=rotation, trans_rc=t, hull=hull_n,
                segpoint_image=to_image(path["segpoint"]),
                pathpoint_image=to_image(path["pathpoint"]),
                boundarypoint_image=to_image(bnd["boundarypoint"]),
                space=space)


# --------------------------------------------------------------------------- A10 boundary_check
def boundary_check(hull, angle_deg, translation_rc, R):
    """Path.boundary_check (Path.py:100-111); MapOffset = R/2 (PathGenerate.py:28)."""
    off = R / 2
    h = coord_rotation((hull - off).T, angle_deg / 180 * np.pi).T + np.asarray(translation_rc, dtype=np.float64) + off
    ok = not ((h[:, 0] < 0) | (h[:, 0] >= R) | (h[:, 1] < 0) | (h[:, 1] >= R)).any()
    return ok, h


# --------------------------------------------------------------------------- A7  search_isle
def search_isle(pathpoint_image, hull, R, map_size, clearance, width_coef=0.2):
    """Path.search_isle (Path.py:502-537). Returns list of (a, b) slice bounds into PathPoint
    and a flag word."""
    step_len = 1 / R * map_size
    thr = int(np.round(clearance / step_len * width_coef))
    P = pathpoint_image
    isles, flags = [], 0
    nh = len(hull)
    for i in range(nh):
        j = 0 if i == nh - 1 else i + 1
        e = hull[j] - hull[i]
        if math.sqrt(e[0] * e[0] + e[1] * e[1]) > 5 / step_len:
            d0 = euclid(P, hull[i][None, :])
            d1 = euclid(P, hull[j][None, :])
            a0 = int(np.argmin(d0))                         # first strict minimum (Path.py:522-527)
            a1 = int(np.argmin(d1))
            lo, hi = min(a0, a1), max(a0, a1)
            if hi == lo:
                flags |= FLAG_EMPTY_ISLE                    # reference: IndexError on boundary[0]
                continue
            b = P[lo:hi]
            v = b[0] - b[-1]
            nrm = math.sqrt(v[0] * v[0] + v[1] * v[1])
            if nrm == 0.0:
                continue                                    # NaN direction: never exceeds thr, p == last
            dirv = v / nrm
            dirv = np.array([dirv[1], -dirv[0]])
            dis = np.abs(dot2(b[:, 0] - b[0, 0], b[:, 1] - b[0, 1], dirv))
            over = np.nonzero(dis > (thr if ARITH == "blas" else thr + TIE_EPS))[0]
            k = int(over[0]) if len(over) else len(b) - 1
            if (b[k] != b[-1]).any():                       # Path.py:535
                isles.append((lo, hi))
    return isles, flags


# --------------------------------------------------------------------------- A8  set_obstacles
def set_obstacles(pathpoint_image, isles, R, map_size, clearance, feed):
    """Path.set_obstacles (Path.py:463-500). `feed` yields float32 uniforms in torch.rand order.

    Float widths follow the reference's torch/numpy mix: `radius`, `motion`, the jitter factor
    are float32 tensors; coordinates, distances and the clipped radius are float64.
    Returns (obstacles [n,3] as [col,row,r], flags)."""
    f32 = np.float32
    P = pathpoint_image
    Podd = P[1::2]                                          # Path.py:487-489 (i % 2)
    c_px = clearance / map_size * R
    size_clearance = clearance / map_size * R * 1.1
    out, flags = [], 0
    for (lo, hi) in isles:
        isle = P[lo:hi]
        center = (isle[0] + isle[-1]) / 2
        v = isle[0] - isle[-1]
        dir_t = v / math.sqrt(v[0] * v[0] + v[1] * v[1])
        dir_n = np.array([dir_t[1], -dir_t[0]])
        mid = isle[int(len(isle) / 2)] - center
        if not (float(np.ravel(dot2(mid[0], mid[1], dir_n))[0]) < 0):
            dir_n = -dir_n
        dis = np.abs(dot2(isle[:, 0] - isle[0, 0], isle[:, 1] - isle[0, 1], dir_n))
        size_max = float(dis.max()) * 2
        if ARITH == "blas":
            peak = isle[int(np.argmax(dis))]                # list.index(max): first maximum
        else:
            peak = isle[int(np.nonzero(dis >= dis.max() - TIE_EPS)[0][0])]
        obs_sum = f32(0)                                    # sum(obs_size): float32 tensors
        n_obs = 0
        size_pre = 0.0                                      # int 0, later float32 tensor or float64 scalar
        size_pre_is_f32 = True
        coord = None
        tries = 0
        while float(obs_sum) < size_max:
            if tries >= POCKET_TRY_CAP:
                flags |= FLAG_POCKET_CAP
                break
            tries += 1
            radius = f32(f32(feed.next() * f32(size_max)) / f32(2))
            random_normal = feed.next() if n_obs else f32(1)
            if size_pre_is_f32:
                acc = f32(radius + f32(size_pre))
            else:                                           # float32 tensor + numpy float64 scalar -> float32
                acc = f32(radius + f32(size_pre))
            acc = f32(acc + (f32(size_clearance) if n_obs == 0 else f32(0)))
            motion = f32(random_normal * acc)
            alt = f32(radius - obs_sum)
            if alt > motion:                                # python max(motion, alt)
                motion = alt
            base = peak if n_obs == 0 else coord
            coord = base + float(motion) * dir_n
            if n_obs:
                jit = f32(f32(f32(f32(feed.next() - f32(0.5)) / f32(0.5)) * radius) / f32(2))
                coord = coord + float(jit) * dir_t
            md = float(euclid(Podd, coord[None, :]).min())
            rad_out = float(radius)
            clipped = False
            if md < float(f32(radius + f32(c_px))):
                rad_out = md - c_px                         # numpy float64 from here on
                clipped = True
            if rad_out > 0:
                obs_sum = f32(obs_sum + motion)
                n_obs += 1
                size_pre = rad_out
                size_pre_is_f32 = not clipped
                out.append([coord[1], coord[0], rad_out])
    return np.array(out, dtype=np.float64).reshape(-1, 3), flags


# --------------------------------------------------------------------------- stage A driver
def make_path(draws, R, map_size, clearance, pocket_feed, hull_order="canonical"):
    """PathGroup.generate body for one path (PathGenerate.py:36-44): generate, draw_boundary,
    path_obstacles (Path.py:144-155)."""
    path = path_generate(draws)
    bnd = draw_boundary(path, clearance)
    canvas = corridor_canvas(path, bnd, R, map_size, clearance)
    hull_raw = convexhull(path["pathpoint"], R, map_size, order=hull_order)
    nrm = space_normalization(path, bnd, canvas, hull_raw, R, map_size)
    pocket_feed.rotation_draw()
    flags = 0
    if path["straight"]:
        isles, obstacles = [], np.zeros([0, 3])
    else:
        isles, f1 = search_isle(nrm["pathpoint_image"], nrm["hull"], R, map_size, clearance)
        obstacles, f2 = set_obstacles(nrm["pathpoint_image"], isles, R, map_size, clearance, pocket_feed)
        flags = f1 | f2
    rec = dict(path)
    rec.update(bnd)
    rec.update(nrm)
    rec.update(canvas=canvas, hull_raw=hull_raw, isles=isles, obstacles=obstacles, flags=flags)
    return rec


def generate_paths(source, n_paths, R, map_size, clearance, first_path_id=0, hull_order="canonical"):
    recs = []
    for p in range(n_paths):
        pid = first_path_id + p
        draws = source.path_draws(pid)
        feed = _FloatFeed(source.pocket_floats(pid, 3 * POCKET_TRY_CAP * 32))
        recs.append(make_path(draws, R, map_size, clearance, feed, hull_order))
    return recs


# --------------------------------------------------------------------------- A11/A12 stage B
def place_and_label(prec, angle, t, R):
    """MapGenerate.generate body after an accepted draw (MapGenerate.py:68-93)."""
    rad = -angle / 180 * np.pi
    c = np.array([R / 2, R / 2])
    tr = np.array([t[1], t[0]], dtype=np.float64)
    segpoint = coord_rotation((prec["segpoint_image"] - c).T, rad).T + c + tr
    pathpoint = coord_rotation((prec["pathpoint_image"] - c).T, rad).T + c + tr
    pobs = []
    for o in prec["obstacles"]:
        q = coord_rotation(np.array([o[1], o[0]]) - c, rad).T + c + tr
        pobs.append([q[1], q[0], o[2]])
    return segpoint, pathpoint, np.array(pobs, dtype=np.float64).reshape(-1, 3)


def obstacle_filter(d, K, pathpoint, R, map_size, obstacles_size, clearance):
    """MapGenerate.generate_map_randomly accept loop (MapGenerate.py:128-143).
    d: float64[3K] in reference draw order (K x, K y, K size). Returns (accept bool[K], kept [n,3])."""
    coord_img = np.stack([d[0:K] * map_size, d[K:2 * K] * map_size], axis=1) / map_size * R
    radius_img = d[2 * K:3 * K] * obstacles_size / map_size * R
    Podd = pathpoint[1::2]
    c_px = clearance / map_size * R
    md = np.array([euclid(Podd, coord_img[k][None, :]).min() for k in range(K)]) if K else np.zeros(0)
    accept = md > radius_img + c_px
    kept = np.stack([coord_img[accept, 1], coord_img[accept, 0], radius_img[accept]], axis=1)
    return accept, kept


def paint_markers(grid, init, end):
    """add_init_end_single (process_map.py:119-145): 7x7 squares at the rounded start and goal."""
    R = grid.shape[0]
    for pt in (init, end):
        r0, c0 = int(np.round(pt[0])), int(np.round(pt[1]))
        for dj in range(-3, 4):
            for dk in range(-3, 4):
                if 0 <= r0 + dj < R and 0 <= c0 + dk < R:
                    grid[r0 + dj, c0 + dk] = GRID_MARK
    return grid


def compose_grid(prec, angle, t, obstacles_all, segpoint, R):
    """MapGenerate.py:102-113: rotate+translate the corridor mask, add the obstacle raster
    (saturating sum: corridor wins over obstacles), paint start/goal."""
    corridor = translate_nearest(rotate_nearest(prec["space"], float(-angle)), float(t[0]), float(t[1]), R, R)
    occ = disc_raster(obstacles_all, R)
    grid = np.where(corridor | ~occ, GRID_FREE, GRID_OBST).astype(np.uint8)
    return paint_markers(grid, segpoint[0], segpoint[PATHSEGNUM])


def label_masks(prec, angle, t, pathpoint, R, bound=None):
    """The two per-map label images process_map.py writes next:
      mask_path  (generate_gen_path, :148-163): every 5th label point with 0 < round(p) < bound -> 255, else 0;
      mask_space (generate_seg_space, :166-191): the target path's Space rotated by -angle, translated by t, thresholded
                 to {0,1} — the same nearest-neighbour rule as the corridor in compose_grid (torchvision: parity unpinned).
    `bound` is the reference's hard-coded 224 (default: R)."""
    bound = R if bound is None else bound
    mp = np.zeros([R, R], dtype=np.uint8)
    for step in range(0, len(pathpoint), 5):
        r0, c0 = int(np.round(pathpoint[step][0])), int(np.round(pathpoint[step][1]))
        if 0 < r0 < bound and 0 < c0 < bound and r0 < R and c0 < R:
            mp[r0, c0] = 255
    ms = translate_nearest(rotate_nearest(prec["space"], float(-angle)), float(t[0]), float(t[1]), R, R).astype(np.uint8)
    return mp, ms


def make_map(prec, map_id, source, R, map_size, obstacles_size, K, clearance, want_grid=True):
    """One placement of one target path (MapGenerate.py:57-124), bounded retry."""
    flags = 0
    attempts = 0
    while True:
        d = source.place_draws(map_id, attempts)
        attempts += 1
        angle = d[0] * 360 - 180
        t = np.array(d[1:3] * R - R / 2, dtype=int)      # truncation toward zero (MapGenerate.py:64)
        ok, _ = boundary_check(prec["hull"], -angle, [t[1], t[0]], R)
        if ok:
            break
        if attempts >= PLACE_TRY_CAP:
            flags |= FLAG_PLACE_CAP
            break
    segpoint, pathpoint, pobs = place_and_label(prec, angle, t, R)
    if hasattr(source, "map_rotation_draw"):
        source.map_rotation_draw()                       # MapGenerate.py:103-104, before generate_map_randomly
    od = source.obst_draws(map_id, K)
    accept, kept = obstacle_filter(od, K, pathpoint, R, map_size, obstacles_size, clearance)
    obstacles_all = np.concatenate([kept, pobs], axis=0)
    out = dict(angle=angle, translation=t, attempts=attempts, segpoint=segpoint, pathpoint=pathpoint,
               accept=accept, obstacles=obstacles_all, n_random=len(kept), flags=flags,
               init=segpoint[0], end=segpoint[PATHSEGNUM], length=prec["length"])
    if want_grid:
        out["grid"] = compose_grid(prec, angle, t, obstacles_all, segpoint, R)
    return out


def generate_maps(source, precs, R, map_size, obstacles_size, K, clearance, placements,
                  first_map_id=0, want_grid=True):
    """MapGenerate.generate loop nest (MapGenerate.py:48-124): for each target path, `placements`
    accepted placements; map id = path index * placements + k."""
    maps = []
    for j, prec in enumerate(precs):
        for k in range(placements):
            mid = first_map_id + j * placements + k
            maps.append(make_map(prec, mid, source, R, map_size, obstacles_size, K, clearance, want_grid))
    return maps
