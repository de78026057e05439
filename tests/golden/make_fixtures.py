#!/usr/bin/env python3
"""Golden-vector generator (runs ONLY in the build container, never on the GPU box).

Imports the reference's EDaGe-PP modules *unmodified* from /root/reference/EDaGe-PP and
records their inputs/outputs under fixed seeds as small .npz/.json fixtures next to this
script.  Absent third-party modules are replaced by `sys.modules` stubs:

  cv2, imgviz            -> empty modules (imported, never called on the recorded paths)
  torchvision            -> ToTensor = u8 HWC / PIL -> f32 CHW / 255 ; RandomRotation and
                            functional.affine = identity on the raster — but RandomRotation still draws its
                            angle as torchvision 0.12 does (`torch.empty(1).uniform_(lo, hi)`), so the global
                            torch stream advances exactly as in the reference ; utils.save_image = no-op ;
                            Resize / ToPILImage = identity

so every *non-raster* quantity recorded here is a genuine reference result; the rotated /
resampled rasters (`Path.Space`, the JPEG occupancy image) are NOT recorded because the
stubs make them meaningless (torchvision / libjpeg rasters are "parity unpinned").

Nothing from the reference is copied: fixtures hold numbers only.

    python tests/golden/make_fixtures.py        # rewrites tests/golden/*.npz
"""
import io
import json
import os
import signal
import sys
import tempfile
import types
import contextlib

import numpy as np
import torch

REF = "/root/reference/EDaGe-PP"
OUT = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------- stubs
def _install_stubs():
    from PIL import Image

    def to_tensor(pic):
        if isinstance(pic, Image.Image):
            a = np.asarray(pic)
        else:
            a = np.asarray(pic)
        if a.ndim == 2:
            a = a[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
        if t.dtype == torch.uint8:
            t = t.float() / 255.0
        return t

    class ToTensor:
        def __call__(self, pic):
            return to_tensor(pic)

    class Identity:
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            return x

    class RandomRotation:
        """Raster identity, but with torchvision 0.12's draw: RandomRotation.forward -> get_params(degrees) ->
        `torch.empty(1).uniform_(lo, hi)`, one draw from the global torch generator even when lo == hi
        (Path.py:160-161, MapGenerate.py:103-104)."""

        def __init__(self, degrees, *a, **k):
            self.degrees = [float(np.ravel(d)[0]) for d in degrees]

        def __call__(self, x):
            torch.empty(1).uniform_(self.degrees[0], self.degrees[1])
            return x

    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")
    tvu = types.ModuleType("torchvision.utils")
    tvt.ToTensor = ToTensor
    tvt.RandomRotation = RandomRotation
    tvt.Resize = Identity
    tvt.ToPILImage = Identity
    tvt.CenterCrop = Identity
    tvf.affine = lambda img, **k: img
    tvt.functional = tvf
    tvu.save_image = lambda *a, **k: None
    tv.transforms = tvt
    tv.utils = tvu
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tvt
    sys.modules["torchvision.transforms.functional"] = tvf
    sys.modules["torchvision.utils"] = tvu
    sys.modules["cv2"] = types.ModuleType("cv2")
    sys.modules["imgviz"] = types.ModuleType("imgviz")
    import matplotlib
    matplotlib.use("Agg")


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield


def _f64(x):
    return np.asarray(x, dtype=np.float64)


class RefHang(Exception):
    pass


@contextlib.contextmanager
def time_limit(seconds):
    """The reference has unbounded loops (Path.py:478 `while sum(obs_size) < size_max`,
    MapGenerate.py:58-62 up to 1e6 retries); a case that does not finish is skipped and listed."""
    def handler(signum, frame):
        raise RefHang()
    old = signal.signal(signal.SIGALRM, handler)
    signal.alarm(seconds)
    try:
        yield
    finally:
        signal.alarm(0)
        signal.signal(signal.SIGALRM, old)


# ----------------------------------------------------------------------------- G1
def fixture_pathseg(PathSegMod):
    out = {}
    for seed in (0, 1, 2, 7):
        for forced in (False, True):
            np.random.seed(seed)
            st = np.random.get_state()
            seg = PathSegMod.PathSeg(polyorder=4, dim=2, is_straight=forced)
            seg.random()
            st2 = np.random.get_state()
            # replay the raw draws that were consumed
            np.random.set_state(st)
            n = (0 if forced else 1) + 1000 + 1
            draws = np.random.random(n)
            assert np.random.get_state()[2] == st2[2]
            k = f"s{seed}_f{int(forced)}"
            out[k + "_draws"] = draws
            out[k + "_poly"] = _f64(seg.Poly)
            out[k + "_endpoint"] = _f64(seg.EndPoint).reshape(-1)
            out[k + "_translation"] = _f64(seg.Translation).reshape(-1)
            out[k + "_grad"] = _f64([np.ravel(seg.GradSt)[0], np.ravel(seg.GradEnd)[0]])
            out[k + "_length"] = _f64(seg.Length).reshape(-1)
            out[k + "_straight"] = np.array([int(seg.is_straight)])
    np.savez_compressed(os.path.join(OUT, "g1_pathseg.npz"), **out)


# ----------------------------------------------------------------------------- G2..G8
def run_path(PathMod, seed, R, clearance, is_straight, map_size=50, tseed=None):
    """One full reference Path: generate -> draw_boundary -> path_obstacles (stubbed rasters)."""
    np.random.seed(seed)
    torch.manual_seed(seed if tseed is None else tseed)
    st = np.random.get_state()
    rec = {}
    with quiet():
        p = PathMod.Path(seg_num=10, poly_order=4, dim=2, clearance=clearance, is_straight=is_straight)
        p.generate(show_now=False)
        rec["seg_poly"] = _f64([s.Poly for s in p.PathSeg])
        rec["seg_endpoint"] = _f64([np.ravel(s.EndPoint)[0] for s in p.PathSeg])
        rec["seg_straight"] = np.array([int(s.is_straight) for s in p.PathSeg])
        rec["seg_rotation"] = _f64([float(s.Rotation) for s in p.PathSeg])
        rec["seg_translation"] = _f64([np.ravel(s.Translation) for s in p.PathSeg])
        rec["seg_length"] = _f64([np.ravel(s.Length)[0] for s in p.PathSeg])
        rec["segpoint"] = _f64(p.SegPoint)
        rec["pathpoint_world"] = _f64(p.PathPoint)
        rec["length"] = _f64(p.Length).reshape(-1)
        # number of numpy draws consumed by generate()
        st2 = np.random.get_state()
        np.random.set_state(st)
        n_straight = int(rec["seg_straight"].sum())
        n = 10 * 1001 + (0 if is_straight else 10)
        rec["draws"] = np.random.random(n)
        assert np.random.get_state()[2] == st2[2] and (np.random.get_state()[1] == st2[1]).all()

        p.draw_boundary(show_now=False)
        rec["boundarypoint_world"] = _f64(p.BoundaryPoint)
        rec["up_dir"] = _f64(p.Boundary.upboundary.direction)
        rec["up_point"] = _f64(p.Boundary.upboundary.point)
        rec["down_point"] = _f64(p.Boundary.downboundary.point)
        rec["init_boundary"] = _f64(p.Boundary.initboundary)
        rec["end_boundary"] = _f64(p.Boundary.endboundary)

        # --- path_space pieces, called the way Path.path_space does (Path.py:113-142)
        p.Resolution, p.MapSize, p.MapOffset = R, map_size, R / 2
        space = torch.zeros([R * 2, R * 2])
        step_len = 1 / R * map_size
        dis = 0.8 * clearance / step_len
        B = p.Boundary
        for i in range(50):
            d = -step_len * B.initboundary[i] / np.linalg.norm(B.initboundary[i])
            space = p.free_space_bydirection(space, B.initboundary[i], d, dis, mapoffset=R)
        for i in range(50):
            v = np.reshape(p.EndPoint, [2]) - B.endboundary[i]
            d = step_len * v / np.linalg.norm(v)
            space = p.free_space_bydirection(space, B.endboundary[i], d, dis, mapoffset=R)
        for i in range(10):
            for j in range(50):
                space = p.free_space_bydirection(space, B.upboundary.point[i][j], step_len * B.upboundary.direction[i][j], dis, mapoffset=R)
        for i in range(10):
            for j in range(50):
                space = p.free_space_bydirection(space, B.downboundary.point[i][j], step_len * B.downboundary.direction[i][j], dis, mapoffset=R)
        nz = torch.nonzero(space).numpy().astype(np.int32)
        rec["canvas_nz"] = nz
        assert set(np.unique(space.numpy())) <= {0.0, 255.0}

    # fresh object for the real call chain (path_obstacles consumes no numpy draws)
    np.random.seed(seed)
    torch.manual_seed(seed if tseed is None else tseed)
    with quiet():
        p = PathMod.Path(seg_num=10, poly_order=4, dim=2, clearance=clearance, is_straight=is_straight)
        p.generate(show_now=False)
        p.draw_boundary(show_now=False)
        # hull in scipy's order before normalisation
        p.Resolution, p.MapSize, p.MapOffset = R, map_size, R / 2
        hull_pts, hull_c = p.convexhull()
        rec["hull_raw"] = _f64(hull_pts.numpy())
        tstate = torch.get_rng_state()
        ok = p.path_obstacles(resolution=R, map_size=map_size, map_offset=R / 2)
        # torch draws consumed by path_obstacles: replay a generous prefix.  The first n_rotation_draws of them
        # belong to space_normalization's RandomRotation (Path.py:160-161), the rest to set_obstacles
        tstate2 = torch.get_rng_state()
        torch.set_rng_state(tstate)
        rec["torch_draws"] = np.array([torch.rand(1).item() for _ in range(600)], dtype=np.float32)
        rec["n_rotation_draws"] = np.array([1])
        torch.set_rng_state(tstate2)
    rec["ok"] = np.array([int(bool(ok))])
    rec["rotation"] = _f64(p.Rotation).reshape(-1)
    rec["translation"] = _f64([float(p.Translation[0]), float(p.Translation[1])])
    rec["hull_norm"] = _f64(np.asarray(p.ConvexHull))
    rec["segpoint_image"] = _f64(p.SegPointImage)
    rec["pathpoint_image"] = _f64(np.asarray(p.PathPoint))
    rec["boundarypoint_image"] = _f64(p.BoundaryPoint)
    obs = [[float(o[0]), float(o[1]), float(o[2])] for o in p.obstacles]
    rec["obstacles"] = _f64(obs).reshape(-1, 3)
    # isles: recover slice bounds by re-running search_isle (pure function of PathPoint/ConvexHull)
    if not is_straight:
        with quiet():
            isles = p.search_isle()
        pp = np.asarray(p.PathPoint)
        bounds = []
        for isle in isles:
            a = np.asarray(isle)
            n = len(a)
            found = None
            for s in range(0, len(pp) - n + 1):
                if np.array_equal(pp[s:s + n], a):
                    found = s
                    break
            assert found is not None
            bounds.append([found, found + n])
        rec["isle_bounds"] = np.array(bounds, dtype=np.int32).reshape(-1, 2)
    else:
        rec["isle_bounds"] = np.zeros((0, 2), np.int32)
    return rec, p


def fixture_paths(PathMod):
    cases = [
        # (seed, R, clearance, straight)
        (0, 64, 3, False), (1, 64, 3, False), (2, 64, 1, False),
        (0, 224, 1, False), (3, 224, 3, False),
        (0, 256, 3, False), (1, 256, 3, False), (4, 256, 1, False), (5, 256, 3, True),
        (6, 128, 3, False), (8, 64, 3, False), (9, 256, 3, False),
    ]
    out = {}
    names = []
    hung = []
    for (seed, R, c, straight) in cases:
        name = f"s{seed}_R{R}_c{c}_st{int(straight)}"
        try:
            with time_limit(40):
                rec, _ = run_path(PathMod, seed, R, c, straight)
        except RefHang:
            hung.append(name)
            print("reference did not terminate:", name, flush=True)
            continue
        print("path case", name, "ok", flush=True)
        names.append(name)
        for k, v in rec.items():
            out[f"{name}/{k}"] = v
    out["cases"] = np.array(names)
    out["reference_nonterminating"] = np.array(hung)
    np.savez_compressed(os.path.join(OUT, "g2_paths.npz"), **out)


# ----------------------------------------------------------------------------- G9
def fixture_boundary_check(PathMod):
    for seed in (11, 12, 13, 14):
        try:
            with time_limit(40):
                rec, p = run_path(PathMod, seed, 256, 3, False)
            break
        except RefHang:
            print("boundary_check fixture: reference did not terminate for seed", seed, flush=True)
    rng = np.random.RandomState(123)
    angles = rng.random_sample(1000) * 360 - 180
    trans = np.array(rng.random_sample((1000, 2)) * 256 - 128, dtype=int)
    oks = []
    hulls = []
    for a, t in zip(angles, trans):
        ok, h = p.boundary_check(-np.array([a]), [t[1], t[0]])
        oks.append(int(ok))
        hulls.append(np.asarray(h))
    np.savez_compressed(os.path.join(OUT, "g9_boundary_check.npz"),
                        hull_norm=rec["hull_norm"], angles=angles, trans=trans,
                        ok=np.array(oks, np.int8), hull_out=_f64(hulls[:16]),
                        R=np.array([256]))


# ----------------------------------------------------------------------------- G10 / config 1
def fixture_mapgenerate(MapGenMod):
    """Config 1: np.random.seed(0); torch.manual_seed(0); MapGenerate(10, 64, 50, 5, 20, 3).generate(100)."""
    captured = []

    def fake_plot_obstacles(size, obstacles, resolution=(224, 224)):
        captured.append([[float(o[0]), float(o[1]), float(o[2])] for o in obstacles])
        return torch.ones([3, resolution[0], resolution[1]])

    MapGenMod.plot_obstacles = fake_plot_obstacles
    R = 64
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.chdir(tmp)
    try:
        np.random.seed(0)
        torch.manual_seed(0)
        MapGenMod.cnt = 0
        with quiet():
            mg = MapGenMod.MapGenerate(path_num=10, resolution=R, map_size=50, obstacles_size=5,
                                       obstacles_num=20, clearance=3)
            # stage-A summaries for each target path
            stageA = {}
            for j, tp in enumerate(mg.PathGroup.TargetPaths):
                stageA[f"p{j}/hull_norm"] = _f64(np.asarray(tp.ConvexHull))
                stageA[f"p{j}/segpoint_image"] = _f64(tp.SegPointImage)
                stageA[f"p{j}/pathpoint_image"] = _f64(np.asarray(tp.PathPoint))
                stageA[f"p{j}/obstacles"] = _f64([[float(o[0]), float(o[1]), float(o[2])] for o in tp.obstacles]).reshape(-1, 3)
                stageA[f"p{j}/length"] = _f64(tp.Length).reshape(-1)
                stageA[f"p{j}/straight"] = np.array([int(tp.is_straight)])
                stageA[f"p{j}/rotation"] = _f64(tp.Rotation).reshape(-1)
                stageA[f"p{j}/translation"] = _f64([float(tp.Translation[0]), float(tp.Translation[1])])
            mg.generate(map_num=100, folder_path=os.path.join(tmp, "out"), round_index=0)
        with open("unsolved_problems.txt") as f:
            problems = [json.loads(l) for l in f]
    finally:
        os.chdir(cwd)
    assert len(mg.MapLabel) == 100 and len(problems) == 100 and len(captured) == 100
    out = dict(stageA)
    out["angle"] = _f64([np.ravel(l[1])[0] for l in mg.MapLabel])
    out["translation"] = np.array([[int(l[2][0]), int(l[2][1])] for l in mg.MapLabel], dtype=np.int64)
    out["segpoint"] = _f64([l[3] for l in mg.MapLabel])
    out["pathpoint"] = _f64([l[4] for l in mg.MapLabel])
    out["n_obs"] = np.array([len(c) for c in captured], dtype=np.int32)
    out["obstacles"] = _f64([o for c in captured for o in c]).reshape(-1, 3)
    out["problem_index"] = np.array([p["Index"] for p in problems])
    out["problem_length"] = _f64([p["Length"] for p in problems])
    out["problem_init"] = _f64([p["Init"] for p in problems])
    out["problem_end"] = _f64([p["End"] for p in problems])
    # final numpy RNG position: lets the oracle prove it consumed exactly the same stream
    out["np_next_draws"] = np.random.random(4)
    # and the torch one: 10 + 100 RandomRotation draws (Path.py:160, MapGenerate.py:103) + set_obstacles' torch.rand
    out["torch_next_draws"] = np.array([torch.rand(1).item() for _ in range(4)], dtype=np.float32)
    np.savez_compressed(os.path.join(OUT, "g10_config1_R64.npz"), **out)


# ----------------------------------------------------------------------------- G11 / G12
def fixture_plan_tail(PM):
    rng = np.random.RandomState(5)
    # collision_check_circle_edge on random segments x obstacle sets
    S, E, OBS, HIT = [], [], [], []
    nob = []
    for t in range(400):
        K = int(rng.randint(1, 30))
        obs = np.concatenate([rng.random_sample((K, 2)) * 224, rng.random_sample((K, 1)) * 18], axis=1)
        s = rng.random_sample(2) * 240 - 8
        e = s + (rng.random_sample(2) - 0.5) * 60
        s32 = torch.tensor(s, dtype=torch.float32)
        e32 = torch.tensor(e, dtype=torch.float32)
        with quiet():
            hit = PM.collision_check_circle_edge(s32, e32, [list(o) for o in obs], 1 / 50 * 224)
        S.append(s32.numpy()); E.append(e32.numpy()); OBS.append(obs); HIT.append(int(bool(hit))); nob.append(K)
    np.savez_compressed(os.path.join(OUT, "g11_collision.npz"), s=np.array(S), e=np.array(E),
                        obs=np.concatenate(OBS, 0), n_obs=np.array(nob, np.int32), hit=np.array(HIT, np.int8),
                        clearance=np.array([1 / 50 * 224]))

    # extract_path on synthetic heat maps (a smooth ridge along a curve)
    from PIL import Image
    out = {}
    ncase = 0
    for t in range(12):
        R = 224 if t % 2 == 0 else 256
        yy, xx = np.mgrid[0:R, 0:R].astype(np.float64)
        # ridge: quadratic bezier from init to end through a random control point
        init = rng.random_sample(2) * (R * 0.3) + R * 0.1
        end = rng.random_sample(2) * (R * 0.3) + R * 0.6
        ctrl = rng.random_sample(2) * R * 0.8 + R * 0.1
        ts = np.linspace(0, 1, 400)[:, None]
        curve = (1 - ts) ** 2 * init + 2 * ts * (1 - ts) * ctrl + ts ** 2 * end
        d2 = np.min((yy[None] - curve[:, 0, None, None]) ** 2 + (xx[None] - curve[:, 1, None, None]) ** 2, axis=0)
        heat = np.exp(-d2 / (2 * 4.0 ** 2))
        if t >= 10:
            heat[:] = 0  # failure case: all-zero candidates
        img = (heat * 255).astype(np.uint8)
        with quiet():
            ok, path = PM.extract_path(Image.fromarray(img, mode="L"), init_state=init, end_state=end, down_sample_rate=2)
        out[f"c{ncase}_img"] = img
        out[f"c{ncase}_init"] = init
        out[f"c{ncase}_end"] = end
        out[f"c{ncase}_ok"] = np.array([int(bool(ok))])
        out[f"c{ncase}_path"] = path.numpy().astype(np.float64) if ok else np.zeros((0, 2))
        ncase += 1
    out["ncase"] = np.array([ncase])
    np.savez_compressed(os.path.join(OUT, "g11_extract_path.npz"), **out)

    # add_init_end_single on a zero image
    img = torch.zeros([3, 64, 64])
    res = PM.add_init_end_single(img, np.array([3.4, 61.5]), np.array([30.5, 31.5]))
    np.savez_compressed(os.path.join(OUT, "g12_init_end.npz"), out=res.numpy().astype(np.float32),
                        init=np.array([3.4, 61.5]), end=np.array([30.5, 31.5]))


# ----------------------------------------------------------------------------- G14 mask_path labels
def fixture_gen_path(PM):
    """generate_gen_path (process_map.py:148-163): every 5th label point -> 255 on a 224x224 'L' PNG."""
    from PIL import Image
    import torchvision
    class _ToPIL:                                            # ToPILImage on a float HxW array with values {0,255}
        def __call__(self, a):
            return Image.fromarray(np.asarray(a, dtype=np.float32), mode="F")
    torchvision.transforms.ToPILImage = _ToPIL
    g = np.load(os.path.join(OUT, "g10_config1_R64.npz"))
    pts = g["pathpoint"][:6] * (224 / 64)                    # config-1 label points scaled to the hard-coded 224 canvas
    tmp = tempfile.mkdtemp()
    with quiet():
        PM.generate_gen_path(list(pts), 0, root=tmp)
    masks = np.stack([np.asarray(Image.open(os.path.join(tmp, f"{i}.png"))) for i in range(len(pts))])
    np.savez_compressed(os.path.join(OUT, "g14_gen_path.npz"), pathpoint=pts, mask=masks.astype(np.uint8))


# ----------------------------------------------------------------------------- G15 obstacle raster (A13)
def fixture_plot_obstacles(PathMod):
    """The reference's obstacle raster, Path.plot_obstacles (Path.py:36-49): matplotlib circles on the default 6.4 x 4.8 in
    figure at dpi 90 -> ./map.jpg -> PIL '1' -> 'RGB' -> ToTensor -> crop [53:383, 73:517] -> T.Resize(R, R), run here with
    the real matplotlib / libjpeg / Pillow and two stand-ins for torchvision: ToTensor (u8 HWC -> f32 CHW / 255) and Resize =
    torch's bilinear interpolate without antialiasing (what torchvision 0.12 does to a tensor).  Recorded: the obstacle
    lists of 20 config-1 maps (scaled to R = 64, 224, 256) and the binarised raster (obstacle = value < 0.5), bit-packed.
    This MEASURES the deviation of the build's explicit rule (oracle/edage_np.py disc_raster); it does not pin it: the
    stand-in Resize and the JPEG decoder make the result environment-dependent by a pixel at the rims."""
    import torchvision

    class _Resize:
        def __init__(self, size):
            self.size = tuple(size)

        def __call__(self, img):
            return torch.nn.functional.interpolate(img[None], size=self.size, mode="bilinear", align_corners=False)[0]
    torchvision.transforms.Resize = _Resize
    PathMod.T.Resize = _Resize
    g = np.load(os.path.join(OUT, "g10_config1_R64.npz"))
    off = np.concatenate([[0], np.cumsum(g["n_obs"])])
    out = {}
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.chdir(tmp)
    try:
        n = 0
        for R in (64, 224, 256):
            for m in range(0, 100, 15):                       # 7 maps per resolution
                obs = g["obstacles"][off[m]:off[m + 1]] * (R / 64.0)
                with quiet():
                    img = PathMod.plot_obstacles((R, R), [list(o) for o in obs], resolution=(R, R))
                occ = (img[0].numpy() < 0.5)
                out[f"c{n}_R"] = np.array([R])
                out[f"c{n}_obs"] = _f64(obs)
                out[f"c{n}_occ"] = np.packbits(occ)
                n += 1
        out["ncase"] = np.array([n])
    finally:
        os.chdir(cwd)
    np.savez_compressed(os.path.join(OUT, "g15_plot_obstacles.npz"), **out)


# ----------------------------------------------------------------------------- G13 AE-ViT
def fixture_aevit():
    """GenNet AEViT(1,1,R,24) (GenNet/networks/ae_vit.py:12-76, the model predict.py:46 builds) with seeded
    random weights in eval mode: weights (as plain arrays), {0,1} inputs and outputs at R = 64, 224, 256 (config 3,
    what bench.py runs) and 512 (config 5: down_time = 4, one stage deeper, ae_vit.py:23)."""
    gen = "/root/reference/GenNet"
    sys.path.insert(0, gen)
    from networks.ae_vit import AEViT
    out = {}
    for R, B in ((64, 3), (224, 2), (256, 2), (512, 1)):
        torch.manual_seed(100 + R)
        m = AEViT(1, 1, R, 24).eval()
        # make BatchNorm statistics non-trivial so eval-mode folding is exercised
        with torch.no_grad():
            for mod in m.modules():
                if isinstance(mod, torch.nn.BatchNorm2d):
                    mod.running_mean.uniform_(-0.2, 0.2)
                    mod.running_var.uniform_(0.5, 1.5)
                    mod.weight.uniform_(0.8, 1.2)
                    mod.bias.uniform_(-0.1, 0.1)
        x = (torch.rand(B, 1, R, R) > 0.5).float()            # my_dataset.py:15: values {0,1}
        with torch.no_grad():
            y = m(x)
        for k, v in m.state_dict().items():
            out[f"R{R}/w/{k}"] = v.numpy()
        out[f"R{R}/x"] = x.numpy().astype(np.uint8)
        out[f"R{R}/y"] = y.numpy()
    np.savez_compressed(os.path.join(OUT, "g13_aevit.npz"), **out)
    sys.path.remove(gen)



# ----------------------------------------------------------------------------- G16 NAT / DiNAT wiring (B1, B2, B5's resize)
def _install_segnet_stubs():
    """SegNet/nat.py imports four third-party names (nat.py:10-14); none of them does arithmetic on the recorded path except the
    attention op:
      timm.models.layers.DropPath          -> identity module (eval mode / drop_path 0: what inference runs)
      mmcv.runner.load_checkpoint          -> no-op (pretrained=None)
      mmseg.utils.get_root_logger          -> no-op
      mmseg.models.builder.BACKBONES       -> a registry whose register_module() returns the class unchanged
      natten.NeighborhoodAttention2D       -> a torch module with NATTEN's parameters (qkv, rpb, proj) whose forward is the build's
                                              statement of the op (oracle/segnet_ref.py na_fp64 = oracle/na_np.py as a gather).
    So everything AROUND the attention — tokenizer, downsampler, Mlp, NATLayer with LayerScale, NATBlock, NAT.forward_tokens, the
    output norms and permutes — is the reference's own code, and the attention op itself stays "parity unpinned" (NATTEN's source is
    not in /root/reference)."""
    import torch.nn as nn
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))          # repo root: oracle/
    from oracle import segnet_ref as SR

    class DropPath(nn.Module):
        def __init__(self, drop_prob=0.0):
            super().__init__()
            assert not drop_prob or True

        def forward(self, x):
            assert not self.training
            return x

    class NeighborhoodAttention2D(nn.Module):
        def __init__(self, dim, kernel_size, dilation=None, num_heads=1, qkv_bias=True, qk_scale=None, attn_drop=0.0, proj_drop=0.0):
            super().__init__()
            assert qk_scale is None and attn_drop == 0.0 and proj_drop == 0.0
            self.num_heads, self.kernel_size, self.dilation = num_heads, kernel_size, dilation or 1
            self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
            self.rpb = nn.Parameter(torch.zeros(num_heads, 2 * kernel_size - 1, 2 * kernel_size - 1))
            self.proj = nn.Linear(dim, dim)

        def forward(self, x):
            return SR.na_fp64(x, self.qkv.weight, self.qkv.bias, self.rpb, self.proj.weight, self.proj.bias, self.num_heads,
                              self.kernel_size, self.dilation)

    class _Registry:
        def register_module(self, *a, **k):
            return lambda cls: cls

    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m
    mod("timm"); mod("timm.models"); mod("timm.models.layers", DropPath=DropPath)
    mod("mmcv"); mod("mmcv.runner", load_checkpoint=lambda *a, **k: None)
    mod("mmseg"); mod("mmseg.utils", get_root_logger=lambda *a, **k: None)
    mod("mmseg.models"); mod("mmseg.models.builder", BACKBONES=_Registry())
    mod("natten", NeighborhoodAttention2D=NeighborhoodAttention2D)


def fixture_nat_wiring():
    """SegNet/nat.py (NAT, :212-332) and SegNet/dinat.py (DiNAT) imported UNMODIFIED with the stubs above, in float64, eval mode:
    two small networks with dilated and padded levels — "dinat": embed 32, depths [2,2,2,1], heads [1,2,4,8] (head dim 32 as on
    every DiNAT-B level), dilations [[1,3],[1,2],[1,2],[1]], LayerScale on, on a 96 x 128 input (levels 24x32, 12x16 — its d = 2
    layer is padded to 14 rows —, 6x8 and 3x4: padded to 7 / 14); "nat": no dilations, no LayerScale (the `if not self.layer_scale`
    branch of NATLayer.forward), 3 levels, out_indices (0, 2), 64 x 64 input.  Recorded: constructor arguments, state-dict key
    order / shapes / checksums (the weights come from tests/_oracle_util.py wiring_weights), the input, every output level.
    Also SegNet/mmseg/ops/wrappers.py (torch only, loaded by path): `resize` and `Upsample` as the heads and the segmentor call
    them (setr_up_head.py:62-66: Upsample(scale_factor=2, mode='bilinear', align_corners=False); encoder_decoder.py:74-78: resize(size=
    img.shape[2:], mode='bilinear', align_corners=False))."""
    import importlib.util
    _install_segnet_stubs()
    seg = "/root/reference/SegNet"
    sys.path.insert(0, seg)
    sys.path.insert(0, os.path.dirname(OUT))                              # tests/: _oracle_util
    import nat as NATMOD
    import dinat as DINATMOD
    from _oracle_util import wiring_weights
    cases = {
        "dinat": (DINATMOD.DiNAT, dict(embed_dim=32, mlp_ratio=2.0, depths=[2, 2, 2, 1], num_heads=[1, 2, 4, 8], drop_path_rate=0.2,
                                       kernel_size=7, dilations=[[1, 3], [1, 2], [1, 2], [1]], out_indices=(0, 1, 2, 3), layer_scale=1e-5),
                  (2, 3, 96, 128), 1),
        "nat": (NATMOD.NAT, dict(embed_dim=32, mlp_ratio=3.0, depths=[1, 2, 1], num_heads=[1, 2, 4], drop_path_rate=0.0, kernel_size=7,
                                 dilations=None, out_indices=(0, 2), layer_scale=None), (1, 3, 64, 64), 2),
    }
    out = {"cases": np.array(sorted(cases))}
    for name, (cls, cfg, xshape, seed) in cases.items():
        m = cls(**cfg).double()
        m.eval()                                                          # (NAT.train returns None, nat.py:291-293: no chaining)
        assert not m.training
        sd = m.state_dict()
        keys = list(sd.keys())
        shapes = [tuple(v.shape) for v in sd.values()]
        w = wiring_weights(keys, shapes, seed)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
        x = torch.from_numpy(np.random.RandomState(1000 + seed).normal(0.0, 1.0, xshape))
        with torch.no_grad():
            ys = m(x)
        out[f"{name}/cfg"] = np.array(json.dumps(cfg))
        out[f"{name}/seed"] = np.array([seed])
        out[f"{name}/keys"] = np.array(keys)
        out[f"{name}/shapes"] = np.array([json.dumps(s) for s in shapes])
        out[f"{name}/checksum"] = np.array([[w[k].sum(), (w[k] ** 2).sum()] for k in keys])
        out[f"{name}/x"] = x.numpy().astype(np.float32)                  # float32-representable input: the GPU test feeds the same values
        x32 = torch.from_numpy(out[f"{name}/x"]).double()
        with torch.no_grad():
            ys = m(x32)
        for i, y in zip(cfg["out_indices"], ys):
            out[f"{name}/y{i}"] = y.numpy()
    spec = importlib.util.spec_from_file_location("_ref_wrappers", os.path.join(seg, "mmseg", "ops", "wrappers.py"))
    W = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(W)
    rs = np.random.RandomState(16)
    a = torch.from_numpy(rs.normal(0, 1, (2, 5, 6, 9)).astype(np.float32)).double()
    out["resize/in"] = a.numpy()
    with torch.no_grad():
        out["resize/upsample2x"] = W.Upsample(scale_factor=2, mode="bilinear", align_corners=False)(a).numpy()
        out["resize/to_24x36"] = W.resize(a, size=(24, 36), mode="bilinear", align_corners=False).numpy()
        out["resize/to_13x7"] = W.resize(a, size=(13, 7), mode="bilinear", align_corners=False).numpy()
        out["resize/to_11x17_ac"] = W.resize(a, size=(11, 17), mode="bilinear", align_corners=True, warning=False).numpy()
    np.savez_compressed(os.path.join(OUT, "g16_nat_wiring.npz"), **out)
    sys.path.remove(seg)


def main():
    _install_stubs()
    if len(sys.argv) > 1 and sys.argv[1] == "nat_wiring":
        fixture_nat_wiring()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "aevit":
        fixture_aevit()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "plot_obstacles":
        sys.path.insert(0, REF)
        import Path as PathMod
        fixture_plot_obstacles(PathMod)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "gen_path":
        sys.path.insert(0, REF)
        import process_map as PM
        fixture_gen_path(PM)
        return
    sys.path.insert(0, REF)
    import PathSeg as PathSegMod
    import Path as PathMod
    fixture_pathseg(PathSegMod)
    fixture_paths(PathMod)
    fixture_boundary_check(PathMod)
    import MapGenerate as MapGenMod
    try:
        with time_limit(1500):
            fixture_mapgenerate(MapGenMod)
    except RefHang:
        print("config-1 fixture: reference did not terminate within 1500 s (seed 0)", flush=True)
    import process_map as PM
    fixture_plan_tail(PM)
    fixture_gen_path(PM)
    fixture_plot_obstacles(PathMod)
    sys.path.remove(REF)
    for mod in ("utils",):
        sys.modules.pop(mod, None)
    fixture_aevit()
    fixture_nat_wiring()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
