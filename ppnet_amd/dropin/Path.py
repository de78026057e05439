"""Path with the reference's interface (EDaGe-PP/Path.py:52-537), backed by the stage-A kernel.

generate() / draw_boundary() / path_obstacles() keep their order and meaning; the kernel computes all
three in one launch, so generate() launches at a provisional resolution for the world-frame attributes
and path_obstacles() re-launches with the caller's resolution on the same draws.
"""
import numpy as np
import torch

from ppnet_amd import _lib as L
from ppnet_amd import edage, rng
from PathSeg import PathSeg

DIM = 2


class BoundaryOneSide:
    def __init__(self):
        self.point = []
        self.direction = []


class Boundary:
    def __init__(self):
        self.upboundary = BoundaryOneSide()
        self.downboundary = BoundaryOneSide()
        self.initboundary = []
        self.endboundary = []


def plot_obstacles(size: tuple, obstacles, resolution: tuple = (224, 224)):
    """Obstacle raster (Path.py:36-49) by the explicit rule of ppn_disc_raster — the data point a pixel centre shows in the
    reference's cropped matplotlib figure lies inside the ellipse the stroked circle inks: returns a
    float tensor [3,R,R], 1 = free, 0 = obstacle, like the reference's ToTensor output."""
    R = int(resolution[0])
    dev = torch.device(rng.device())
    obs = torch.zeros(1, max(len(obstacles), 1), 3, dtype=torch.float64, device=dev)
    if len(obstacles):
        obs[0, :len(obstacles)] = torch.tensor([[float(o[0]), float(o[1]), float(o[2])] for o in obstacles],
                                               dtype=torch.float64, device=dev) * (R / float(size[0]))
    cnt = torch.tensor([len(obstacles)], dtype=torch.int32, device=dev)
    g = edage.disc_raster(obs, cnt, R)[0]
    return (g == L.GRID_FREE).to(torch.float32).unsqueeze(0).expand(3, R, R).contiguous()


def _qhull_start(pb, R, map_size):
    """Index, in the stage-A kernel's canonical hull cycle (pb.hull_raw[0]: lexicographically smallest lattice point first,
    counter-clockwise), of the vertex scipy.spatial.ConvexHull lists first for the same lattice points — Path.convexhull's
    input and call (Path.py:388-395).  0 when Qhull's vertex set is not the kernel's (never seen; the canonical order then stays)."""
    from scipy.spatial import ConvexHull
    world = pb.pathpoint_world[0].cpu().numpy()
    pts = np.round(world / (map_size / R) + R).astype(np.int64)               # coord_euclidean2image, Path.py:378-386
    hn = int(pb.hull_n[0])
    canon = np.rint(pb.hull_raw[0, :hn].cpu().numpy()).astype(np.int64)
    try:
        first = pts[ConvexHull(pts).vertices[0]]
    except Exception:                                                          # degenerate input: Qhull raises, so does the reference
        return 0
    k = np.where((canon == first).all(axis=1))[0]
    return int(k[0]) if len(k) == 1 else 0


class Path:
    def __init__(self, seg_num=3, poly_order=3, dim=2, clearance=1, is_straight=True):
        self.device = torch.device(rng.device())
        self.PathSeg = []
        self.SegPoint = [[0, 0]]
        self.PathPoint = []
        self.SegPointImage = []
        self.obstacles = []
        self.SegNum = seg_num
        self.PolyOrder = poly_order
        self.Dim = dim
        self.Boundary = Boundary()
        self.BoundaryPoint = []
        self.Clearance = clearance
        self.EndPoint = np.array([0, 0])
        self.Translation = np.array([0, 0])
        self.Rotation = 0
        self.Space = torch.zeros([1])
        self.PathObs = torch.zeros([1])
        self.Resolution = 0
        self.MapSize = 0
        self.MapOffset = 0
        self.ConvexHull = []
        self.is_straight = is_straight
        self.Length = 0
        self.flags = 0
        self._draws = None
        self._pocket = None
        self._path_id = None
        self._torch_state = None

    # ------------------------------------------------------------------ draws
    def _prepare_draws(self):
        if self._path_id is not None:
            return
        self._path_id = rng.take_path_ids(1)
        if rng.mode() == "mt19937":
            d = np.ones(L.DRAWS_PER_PATH)
            d[0] = 0.0 if self.is_straight else 1.0
            for s in range(L.SEGS):
                b = 1 + s * L.DRAWS_PER_SEG
                if not self.is_straight:                       # PathSeg.py:19 short-circuit
                    d[b] = np.random.random(1)[0]
                d[b + 1:b + 1001] = np.random.random(1000)     # PathSeg.py:23
                d[b + 1001] = np.random.random(1)[0]           # PathSeg.py:32
            self._draws = torch.tensor(d[None, :], dtype=torch.float64, device=self.device)

    def _launch(self, R, map_size, with_pocket):
        mt_pocket = rng.mode() == "mt19937" and with_pocket
        if mt_pocket:
            # space_normalization's RandomRotation draws its angle from the global torch generator (torchvision 0.12:
            # `torch.empty(1).uniform_(a, a)`, one draw, Path.py:160-161) before set_obstacles' torch.rand
            torch.rand(1)
            # torch.rand is then consumed a data-dependent number of times (Path.py:479-485): pre-draw, then
            # rewind and advance the global generator by what the kernel actually used
            self._torch_state = torch.get_rng_state()
        force = torch.tensor([1 if self.is_straight else 0], dtype=torch.int8, device=self.device)
        n = 3 * L.POCKET_TRY_CAP
        hull_start = None          # replay mode: Qhull's first vertex as an index into the kernel's canonical cycle, once known
        while True:
            pocket = None
            if mt_pocket:
                torch.set_rng_state(self._torch_state)
                pocket = torch.tensor([[torch.rand(1).item() for _ in range(n)]], dtype=torch.float32, device=self.device)
            pb = edage.generate_paths(1, R, map_size, self.Clearance, seed=rng.seed(), first_path_id=self._path_id,
                                      device=self.device, draws=self._draws, pocket_draws=pocket, debug=True,
                                      force_straight=force, hull_start=hull_start)
            torch.cuda.synchronize(self.device)
            if pocket is None:
                return pb
            if hull_start is None and not self.is_straight:
                # set_obstacles consumes torch.rand isle by isle in the order of Qhull's vertex list (Path.py:388-395,463-537)
                hs = _qhull_start(pb, R, map_size)
                hull_start = torch.tensor([hs], dtype=torch.int32, device=self.device)
                if hs > 0:
                    continue                                     # same draws, the reference's vertex order
            used = int(pb.pocket_draws_used[0])
            if used <= n and not (int(pb.flags[0]) & L.FLAG_POCKET_DRAWS):
                break
            if n >= 3 * L.POCKET_TRY_CAP * L.MAX_ISLES:           # cannot happen: the kernel's own caps bound the draws
                raise RuntimeError("set_obstacles consumed more torch.rand draws than the kernel's caps allow")
            n = 3 * L.POCKET_TRY_CAP * L.MAX_ISLES                # every isle at its try cap: the most the kernel can use
        torch.set_rng_state(self._torch_state)
        for _ in range(used):
            torch.rand(1)
        return pb

    # ------------------------------------------------------------------ reference API
    def generate(self, show_now=True, polys=None):
        if polys is not None:
            raise NotImplementedError("explicit polynomials are not part of the accelerated path")
        if self.SegNum != L.SEGS or self.PolyOrder != 4 or self.Dim != 2:
            raise NotImplementedError("the MI355X path implements the generator's geometry: seg_num=10, poly_order=4, dim=2")
        self._prepare_draws()
        pb = self._launch(224, 50, with_pocket=False)
        self._fill_world(pb)

    def _fill_world(self, pb):
        self.PathSeg = []
        for i in range(L.SEGS):
            s = PathSeg(4, 2, is_straight=bool(pb.seg_straight[0, i]))
            s.Poly = pb.seg_poly[0, i].cpu().numpy()
            s.EndPoint = np.array([float(pb.seg_endpoint[0, i])])
            s.Length = float(pb.seg_length[0, i])
            s.Translation = pb.seg_translation[0, i].cpu().numpy()
            s.Rotation = float(pb.seg_rotation[0, i])
            s.GradSt, s.GradEnd = (float(v) for v in pb.seg_grad[0, i])
            self.PathSeg.append(s)
        self.SegPoint = pb.segpoint_world[0].cpu().numpy()
        self.PathPoint = pb.pathpoint_world[0].cpu().numpy()
        self.Length = float(pb.length[0])
        self.EndPoint = self.SegPoint[-1]
        self.is_straight = bool(pb.straight[0])
        self._world_boundary = pb.boundary_world[0].cpu().numpy()

    def draw_boundary(self, show_now=True):
        b = self._world_boundary                                 # init reversed | up | end | down reversed
        self.BoundaryPoint = b
        self.Boundary.initboundary = list(b[0:50][::-1])
        self.Boundary.upboundary.point = list(b[50:550].reshape(10, 50, 2))
        self.Boundary.endboundary = list(b[550:600])
        self.Boundary.downboundary.point = list(b[600:1100][::-1].reshape(10, 50, 2))

    def path_obstacles(self, resolution=224, map_size=50, map_offset=112):
        self.Resolution, self.MapSize, self.MapOffset = resolution, map_size, map_offset
        self._prepare_draws()
        pb = self._launch(resolution, map_size, with_pocket=True)
        self._fill_world(pb)
        self._fill_image(pb, 0)
        return True

    def _fill_image(self, pb, j):
        R = pb.R
        self.Resolution, self.MapSize, self.MapOffset = R, pb.map_size, R / 2
        hn = int(pb.hull_n[j])
        self.ConvexHull = pb.hull[j, :hn].cpu()
        self.Rotation = float(pb.rotation[j])
        t = pb.trans_rc[j].cpu()
        self.Translation = [t[1], t[0]]                          # [t_col, t_row], Path.py:171
        self.SegPointImage = pb.segpoint_image[j].cpu().numpy()
        self.PathPoint = pb.pathpoint_image[j].cpu()
        self.Space = pb.space_mask()[j].to(torch.float32).unsqueeze(0).expand(3, R, R).contiguous()
        no = int(pb.n_obstacles[j])
        self.obstacles = [[float(o[0]), float(o[1]), float(o[2])] for o in pb.obstacles[j, :no].cpu()]
        self.PathObs = torch.ones([3, R, R]) if self.is_straight else self.PathObs
        self.flags = int(pb.flags[j])
        self._pb, self._j = pb, j

    def boundary_check(self, angle, translation):
        hull = self.ConvexHull.to(self.device).contiguous()
        a = torch.tensor(np.ravel(np.asarray(angle, dtype=np.float64))[:1], device=self.device)
        t = torch.tensor([[float(translation[0]), float(translation[1])]], dtype=torch.float64, device=self.device)
        ok, hull_out = edage.boundary_check(hull, a, t, int(self.Resolution), return_hull=True)
        return bool(ok[0]), hull_out[0].cpu().numpy()

    @staticmethod
    def coord_rotation(x, radians):
        rotation = np.reshape([[np.cos(radians), -np.sin(radians)], [np.sin(radians), np.cos(radians)]], [2, 2])
        return np.dot(rotation, x)
