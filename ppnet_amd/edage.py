"""Batched EDaGe-PP on MI355X: device buffers (torch) + calls through the C ABI.

`generate_paths` = PathGroup.generate for n independent paths (reference EDaGe-PP/PathGenerate.py:33-50);
`generate_maps`  = the body of MapGenerate.generate for n_paths x placements maps (MapGenerate.py:48-124).
All results stay resident in HBM as torch tensors; nothing here computes on the CPU.
"""
import ctypes as C

import torch

from . import _lib as L


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _check_R(R):
    if R % 32 != 0 or not (32 <= R <= 512):
        raise ValueError(f"resolution must be a multiple of 32 in [32, 512], got {R}")


class PathsBatch:
    """Device-resident results of stage A (one row per target path); field names follow ppn_paths_t."""

    def __init__(self, n, R, map_size, clearance, device, debug=False):
        _check_R(R)
        self.n, self.R, self.map_size, self.clearance, self.device = n, R, float(map_size), float(clearance), device
        f64 = dict(dtype=torch.float64, device=device)
        i32 = dict(dtype=torch.int32, device=device)
        z = torch.zeros
        self.seg_poly = z(n, 10, 5, **f64)
        self.seg_endpoint = z(n, 10, **f64)
        self.seg_rotation = z(n, 10, **f64)
        self.seg_translation = z(n, 10, 2, **f64)
        self.seg_length = z(n, 10, **f64)
        self.seg_straight = z(n, 10, **i32)
        self.segpoint_world = z(n, 11, 2, **f64)
        self.pathpoint_world = z(n, L.PATH_POINTS, 2, **f64)
        self.boundary_world = z(n, L.BOUNDARY_POINTS, 2, **f64) if debug else None
        self.canvas_bits = z(n, (2 * R) * (2 * R) // 32, **i32) if debug else None
        self.hull_raw = z(n, L.MAX_HULL, 2, **f64) if debug else None
        self.hull = z(n, L.MAX_HULL, 2, **f64)
        self.hull_n = z(n, **i32)
        self.rotation = z(n, **f64)
        self.trans_rc = z(n, 2, **f64)
        self.segpoint_image = z(n, 11, 2, **f64)
        self.pathpoint_image = z(n, L.PATH_POINTS, 2, **f64)
        self.space_bits = z(n, R * R // 32, **i32)
        self.isles = z(n, L.MAX_ISLES, 2, **i32)
        self.n_isles = z(n, **i32)
        self.obstacles = z(n, L.MAX_POCKET, 3, **f64)
        self.n_obstacles = z(n, **i32)
        self.length = z(n, **f64)
        self.straight = z(n, **i32)
        self.flags = z(n, **i32)
        self.max_step_px = z(n, **f64)
        self.seg_grad = z(n, 10, 2, **f64)
        self.pocket_draws_used = z(n, **i32)
        self.struct = L.PathsStruct(**{name: _ptr(getattr(self, name)) for name, _ in L.PathsStruct._fields_})

    def view(self, first, n):
        """Paths [first, first + n) of this batch as a PathsBatch of their own (no copies: narrowed tensors, a new pointer
        struct) — stage A can be launched for several stage-B batches at once and consumed batch by batch."""
        v = object.__new__(PathsBatch)
        v.n, v.R, v.map_size, v.clearance, v.device = n, self.R, self.map_size, self.clearance, self.device
        for name, _ in L.PathsStruct._fields_:
            t = getattr(self, name)
            setattr(v, name, t.narrow(0, first, n) if t is not None else None)
        v.struct = L.PathsStruct(**{name: _ptr(getattr(v, name)) for name, _ in L.PathsStruct._fields_})
        return v

    def space_mask(self):
        """Path.Space as a bool tensor [n, R, R] (unpacked on the device)."""
        return _unpack_bits(self.space_bits, self.R, self.R)

    def canvas_mask(self):
        return _unpack_bits(self.canvas_bits, 2 * self.R, 2 * self.R)


def _unpack_bits(words, h, w):
    shifts = torch.arange(32, device=words.device, dtype=torch.int32)
    bits = (words.unsqueeze(-1) >> shifts) & 1
    return bits.reshape(words.shape[0], h, w).bool()


class MapsBatch:
    """Device-resident results of stage B (one row per map); field names follow ppn_maps_t."""

    def __init__(self, n, R, K, device, want_pathpoint=True, want_accept=True):
        self.n, self.R, self.K, self.device = n, R, K, device
        f64 = dict(dtype=torch.float64, device=device)
        i32 = dict(dtype=torch.int32, device=device)
        e = torch.empty
        self.grid = e(n, R, R, dtype=torch.uint8, device=device)
        self.angle = e(n, **f64)
        self.translation = e(n, 2, **i32)
        self.attempts = e(n, **i32)
        self.segpoint = e(n, 11, 2, **f64)
        self.pathpoint = e(n, L.PATH_POINTS, 2, **f64) if want_pathpoint else None
        self.accept = e(n, max(K, 1), dtype=torch.uint8, device=device) if want_accept else None
        self.obstacles = torch.zeros(n, K + L.MAX_POCKET, 3, **f64)
        self.n_obstacles = e(n, 2, **i32)
        self.flags = e(n, **i32)
        self.records = e(n, 26, **f64)             # shard.RECORD_WIDTH: angle, flags, translation, segpoint (all-gather unit)
        self.struct = L.MapsStruct(**{name: _ptr(getattr(self, name)) for name, _ in L.MapsStruct._fields_})

    def with_records(self, records):
        """This batch with its all-gather records written somewhere else: `records` is a contiguous [n, 26] float64 device
        tensor, typically one step's slice of a staging ring that holds the records of several steps until ONE collective
        ships them (shard.RecordRing).  No copies: the same tensors, a new pointer struct."""
        assert tuple(records.shape) == (self.n, 26) and records.dtype == torch.float64 and records.is_contiguous()
        v = object.__new__(MapsBatch)
        v.__dict__.update(self.__dict__)
        v.records = records
        v.struct = L.MapsStruct(**{name: _ptr(getattr(v, name)) for name, _ in L.MapsStruct._fields_})
        return v


def generate_paths(n_paths, resolution=224, map_size=50, clearance=1, seed=0, first_path_id=0, device="cuda:0",
                   draws=None, pocket_draws=None, debug=False, out=None, force_straight=None, hull_start=None):
    """Stage A for `n_paths` paths. draws / pocket_draws: optional device tensors that replace the
    Philox streams ([n, 10021] float64 in the fixed layout; [n, stride] float32 in torch.rand order).
    hull_start: optional int32 [n], the hull's first vertex as an index into the canonical cycle (-1 = canonical): the
    reference's Qhull order, for callers that replay its torch.rand stream isle by isle (ppn_edage_paths_ex2)."""
    device = torch.device(device)
    pb = out if out is not None else PathsBatch(n_paths, resolution, map_size, clearance, device, debug=debug)
    stride = 0
    if draws is not None:
        assert draws.dtype == torch.float64 and draws.is_contiguous() and tuple(draws.shape) == (n_paths, L.DRAWS_PER_PATH)
    if pocket_draws is not None:
        assert pocket_draws.dtype == torch.float32 and pocket_draws.is_contiguous() and pocket_draws.shape[0] == n_paths
        stride = pocket_draws.shape[1]
    if force_straight is not None:
        assert force_straight.dtype == torch.int8 and force_straight.is_contiguous() and force_straight.shape[0] == n_paths
    if hull_start is not None:
        assert hull_start.dtype == torch.int32 and hull_start.is_contiguous() and hull_start.shape[0] == n_paths and hull_start.is_cuda
    with torch.cuda.device(device):
        rc = L.lib.ppn_edage_paths_ex2(n_paths, first_path_id, resolution, float(map_size), float(clearance), seed,
                                       _ptr(draws), _ptr(pocket_draws), stride, _ptr(force_straight), _ptr(hull_start),
                                       C.byref(pb.struct), _stream_ptr(device))
    L.check(rc, "ppn_edage_paths")
    return pb


def generate_maps(paths, placements, obstacles_size=5, obstacles_num=50, seed=0, first_map_id=0, place_draws=None,
                  obst_draws=None, want_pathpoint=True, want_accept=True, out=None, phase="both"):
    """Stage B: `placements` maps for each target path in `paths` (a PathsBatch).

    phase="place" runs only the placement / label / filter kernel (everything but `grid`), phase="raster" only the
    grid kernel on a MapsBatch a "place" call filled (pass it as `out`); "both" is the two back to back."""
    assert phase in ("both", "place", "raster")
    n = paths.n * placements
    K = int(obstacles_num)
    mb = out if out is not None else MapsBatch(n, paths.R, K, paths.device, want_pathpoint, want_accept)
    if place_draws is not None:
        assert place_draws.dtype == torch.float64 and place_draws.is_contiguous() and tuple(place_draws.shape) == (n, 3)
    if obst_draws is not None:
        assert obst_draws.dtype == torch.float64 and obst_draws.is_contiguous() and tuple(obst_draws.shape) == (n, 3 * K)
    with torch.cuda.device(paths.device):
        if phase == "raster":
            assert out is not None, "phase='raster' needs the MapsBatch of the matching 'place' call"
            rc = L.lib.ppn_edage_maps_raster(C.byref(paths.struct), paths.n, placements, paths.R, K,
                                             C.byref(mb.struct), _stream_ptr(paths.device))
        else:
            fn = L.lib.ppn_edage_maps if phase == "both" else L.lib.ppn_edage_maps_place
            rc = fn(C.byref(paths.struct), paths.n, placements, first_map_id, paths.R, paths.map_size,
                    float(obstacles_size), K, paths.clearance, seed, _ptr(place_draws), _ptr(obst_draws),
                    C.byref(mb.struct), _stream_ptr(paths.device))
    L.check(rc, "ppn_edage_maps" if phase == "both" else "ppn_edage_maps_" + phase)
    return mb


def label_masks(paths, maps, placements, bound=None, want_path=True, want_space=True):
    """generate_gen_path / generate_seg_space (process_map.py:148-191) for every map of a MapsBatch:
    returns (mask_path u8 [n,R,R] with 255 on every 5th label point, mask_space u8 [n,R,R] in {0,1})."""
    R, n, dev = paths.R, maps.n, paths.device
    mp = torch.empty(n, R, R, dtype=torch.uint8, device=dev) if want_path else None
    ms = torch.empty(n, R, R, dtype=torch.uint8, device=dev) if want_space else None
    with torch.cuda.device(dev):
        rc = L.lib.ppn_label_masks(C.byref(paths.struct), C.byref(maps.struct), paths.n, placements, R,
                                   R if bound is None else int(bound), _ptr(mp), _ptr(ms), _stream_ptr(dev))
    L.check(rc, "ppn_label_masks")
    return mp, ms


def boundary_check(hull, angle_deg, translation_rc, resolution, return_hull=False):
    """Path.boundary_check for n (angle, translation) pairs against one hull [h,2] (device tensors)."""
    n = angle_deg.shape[0]
    ok = torch.empty(n, dtype=torch.uint8, device=hull.device)
    hull = hull.contiguous()
    hull_out = torch.empty(n, hull.shape[0], 2, dtype=torch.float64, device=hull.device) if return_hull else None
    with torch.cuda.device(hull.device):
        rc = L.lib.ppn_boundary_check_ex(_ptr(hull), hull.shape[0], _ptr(angle_deg.contiguous()),
                                         _ptr(translation_rc.contiguous()), n, resolution, _ptr(ok), _ptr(hull_out),
                                         _stream_ptr(hull.device))
    L.check(rc, "ppn_boundary_check")
    return (ok.bool(), hull_out) if return_hull else ok.bool()


def obstacle_filter(pathpoint, draws, K, resolution, map_size, obstacles_size, clearance):
    """generate_map_randomly's accept loop (MapGenerate.py:128-143): pathpoint [n,1000,2], draws [n,3K] (device f64).
    Returns (accept [n,K] bool, obstacles [n,K,3] as [col,row,r], counts [n])."""
    n = pathpoint.shape[0]
    dev = pathpoint.device
    accept = torch.empty(n, K, dtype=torch.uint8, device=dev)
    obstacles = torch.zeros(n, K, 3, dtype=torch.float64, device=dev)
    counts = torch.empty(n, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = L.lib.ppn_obstacle_filter(_ptr(pathpoint.contiguous()), _ptr(draws.contiguous()), n, K, resolution,
                                       float(map_size), float(obstacles_size), float(clearance), _ptr(accept),
                                       _ptr(obstacles), _ptr(counts), _stream_ptr(dev))
    L.check(rc, "ppn_obstacle_filter")
    return accept.bool(), obstacles, counts


def paint_markers(grid, init, end):
    """add_init_end_single on u8 grids [n,R,R] in place; init / end [n,2] f64 (row, col)."""
    with torch.cuda.device(grid.device):
        rc = L.lib.ppn_paint_markers(_ptr(grid), grid.shape[0], grid.shape[1], _ptr(init.contiguous()),
                                     _ptr(end.contiguous()), _stream_ptr(grid.device))
    L.check(rc, "ppn_paint_markers")
    return grid


def grid_to_rgb(grid):
    """u8 codes [n,R,R] -> float image [n,3,R,R] in [0,1] as the reference saves it: free white, obstacle black,
    start/goal red (the [255,0,0] paint saturates to (1,0,0), process_map.py:120,128)."""
    free = (grid == L.GRID_FREE).to(torch.float32)
    mark = (grid == L.GRID_MARK).to(torch.float32)
    return torch.stack([free + mark, free, free], dim=1)


def disc_raster(obstacles, counts, resolution):
    """Explicit obstacle raster (stands in for Path.plot_obstacles): obstacles [n, stride, 3] f64, counts [n] i32."""
    n, stride = obstacles.shape[0], obstacles.shape[1]
    grid = torch.empty(n, resolution, resolution, dtype=torch.uint8, device=obstacles.device)
    with torch.cuda.device(obstacles.device):
        rc = L.lib.ppn_disc_raster(_ptr(obstacles.contiguous()), _ptr(counts.contiguous()), stride, n, resolution,
                                   _ptr(grid), _stream_ptr(obstacles.device))
    L.check(rc, "ppn_disc_raster")
    return grid
