"""PathGroup with the reference's interface (EDaGe-PP/PathGenerate.py:19-50)."""
import numpy as np
import torch

from ppnet_amd import edage, rng
from Path import Path

ORDER = 4
PATHSEGNUM = 10
DIM = 2


class PathGroup:
    def __init__(self, path_num=5, resolution=224, map_size=50):
        self.device = torch.device(rng.device())
        self.Paths = []
        self.TargetPaths = []
        self.PathNum = path_num
        self.Resolution = resolution
        self.MapSize = map_size
        self.MapOffset = self.Resolution / 2
        self.SpacesObs = []
        self.SpacesFree = []
        self.Rotation = []
        self.Translation = []
        self.batch = None          # device-resident PathsBatch (philox mode): what MapGenerate launches from

    def generate(self, path_seg_num=3, poly_order=4, dim=2, clearance=1):
        if path_seg_num != PATHSEGNUM or poly_order != ORDER or dim != DIM:
            raise NotImplementedError("the MI355X path implements the generator's geometry: 10 segments, order 4, 2-D")
        if rng.mode() == "philox":
            # every path is independent: one launch for the whole group
            first = rng.take_path_ids(self.PathNum)
            pb = edage.generate_paths(self.PathNum, self.Resolution, self.MapSize, clearance, seed=rng.seed(),
                                      first_path_id=first, device=self.device, debug=True)
            torch.cuda.synchronize(self.device)
            self.batch = pb
            for j in range(self.PathNum):
                p = Path(seg_num=path_seg_num, poly_order=poly_order, dim=dim, clearance=clearance,
                         is_straight=bool(pb.straight[j]))
                p._path_id = first + j
                p._fill_world(_row(pb, j))
                p.draw_boundary(show_now=False)
                p._fill_image(pb, j)
                self.Paths.append(p)
                self.TargetPaths.append(p)
        else:
            i = 0
            while i < self.PathNum:                       # PathGenerate.py:35-44, the reference's stream order
                is_straight = False if np.random.random(1) > 0.01 else True
                path = Path(seg_num=path_seg_num, poly_order=poly_order, dim=dim, clearance=clearance, is_straight=is_straight)
                path._prepare_draws()
                rst = path.path_obstacles(resolution=self.Resolution, map_size=self.MapSize, map_offset=self.MapOffset)
                path.draw_boundary(show_now=False)
                print('path generate result:', rst, "Length:", path.Length)
                if rst:
                    i = i + 1
                    self.Paths.append(path)
                    self.TargetPaths.append(path)
        if not np.size(self.TargetPaths):
            return False
        print('target path list length:', len(self.TargetPaths))
        return True


class _row:
    """View of row j of a PathsBatch that looks like a one-row batch to Path._fill_world."""

    def __init__(self, pb, j):
        self._pb, self._j = pb, j

    def __getattr__(self, name):
        v = getattr(self._pb, name)
        return v[self._j:self._j + 1] if isinstance(v, torch.Tensor) else v
