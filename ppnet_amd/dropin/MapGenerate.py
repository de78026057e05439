"""MapGenerate with the reference's interface (EDaGe-PP/MapGenerate.py:27-151), backed by the stage-B kernel.

generate() produces path_num^2 maps per outer iteration exactly like the reference: occupancy image (JPEG),
MapLabel entry [label, angle, translation, segpoint, pathpoint] and the JSON problem line.
"""
import json
import os

import numpy as np
import torch

from ppnet_amd import _lib as L
from ppnet_amd import edage, rng
from PathGenerate import PathGroup

ORDER = 4
PATHSEGNUM = 10
DIM = 2
total_record = 10000
cnt = 0
SPECULATE = 64        # placement attempts evaluated per device call when replaying the MT19937 stream


class MapGenerate:
    def __init__(self, path_num=5, resolution=224, map_size=50, obstacles_size=5, obstacles_num=50, clearance=1):
        self.device = torch.device(rng.device())
        self.Resolution = resolution
        self.MapSize = map_size
        self.ObstaclesNum = obstacles_num
        self.ObstacleSize = obstacles_size
        self.Clearance = clearance
        self.MapData = []
        self.PathGroup = PathGroup(path_num=path_num, resolution=resolution, map_size=map_size)
        self.PathGroup.generate(path_seg_num=PATHSEGNUM, poly_order=ORDER, dim=2, clearance=clearance)
        self.MapLabel = []
        self.last_batch = None      # device-resident MapsBatch of the most recent generate() iteration

    # ------------------------------------------------------------------ batches
    def _paths_batch(self):
        pg = self.PathGroup
        if pg.batch is not None:
            return pg.batch
        P = len(pg.TargetPaths)                         # mt19937 mode: gather the per-path launches into one batch
        pb = edage.PathsBatch(P, self.Resolution, self.MapSize, self.Clearance, self.device)
        for j, tp in enumerate(pg.TargetPaths):
            src, k = tp._pb, tp._j
            for name in ("hull", "hull_n", "segpoint_image", "pathpoint_image", "space_bits", "obstacles", "n_obstacles",
                         "flags", "max_step_px", "length"):
                getattr(pb, name)[j] = getattr(src, name)[k]
        pg.batch = pb
        return pb

    def _mt_draws(self, pb, placements):
        """Replay MapGenerate.py:57-67,128-130 on the global numpy stream: rejection loop (3 draws per attempt,
        hull test on the device, SPECULATE attempts per call), then 3K obstacle draws — map after map."""
        R, K = self.Resolution, self.ObstaclesNum
        place, obst = [], []
        for j in range(pb.n):
            hull = pb.hull[j, :int(pb.hull_n[j])]
            for _ in range(placements):
                tries = 0
                while True:
                    st = np.random.get_state()
                    u = np.random.random(3 * SPECULATE).reshape(SPECULATE, 3)
                    ang = torch.tensor(-(u[:, 0] * 360 - 180), device=self.device)
                    t = np.array(u[:, 1:3] * R - R / 2, dtype=int)
                    tr = torch.tensor(t[:, ::-1].astype(np.float64).copy(), device=self.device)
                    ok = edage.boundary_check(hull, ang, tr, R).cpu().numpy()
                    np.random.set_state(st)
                    if ok.any():
                        a = int(np.argmax(ok))
                        np.random.random(3 * (a + 1))                 # consume up to and including the accepted attempt
                        place.append(u[a])
                        break
                    np.random.random(3 * SPECULATE)
                    tries += SPECULATE
                    if tries > 1000000:
                        raise RuntimeError('Error:Repeated over 1000000 times!')
                torch.rand(1)       # MapGenerate.py:103-104: RandomRotation's angle draw (torchvision 0.12), one per placed map
                obst.append(np.concatenate([np.random.random(K), np.random.random(K), np.random.random(K)]))
        return (torch.tensor(np.array(place), device=self.device), torch.tensor(np.array(obst), device=self.device))

    # ------------------------------------------------------------------ reference API
    def generate(self, map_num=100, folder_path='./', round_index=0, save_images=True):
        global cnt
        P = len(self.PathGroup.TargetPaths)
        print('generating {} maps by using {} target paths'.format(map_num, P))
        pb = self._paths_batch()
        for i in range(int(np.round(map_num / P ** 2))):
            if save_images:
                os.makedirs(os.path.join(folder_path, 'data'), exist_ok=True)
                os.makedirs(os.path.join(folder_path, 'GMM'), exist_ok=True)
            if rng.mode() == "philox":
                mb = edage.generate_maps(pb, P, self.ObstacleSize, self.ObstaclesNum, seed=rng.seed(),
                                         first_map_id=rng.take_map_ids(P * P))
            else:
                place, obst = self._mt_draws(pb, P)
                mb = edage.generate_maps(pb, P, self.ObstacleSize, self.ObstaclesNum, place_draws=place, obst_draws=obst)
            torch.cuda.synchronize(self.device)
            self.last_batch = mb
            angle, trans = mb.angle.cpu().numpy(), mb.translation.cpu().numpy()
            segpoint, pathpoint = mb.segpoint.cpu().numpy(), mb.pathpoint.cpu().numpy()
            n_obs, obstacles = mb.n_obstacles.cpu().numpy(), mb.obstacles.cpu().numpy()
            rgb = edage.grid_to_rgb(mb.grid) if save_images else None
            for j in range(P):
                tp = self.PathGroup.TargetPaths[j]
                label = [[s.Poly, s.EndPoint] for s in tp.PathSeg]
                if save_images:
                    _save_image(tp.Space, r'{}/data/{}.jpg'.format(folder_path, j))
                for k in range(P):
                    m = j * P + k
                    index = i * P ** 2 + m
                    self.MapLabel.append([label, np.array([angle[m]]), [int(trans[m][0]), int(trans[m][1])],
                                          segpoint[m], pathpoint[m]])
                    if cnt < total_record:                            # MapGenerate.py:144-149
                        problem = {"Index": index + round_index * 100, "Init": list(segpoint[m][0]),
                                   "End": list(segpoint[m][10]), "Length": tp.Length,
                                   "Obstacles": [list(o) for o in obstacles[m][:n_obs[m][0]]]}
                        with open("./unsolved_problems.txt", "a") as f:
                            f.write(json.dumps(problem) + "\n")
                        cnt += 1
                    if save_images:
                        _save_image(rgb[m], r'{}/{}.jpg'.format(folder_path, index))

    def generate_map_randomly(self, path_point, init, end, length, path_obstacles, index):
        """MapGenerate.py:126-151: K random obstacles, clearance filter, obstacle raster -> tensor [3,R,R]."""
        global cnt
        R, K = self.Resolution, self.ObstaclesNum
        if rng.mode() == "philox":
            from ppnet_amd.philox import doubles_device
            d = doubles_device(rng.seed(), 4, rng.take_map_ids(1), 0, 3 * K, self.device)[None, :]
        else:
            d = torch.tensor(np.concatenate([np.random.random(K), np.random.random(K), np.random.random(K)])[None, :],
                             device=self.device)
        pp = torch.tensor(np.asarray(path_point, dtype=np.float64)[None], device=self.device)
        _, obs, counts = edage.obstacle_filter(pp, d, K, R, self.MapSize, self.ObstacleSize, self.Clearance)
        kept = obs[0, :int(counts[0])].cpu().numpy().tolist()
        allobs = kept + [[float(o[0]), float(o[1]), float(o[2])] for o in path_obstacles]
        if cnt < total_record:
            problem = {"Index": index, "Init": list(init), "End": list(end), "Length": length, "Obstacles": allobs}
            with open("./unsolved_problems.txt", "a") as f:
                f.write(json.dumps(problem) + "\n")
            cnt += 1
        from Path import plot_obstacles
        return plot_obstacles((R, R), allobs, resolution=(R, R)).to(self.device)


def _save_image(img, path):
    """torchvision.utils.save_image for a float [3,R,R] tensor in [0,1] (clamped), without torchvision."""
    from PIL import Image
    a = (img.detach().clamp(0, 1) * 255 + 0.5).to(torch.uint8).permute(1, 2, 0).cpu().numpy()
    Image.fromarray(a, mode="RGB").save(path)
