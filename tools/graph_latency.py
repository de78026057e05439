#!/usr/bin/env python3
"""PPNet batch latency, eager launches against one HIP graph (PPNet.capture), at small and large batches.
    python tools/graph_latency.py [batches...]      (default 1 4 16 64 256)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ppnet_amd import edage, evaluate
from ppnet_amd.ppnet import PPNet

R = 256
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = PPNet(resolution=R).to(dev).eval()
pb = edage.generate_paths(3, R, 50, 3, seed=0, device=dev)
mb = edage.generate_maps(pb, 100, 5, 20, seed=0)
ridge_all = evaluate.label_heatmaps(pb, mb, 100)
for B in [int(a) for a in sys.argv[1:]] or [1, 4, 16, 64, 256]:
    g = mb.grid[:B].contiguous()
    init, end = mb.segpoint[:B, 0].contiguous(), mb.segpoint[:B, 10].contiguous()
    obs, n_obs = mb.obstacles[:B].contiguous(), mb.n_obstacles[:B, 0].contiguous()
    ridge = ridge_all[:B].contiguous()

    def eager():                                        # the captured body: both networks on the grids, the tail on the ridge maps
        heat = model.heatmap(model.segment_u8(g))
        return model.plan_tail(ridge, init, end, obs, n_obs), heat
    cp = model.capture(g, init, end, obs, n_obs, tail_heat=ridge)
    n = 20 if B <= 64 else 5
    out = []
    for fn in (eager, cp.replay):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / n * 1e3)
    print(f"batch {B:4d}: eager {out[0]:7.3f} ms   HIP graph {out[1]:7.3f} ms   ({out[0] / out[1]:.2f}x)", flush=True)
    del cp
