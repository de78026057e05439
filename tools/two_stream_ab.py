#!/usr/bin/env python3
"""PPNet batches issued on ONE stream against consecutive batches alternating over TWO (or more) streams: the HBM-bound kernels of one batch
(attention, LayerNorm, GenNet, tail) beside the matrix-core kernels of the other.  ms per batch of 256, same box, alternating runs.
    python tools/two_stream_ab.py [batch] [streams]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ppnet_amd import edage, evaluate
from ppnet_amd.ppnet import PPNet

R = 256
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = PPNet(resolution=R).to(dev).eval()
pb = edage.generate_paths(3, R, 50, 3, seed=0, device=dev)
mb = edage.generate_maps(pb, 100, 5, 20, seed=0)
g = mb.grid[:B].contiguous()
init, end = mb.segpoint[:B, 0].contiguous(), mb.segpoint[:B, 10].contiguous()
obs, n_obs = mb.obstacles[:B].contiguous(), mb.n_obstacles[:B, 0].contiguous()
ridge = evaluate.label_heatmaps(pb, mb, 100)[:B].contiguous()
streams = [torch.cuda.Stream(dev) for _ in range(NS)]


def one():
    heat = model.heatmap(model.segment_u8(g))
    return model.plan_tail(ridge, init, end, obs, n_obs), heat


def run(n, ns):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    keep = []
    for i in range(n):
        with torch.cuda.stream(streams[i % ns]):
            keep.append(one())
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for ns in range(1, NS + 1):
    run(2 * ns, ns)
for rep in range(3):
    print("  ".join(f"{ns} stream(s): {run(12, ns):7.3f} ms/batch" for ns in range(1, NS + 1)), flush=True)
