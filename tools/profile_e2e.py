#!/usr/bin/env python3
"""The config-5 step (generate -> SegNet -> GenNet -> planner tail at 512 x 512, 256 problems) a few times on one stream, for the kernel
tracer: rocprofv3 --kernel-trace -- python3 tools/profile_e2e.py, then tools/kernel_breakdown.py <dir> extract_paths_kernel 0 40."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ppnet_amd import edage, shard
dev = torch.device("cuda:0")
model = bench.bench_ppnet(torch, dev, bench.R5)
pb = edage.PathsBatch(bench.PATHS5, bench.R5, bench.MAP_SIZE, bench.CLEARANCE, dev)
mb = edage.MapsBatch(bench.PATHS5 * bench.PLACEMENTS5, bench.R5, bench.K, dev)
def one(it):
    fp, _, fm = shard.local_ids(bench.PATHS5, bench.PLACEMENTS5, 0, 1, batch_index=it)
    return model.generate_and_plan(pb, mb, bench.PLACEMENTS5, fp, fm, seed=bench.SEED + 5, obstacles_size=bench.OBST_SIZE, obstacles_num=bench.K)
for i in range(3):
    one(i)
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(3):
    one(10 + i)
torch.cuda.synchronize()
print("ms per step (one stream)", (time.perf_counter() - t) / 3 * 1e3)
