"""Diagnostic: which aten ops of a PPNet batch launch copy / elementwise kernels, with shapes (torch.profiler)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from ppnet_amd import edage
from ppnet_amd.ppnet import PPNet
dev = torch.device("cuda:0")
pb = edage.generate_paths(16, 256, 50, 3, seed=0, device=dev)
mb = edage.generate_maps(pb, 16, 5, 20, seed=0)
torch.manual_seed(0)
m = PPNet(256).to(dev).eval()
args = (mb.grid, mb.segpoint[:, 0].contiguous(), mb.segpoint[:, 10].contiguous(), mb.obstacles, mb.n_obstacles[:, 0].contiguous(), 1 / 50 * 224)
for _ in range(2):
    m.plan(*args)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    m.plan(*args)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=150, max_name_column_width=40, max_shapes_column_width=60))
