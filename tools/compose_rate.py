"""Fraction of stage-B maps whose corridor-compose pass runs (PPN_FLAG_CORRIDOR_PASS) at the bench configuration (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppnet_amd import edage, _lib
dev = torch.device("cuda:0")
for R in (256, 224, 512):
    pb = edage.generate_paths(100, R, 50, 3, seed=0, device=dev)
    mb = edage.generate_maps(pb, 100, obstacles_size=5, obstacles_num=20, seed=0)
    torch.cuda.synchronize()
    f = ((mb.flags & _lib.FLAG_CORRIDOR_PASS) != 0).float().mean().item()
    print(f"R={R}: compose pass on {100 * f:.2f} % of maps; max_step_px mean {pb.max_step_px.mean().item():.3f} max {pb.max_step_px.max().item():.3f}")
