"""Diagnostic: how edage_maps_kernel's time moves when outputs are dropped (which resource bounds it?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from ppnet_amd import edage
dev = torch.device("cuda:0")
pb = edage.generate_paths(100, 256, 50, 3, seed=0, device=dev)
def run(want_pp, want_acc, n=30):
    mb = edage.MapsBatch(10000, 256, 20, dev, want_pathpoint=want_pp, want_accept=want_acc)
    for _ in range(5): edage.generate_maps(pb, 100, 5, 20, seed=0, out=mb)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): edage.generate_maps(pb, 100, 5, 20, seed=0, out=mb)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("full outputs        ms", run(True, True))
print("no pathpoint labels ms", run(False, True))
print("no labels, no accept ms", run(False, False))
