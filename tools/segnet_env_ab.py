#!/usr/bin/env python3
"""SegNet batch time with and without environment knobs that the library reads per launch, alternating in ONE process on one box:
    python tools/segnet_env_ab.py PPNET_NA_HALO_BLOCK=4x4 [OTHER=VALUE ...]      (each knob is one arm; the first arm is "no knob")"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppnet_amd import edage
from ppnet_amd.ppnet import PPNet
dev = torch.device("cuda:0")
B = 256
pb = edage.generate_paths(4, 256, 50, 3, seed=0, device=dev)
mb = edage.generate_maps(pb, B // 4, 5, 20, seed=0)
g = mb.grid[:B]
torch.manual_seed(0)
model = PPNet(resolution=256).to(dev).eval()
def run(n):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        model.segment_u8(g)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
KN = tuple(a for a in sys.argv[1:] if "=" in a)
def setk(k):
    for x in KN: os.environ.pop(x.split("=")[0], None)
    if k: os.environ[k.split("=")[0]] = k.split("=", 1)[1]
for k in (None,) + KN:
    setk(k); run(2)
for rnd in range(4):
    for k in (None,) + KN:
        setk(k)
        print(f"round {rnd} {k or 'default':40s} SegNet {run(8):7.3f} ms per batch of {B}", flush=True)
