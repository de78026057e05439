#!/usr/bin/env python3
"""Distribution of the bf16-vs-float32 logit / margin error of bench.py's PPNet objects (what `ppnet.parity` summarises)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from ppnet_amd import edage, fused
from ppnet_amd.segnet import IMG_MEAN, IMG_STD
dev = torch.device("cuda:0")
pb = edage.generate_paths(4, 256, 50, 3, seed=0, device=dev)
mb = edage.generate_maps(pb, 4, 5, 20, seed=0)
g = mb.grid
p16 = bench.bench_ppnet(torch, dev, 256, calibrate=g[:4])
p32 = bench.bench_ppnet(torch, dev, 256, weights_dtype="f32", calibrate=g[:4])
with torch.no_grad():
    l32 = p32.segnet.encode_decode(fused.grid_to_image(g, IMG_MEAN, IMG_STD, torch.float32)).float()
    l16 = p16.segnet.encode_decode(fused.grid_to_image(g, IMG_MEAN, IMG_STD, torch.bfloat16)).float()
e = (l16 - l32).abs().flatten()
rms32 = float(l32.pow(2).mean().sqrt())
m32, m16 = l32[:, 1] - l32[:, 0], l16[:, 1] - l16[:, 0]
me = (m16 - m32).abs().flatten()
print("logit rms", rms32, "margin std", float(m32.std()))
for q in (0.5, 0.9, 0.99, 0.999, 0.9999, 1.0):
    print(f"q{q}: logit err {float(torch.quantile(e[:4000000], q)) / rms32:.5f} of logit rms   margin err {float(torch.quantile(me[:4000000], q)) / float(m32.std()):.4f} of margin std")
rms = float((l16 - l32).pow(2).mean().sqrt())
agree = (m16 > 0) == (m32 > 0)
for k in (3, 6, 12, 24, 48):
    sure = m32.abs() > k * rms
    print(f"band {k} x rms: pixels outside {float(sure.float().mean()):.4f}, flips outside {float((~agree & sure).float().mean()):.2e}")
print(bench.ppnet_parity(torch, p16, p32, g))
