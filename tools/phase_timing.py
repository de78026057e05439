"""Diagnostic: per-phase cycle shares of edage_maps_kernel (build: make -C ppnet_amd/csrc timing)."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import ppnet_amd._lib as L
tl = C.CDLL(os.path.join(ROOT, "ppnet_amd", "libppnet_hip_timing.so"))
for n in ("ppn_edage_paths", "ppn_edage_maps"):
    getattr(tl, n).argtypes = getattr(L.lib, n).argtypes
    getattr(tl, n).restype = C.c_int
L.lib = tl
from ppnet_amd import edage
dev = torch.device("cuda:0")
pb = edage.generate_paths(100, 256, 50, 3, seed=0, device=dev)
mb = edage.MapsBatch(10000, 256, 20, dev)
buf = (C.c_ulonglong * 16)()
for it in range(3):
    edage.generate_maps(pb, 100, 5, 20, seed=0, out=mb)
torch.cuda.synchronize()
tl.ppn_debug_phase_cycles(buf, 1)
edage.generate_maps(pb, 100, 5, 20, seed=0, out=mb)
torch.cuda.synchronize()
tl.ppn_debug_phase_cycles(buf, 1)
names = ["load space/hull/cand", "placement", "labels", "filter+compact", "pocket+zero+obs out", "raster1 spans", "raster2 corridor", "raster3 store"]
tot = sum(buf[:8])
for i, n in enumerate(names):
    print(f"{n:24s} {buf[i]/10000:10.0f} cycles/WG  {100*buf[i]/tot:5.1f}%")
print("total cycles/WG", tot / 10000, "(s_memtime = shader cycles)")
