"""Diagnostic: per-phase cycle shares of edage_maps_kernel (build: make -C ppnet_amd/csrc timing)."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import ppnet_amd._lib as L
tl = C.CDLL(os.environ.get("PPNET_TIMING_LIB") or os.path.join(ROOT, "ppnet_amd", "libppnet_hip_timing.so"))
for n in ("ppn_edage_paths", "ppn_edage_paths_ex", "ppn_edage_paths_ex2", "ppn_edage_maps"):
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

# ---- stage A
pbuf = (C.c_ulonglong * 16)()
tl.ppn_debug_paths_phase_cycles(pbuf, 1)
import time
for it in range(3):
    edage.generate_paths(100, 256, 50, 3, seed=0, device=dev, out=pb)
torch.cuda.synchronize()
tl.ppn_debug_paths_phase_cycles(pbuf, 1)
t0 = time.perf_counter()
edage.generate_paths(100, 256, 50, 3, seed=0, device=dev, out=pb)
torch.cuda.synchronize()
print("paths kernel wall ms (incl. launch)", (time.perf_counter() - t0) * 1e3)
tl.ppn_debug_paths_phase_cycles(pbuf, 1)
pn = ["A1 fits", "A2 chaining", "path points + length", "A3/A4 rays", "A5 hull", "A6 normalise", "space bits", "A7 isles", "A8 pockets"]
ptot = sum(pbuf[:9])
for i, n in enumerate(pn):
    print(f"{n:24s} {pbuf[i]/100:10.0f} cycles/WG  {100*pbuf[i]/ptot:5.1f}%")
print("total cycles/WG", ptot / 100)
