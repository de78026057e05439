#!/usr/bin/env python3
"""PPNet batch time under the GEMM knobs, alternating in ONE process on one box (the knobs are read at call time):
default (round 5) = the build's LN-folded / accumulating GEMMs (ppn_nat_gemm_bf16) + the fused MLP kernel (ppn_nat_mlp_bf16) on every
level; PPNET_LIBRARY_GEMM_FROM_C=512 = round 4's gate (LayerNorm kernels + the vendor GEMM from C = 512 on); =1024 = the vendor path on level 3 only; PPNET_LIBRARY_GEMM = the vendor library on every level (round 2's
default); PPNET_NO_FUSED_MLP = level 1's MLP as two GEMM launches again; PPNET_NO_LN_FOLD = LayerNorm kernels + the build's
one-tile-per-workgroup GEMM (ppn_gemm_bf16) where the vendor's is not selected."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppnet_amd import edage
from ppnet_amd.ppnet import PPNet
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pb = edage.generate_paths(4, 256, 50, 3, seed=0, device=dev)
mb = edage.generate_maps(pb, B // 4, 5, 20, seed=0)
g = mb.grid[:B]
torch.manual_seed(0)
model = PPNet(resolution=256).to(dev).eval()
def run(n):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        m = model.segment_u8(g)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
KN = ("PPNET_LIBRARY_GEMM_FROM_C=512", "PPNET_LIBRARY_GEMM_FROM_C=1024", "PPNET_LIBRARY_GEMM=1", "PPNET_NO_FUSED_MLP=1", "PPNET_NO_LN_FOLD=1")
def setk(k):
    for x in KN: os.environ.pop(x.split("=")[0], None)
    if k: os.environ[k.split("=")[0]] = k.split("=")[1]
for k in (None,) + KN:
    setk(k); run(2)
for rnd in range(3):
    for k in (None,) + KN:
        setk(k)
        print(f"round {rnd} {k or 'default (own kernels on every level)':58s} SegNet {run(8):7.3f} ms per batch of {B}", flush=True)
