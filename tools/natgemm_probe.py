#!/usr/bin/env python3
"""Per-k-tile and per-tile cost of ppn_nat_gemm_bf16 from two shapes that differ only in K (diagnostic; PPNET_HIP_LIB selects the build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppnet_amd import fused
dev = torch.device("cuda:0")
def timeit(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
name = os.environ.get("PPNET_HIP_LIB", "default").split("/")[-1]
for mode in ("ln", "ln_gelu", "acc"):
    res = []
    for K in (512, 1024):
        M, N = 65536, (1536 if mode != "acc" else 512)
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
        b32 = torch.randn(N, device=dev)
        out = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
        if mode == "acc":
            st = torch.empty(fused.nat_partials(N), M, 2, dtype=torch.float32, device=dev)
            t = timeit(lambda: fused.nat_gemm(a, w, b32, "acc", out, stats_out=st))
        else:
            st = fused.row_stats(a); cs = w.float().sum(1).contiguous()
            t = timeit(lambda: fused.nat_gemm(a, w, b32, mode, out, colsum=cs, stats_in=st))
        res.append(t)
    tiles = (65536 // 256) * (N // 256) / 256.0
    per_k = (res[1] - res[0]) / tiles / 8 * 1e3
    fixed = res[0] / tiles * 1e3 - (8 + (4 if mode == "acc" else 0)) * per_k
    print(f"{name:28s} {mode:8s} K=512 {res[0]:.4f} ms  K=1024 {res[1]:.4f} ms  -> {per_k:.3f} us per k-tile, {fixed:.2f} us fixed per tile ({tiles:.0f} tiles per CU)")
