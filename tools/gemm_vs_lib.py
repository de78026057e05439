#!/usr/bin/env python3
"""The NAT projection shapes of DiNAT-B at batch 256 (levels 1-3): own MFMA GEMM (ppn_gemm_bf16) against the vendor library call
the default path makes (torch.addmm / F.linear -> hipBLASLt with the shipped TunableOp table).  ms per call, TF/s, GB/s."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from ppnet_amd import fused
from ppnet_amd.ppnet import _use_tuned_gemms

_use_tuned_gemms()
dev = torch.device("cuda:0")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for lvl, (M, C) in enumerate([(262144, 256), (65536, 512), (16384, 1024)], start=1):
    for name, N, K, epi in (("qkv ", 3 * C, C, "bias"), ("proj", C, C, "accum"), ("fc1 ", 2 * C, C, "bias_gelu"), ("fc2 ", C, 2 * C, "accum")):
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
        b32 = torch.randn(N, device=dev)
        b16 = b32.to(torch.bfloat16)
        out = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
        if epi == "accum":
            lib = lambda: out.addmm_(a, w.t())
        elif epi == "bias":
            lib = lambda: F.linear(a, w, b16)
        else:                                              # what Mlp.hidden calls: bias + erf GELU in the library GEMM's epilogue
            lib = lambda: torch._addmm_activation(b16, a, w.t(), use_gelu=True)
        own = lambda: fused.gemm_bf16(a, w, b32, epi, out=out)
        own_p = lambda: fused.gemm_bf16(a, w, b32, epi, out=out, persistent_blocks=256)
        t_lib, t_own, t_p = timeit(lib), timeit(own), timeit(own_p)
        fl = 2.0 * M * N * K
        by = 2.0 * (M * K + M * N * (2 if epi == "accum" else 1) + N * K)
        print(f"level {lvl} {name} M={M:7d} N={N:5d} K={K:5d}: library {t_lib:.4f} ms ({fl / t_lib / 1e9:6.0f} TF/s, {by / t_lib / 1e6:5.0f} GB/s)   "
              f"own {t_own:.4f} ms ({fl / t_own / 1e9:6.0f} TF/s)   own persistent {t_p:.4f} ms   HBM floor at 5 TB/s {by / 5e9:.4f} ms", flush=True)
