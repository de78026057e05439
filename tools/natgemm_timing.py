#!/usr/bin/env python3
"""The NAT projection shapes of DiNAT-B at batch 256 (levels 1-3): the build's kernels (ppn_nat_gemm_bf16: LayerNorm folded into
qkv / fc1, residual + row statistics in proj / fc2) against the vendor GEMM they replace (torch.addmm_ / F.linear / _addmm_activation ->
hipBLASLt with the shipped TunableOp table).  ms per call; TF/s and GB/s of the build's kernel.  The vendor call is the GEMM ALONE:
in the network it runs behind a LayerNorm kernel (qkv, fc1) or in front of a residual + LayerNorm kernel (proj, fc2) that the build's
kernels fold in — proj / fc2 here read and rewrite the residual stream and emit its row statistics, the vendor's addmm_ does not —
so the like-for-like comparison is the whole network under the knobs: tools/ppnet_ab.py."""
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


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


tot_lib = tot_own = tot_fused = 0.0
per = {}
for lvl, (M, C, depth) in enumerate([(262144, 256, 4), (65536, 512, 18), (16384, 1024, 5)], start=1):
    for name, N, K, mode in (("qkv ", 3 * C, C, "ln"), ("proj", C, C, "acc"), ("fc1 ", 2 * C, C, "ln_gelu"), ("fc2 ", C, 2 * C, "acc")):
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
        b32 = torch.randn(N, device=dev)
        b16 = b32.to(torch.bfloat16)
        out = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
        if mode == "acc":
            lib = lambda: out.addmm_(a, w.t())
            st = torch.empty(fused.nat_partials(N), M, 2, dtype=torch.float32, device=dev)
            own = lambda: fused.nat_gemm(a, w, b32, "acc", out, stats_out=st)
        else:
            lib = (lambda: F.linear(a, w, b16)) if mode == "ln" else (lambda: torch._addmm_activation(b16, a, w.t(), use_gelu=True))
            st = fused.row_stats(a)
            cs = w.float().sum(1).contiguous()
            own = lambda: fused.nat_gemm(a, w, b32, mode, out, colsum=cs, stats_in=st)
        t_lib, t_own = timeit(lib), timeit(own)
        fl = 2.0 * M * N * K
        by = 2.0 * (M * K + M * N * (2 if mode == "acc" else 1) + N * K)
        tot_lib += depth * t_lib; tot_own += depth * t_own
        per[(lvl, name.strip())] = (t_lib, t_own)
        print(f"level {lvl} {name} M={M:7d} N={N:5d} K={K:5d}: library GEMM alone {t_lib:.4f} ms"
              + f"   own {t_own:.4f} ms ({fl / t_own / 1e9:6.0f} TF/s, {by / t_own / 1e6:5.0f} GB/s)   own / library {t_own / t_lib:.2f}"
              + f"   HBM floor at 6 TB/s {by / 6e9:.4f} ms", flush=True)
    # the MLP as ONE kernel (ppn_nat_mlp_bf16): LN -> fc1 -> GELU -> fc2 -> residual, the hidden activation never in HBM
    hid = 2 * C
    if fused.nat_mlp_ok(M, C, hid):
        s_ = torch.randn(M, C, device=dev, dtype=torch.bfloat16)
        w1 = (torch.randn(hid, C, device=dev) * 0.05).to(torch.bfloat16)
        w2 = (torch.randn(C, hid, device=dev) * 0.05).to(torch.bfloat16)
        hb = torch.stack([w1.float().sum(1), torch.randn(hid, device=dev)], dim=1).contiguous()
        b2 = torch.randn(C, device=dev)
        wpk = fused.nat_mlp_pack(w1, w2)
        stf = torch.empty(C // 128, M, 2, dtype=torch.float32, device=dev)
        t_f = timeit(lambda: fused.nat_mlp_(s_, wpk, hb, b2, hid, stats_out=stf))
        lib2 = per[(lvl, "fc1")][0] + per[(lvl, "fc2")][0]
        own2 = per[(lvl, "fc1")][1] + per[(lvl, "fc2")][1]
        fl = 2.0 * M * C * hid * 2
        print(f"level {lvl} MLP fused M={M:7d} C={C:5d} hidden={hid:5d}: {t_f:.4f} ms ({fl / t_f / 1e9:6.0f} TF/s, {4.0 * M * C / t_f / 1e6:5.0f} GB/s of s)"
              + f"   library fc1 + fc2 alone {lib2:.4f} ms   own two-kernel form {own2:.4f} ms   fused / library {t_f / lib2:.2f}", flush=True)
        tot_fused += depth * (t_f - own2)
print(f"DiNAT-B dense half of levels 1-3 per batch of 256 with the fused MLP where it applies: {tot_own + tot_fused:.2f} ms")
print(f"DiNAT-B dense half of levels 1-3 per batch of 256: library GEMMs alone {tot_lib:.2f} ms, own kernels (LayerNorm / residual / statistics inside) {tot_own:.2f} ms")
