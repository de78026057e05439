#!/usr/bin/env python3
"""The fused MLP kernel alone (ppn_nat_mlp_bf16, level-1 shape of DiNAT-B at batch 256): ms per launch, interleaved rounds.
PPNET_HIP_LIB selects a diagnostic build (make abl: one part of the kernel removed each)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppnet_amd import fused
dev = torch.device("cuda:0")
M, C = 262144, 256
hid = int(sys.argv[1]) if len(sys.argv) > 1 else 512
s_ = torch.randn(M, C, device=dev, dtype=torch.bfloat16)
w1 = (torch.randn(hid, C, device=dev) * 0.05).to(torch.bfloat16)
w2 = (torch.randn(C, hid, device=dev) * 0.05).to(torch.bfloat16)
hb = torch.stack([w1.float().sum(1), torch.randn(hid, device=dev)], dim=1).contiguous()
b2 = torch.randn(C, device=dev)
wpk = fused.nat_mlp_pack(w1, w2)
st = torch.empty(C // 128, M, 2, dtype=torch.float32, device=dev)
for _ in range(20):
    fused.nat_mlp_(s_, wpk, hb, b2, hid, stats_out=st)
torch.cuda.synchronize()
ts = []
for rnd in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(40):
        fused.nat_mlp_(s_, wpk, hb, b2, hid, stats_out=st)
    b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) / 40)
print(os.path.basename(os.environ.get("PPNET_HIP_LIB", "libppnet_hip.so")), "hidden", hid, "ms per launch: min %.4f median %.4f" % (min(ts), sorted(ts)[2]))
