#!/usr/bin/env python3
"""s_memtime trace of the fused MLP kernel's workgroup 0 (build: make -C ppnet_amd/csrc trace; PPNET_HIP_LIB=.../libppnet_hip_trace.so).
Slots: 1 block top, 2 rows landed, 3 statistics + O init done, 10 chunk top, 11 after the chunk barrier, 4 chunk loop done,
5 last stage 2 done, 6 epilogue issued."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppnet_amd import fused
dev = torch.device("cuda:0")
M, C, hid = 262144, 256, 512
s_ = torch.randn(M, C, device=dev, dtype=torch.bfloat16)
w1 = (torch.randn(hid, C, device=dev) * 0.05).to(torch.bfloat16)
w2 = (torch.randn(C, hid, device=dev) * 0.05).to(torch.bfloat16)
hb = torch.stack([w1.float().sum(1), torch.randn(hid, device=dev)], dim=1).contiguous()
b2 = torch.randn(C, device=dev)
wpk = fused.nat_mlp_pack(w1, w2)
st = torch.zeros(C // 128, M, 2, dtype=torch.float32, device=dev)
for _ in range(10):
    fused.nat_mlp_(s_, wpk, hb, b2, hid, stats_out=st)
torch.cuda.synchronize()
raw = st.view(torch.int64).cpu().numpy().reshape(-1)
for wave in (0, 1, 4, 5):
    r = raw[wave * 8192: wave * 8192 + 8192].reshape(-1, 2)
    n = 0
    while n < len(r) and r[n, 1] != 0:
        n += 1
    r = r[:n]
    t0 = r[0, 0]
    ev = [(int(t - t0), int(s)) for t, s in r]
    print(f"wave {wave}: {n} stamps, total {ev[-1][0]} ticks")
    # per block summary
    blocks, cur = [], None
    for t, s in ev:
        if s == 1:
            cur = {"top": t, "iters": []}
            blocks.append(cur)
        elif s == 2: cur["rows"] = t
        elif s == 3: cur["init"] = t
        elif s == 10: cur["iters"].append([t, None])
        elif s == 11: cur["iters"][-1][1] = t
        elif s == 4: cur["loop_end"] = t
        elif s == 5: cur["s2"] = t
        elif s == 6: cur["epi"] = t
    for i, b in enumerate(blocks[:4]):
        its = b["iters"]
        dur = [its[k + 1][0] - its[k][0] for k in range(len(its) - 1)]
        bar = [x[1] - x[0] for x in its]
        print(f"  block {i}: wait rows {b['rows'] - b['top']}, stats+init {b['init'] - b['rows']}, loop {b['loop_end'] - b['init']} (iter mean {sum(dur) / max(len(dur), 1):.0f}, "
              f"barrier wait mean {sum(bar) / len(bar):.0f}, first iters {dur[:4]}), last stage 2 {b['s2'] - b['loop_end']}, epilogue {b['epi'] - b['s2']}")
