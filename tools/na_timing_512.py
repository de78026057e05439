"""The NA kernel on every (level, dilation) shape of DiNAT-B at 512x512 (config 5), batch 256, bf16: ms per launch.
Knobs (read by the library): PPNET_NA_MFMA=1 sends every launch to the matrix-core kernels, PPNET_NA_RT=4|8|16 forces their region size."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppnet_amd.na import na2d_forward
dev = torch.device("cuda", 0)
B = int(os.environ.get("NA_B", "256"))
tot = 0.0
for side, C, heads, dils, counts in ((128, 128, 4, (1, 16), (2, 1)), (64, 256, 8, (1, 4, 8), (2, 1, 1)), (32, 512, 16, (1, 2, 3, 4), (9, 3, 3, 3)), (16, 1024, 32, (1, 2), (3, 2))):
    for d, cnt in zip(dils, counts):
        pad = max(side, 7 * d)
        qkv = torch.randn(B, side, side, 3 * C, device=dev, dtype=torch.bfloat16)
        rpb = torch.randn(heads, 13, 13, device=dev)
        kw = dict(pad_kv=torch.randn(3 * C, device=dev, dtype=torch.bfloat16), padded_hw=(pad, pad)) if pad > side else {}
        for _ in range(2):
            na2d_forward(qkv, rpb, heads, d, 32 ** -0.5, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            na2d_forward(qkv, rpb, heads, d, 32 ** -0.5, **kw)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        byts = (B * side * side * 3 * C + B * side * side * C) * 2
        tot += ms * cnt
        print(f"side {side:3d} C {C:4d} d {d:2d} pad {pad:3d} x{cnt}: {ms:7.4f} ms  hbm-time frac {byts / 8e12 * 1e3 / ms:5.2f}", flush=True)
        del qkv
print(f"all 30 layers: {tot:.2f} ms   knobs: MFMA={os.environ.get('PPNET_NA_MFMA')} RT={os.environ.get('PPNET_NA_RT')}")
