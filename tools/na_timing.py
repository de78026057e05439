"""Times the NA kernel alone on every (level, dilation) shape of DiNAT-B at 256x256, batch 256, bf16 (diagnostic).
Prints ms per launch, queries/ns and the fraction of the HBM time its q/k/v/out bytes would take at 8 TB/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppnet_amd.na import na2d_forward

dev = torch.device("cuda", 0)
B = 256
shapes = []   # (side, C, heads, dilation)
for side, C, heads, dils in ((64, 128, 4, (1, 16)), (32, 256, 8, (1, 4, 8)), (16, 512, 16, (1, 2, 3, 4)), (8, 1024, 32, (1, 2))):
    for d in dils:
        shapes.append((side, C, heads, d))
if os.environ.get("NA_SHAPE"):                               # "side,dilation": one shape only (counter passes)
    want = tuple(int(v) for v in os.environ["NA_SHAPE"].split(","))
    shapes = [sh for sh in shapes if (sh[0], sh[3]) == want]
for side, C, heads, d in shapes:
    pad = max(side, 7 * d)
    qkv = torch.randn(B, side, side, 3 * C, device=dev, dtype=torch.bfloat16)     # real tokens; the pad is virtual (as the module runs it)
    rpb = torch.randn(heads, 13, 13, device=dev)
    kw = dict(pad_kv=torch.randn(3 * C, device=dev, dtype=torch.bfloat16), padded_hw=(pad, pad)) if pad > side else {}
    for _ in range(3):
        na2d_forward(qkv, rpb, heads, d, 32 ** -0.5, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        na2d_forward(qkv, rpb, heads, d, 32 ** -0.5, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    nq = B * side * side * heads
    byts = (B * side * side * 3 * C + B * side * side * C) * 2
    print(f"side {side:3d} C {C:4d} d {d:2d} pad {pad:3d}: {ms:7.4f} ms  {nq / ms / 1e6:6.2f} q/ns  bytes {byts / 1e6:7.1f} MB  hbm-time frac {byts / 8e12 * 1e3 / ms:5.2f}")
    del qkv
