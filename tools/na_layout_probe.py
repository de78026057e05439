"""Does the attention kernels' time follow the qkv LAYOUT?  The same number of (image, head) problems and bytes, once as DiNAT-B stores
them (heads x 32 channels interleaved per token: a head's q / k / v are 64-byte pieces of a 768-byte token row) and once with ONE head
per token row (heads = 1, B x heads images: a token row is that head's q | k | v, 192 bytes, all of it used by the wave that touches it)
— the second is what a head-major qkv buffer would look like to the memory system.  Prints ms per launch for both (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppnet_amd.na import na2d_forward

dev = torch.device("cuda", 0)
def run(B, side, heads, d):
    C = heads * 32
    pad = max(side, 7 * d)
    qkv = torch.randn(B, side, side, 3 * C, device=dev, dtype=torch.bfloat16)
    rpb = torch.randn(heads, 13, 13, device=dev)
    kw = dict(pad_kv=torch.randn(3 * C, device=dev, dtype=torch.bfloat16), padded_hw=(pad, pad)) if pad > side else {}
    for _ in range(3):
        na2d_forward(qkv, rpb, heads, d, 32 ** -0.5, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        na2d_forward(qkv, rpb, heads, d, 32 ** -0.5, **kw)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10
for side, heads, dils in ((64, 4, (1, 16)), (32, 8, (1, 4, 8)), (16, 16, (1, 2, 3, 4))):
    for d in dils:
        a = run(256, side, heads, d)
        b = run(256 * heads, side, 1, d)
        print(f"side {side:3d} d {d:2d}: {heads:2d} heads interleaved {a:7.4f} ms   one head per token row {b:7.4f} ms   ratio {b / a:5.2f}", flush=True)
