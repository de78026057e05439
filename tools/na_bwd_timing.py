"""Times the NA backward (query pass + key pass + drpb sum) on the (level, dilation) shapes of DiNAT-B at 256x256, 8 images, float32
— the SegNet training step's shapes (diagnostic).  NA_BWD_B overrides the batch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppnet_amd.na import na2d_autograd

dev = torch.device("cuda", 0)
B = int(os.environ.get("NA_BWD_B", "8"))
tot = 0.0
for side, C, heads, dils, layers in ((64, 128, 4, (1, 8), (2, 1)), (32, 256, 8, (1, 4), (2, 2)), (16, 512, 16, (1, 2), (9, 9)), (8, 1024, 32, (1,), (5,))):
    for d, n in zip(dils, layers):
        qkv = torch.randn(B, side, side, 3 * C, device=dev, requires_grad=True)
        rpb = torch.randn(heads, 13, 13, device=dev, requires_grad=True)
        out = na2d_autograd(qkv, rpb, heads, d, 32 ** -0.5)
        g = torch.randn_like(out)
        for _ in range(3):
            out.backward(g, retain_graph=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            out.backward(g, retain_graph=True)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        tot += ms * n
        print(f"side {side:3d} C {C:4d} heads {heads:2d} d {d}: {ms * 1e3:7.1f} us per backward (incl. autograd glue)  x {n} layers")
print(f"DiNAT-B, {B} images: {tot:.2f} ms of NA backward per step")
