#!/usr/bin/env python3
"""NAT-Base + UPerHead batch time with the head's pooling / resize / concatenation steps on the build's kernels (ppn_adaptive_pools_nhwc,
ppn_upsample2x_add_nhwc, ppn_resize_concat_nhwc: the default) against PPNET_UPER_UNFUSED_RESIZE=1 (adaptive pools, interpolate + add,
interpolate + cat + channels_last copy on the framework), alternating in ONE process on one box; the labels of the two forms are
compared (the pooled 1x1 ConvModules move from F.linear to the GEMM kernel with them: bfloat16 roundings differ, labels at ties flip)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppnet_amd import edage
from ppnet_amd.segnet import NAT_BASE_UPER, SegNet
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pb = edage.generate_paths(4, 256, 50, 3, seed=0, device=dev)
g = edage.generate_maps(pb, B // 4, 5, 20, seed=0).grid[:B]
torch.manual_seed(0)
net = SegNet(**NAT_BASE_UPER).to(dev).eval().prepare_inference().to(torch.bfloat16)
def run(n):
    torch.cuda.synchronize(); t = time.perf_counter()
    with torch.no_grad():
        for _ in range(n):
            m = net.labels_u8(g)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3, m
lab = {}
for k in (None, "1"):
    os.environ.pop("PPNET_UPER_UNFUSED_RESIZE", None)
    if k: os.environ["PPNET_UPER_UNFUSED_RESIZE"] = k
    lab[k] = run(2)[1].clone()
print("labels equal between the two forms:", bool(torch.equal(lab[None], lab["1"])), " differing:", int((lab[None] != lab["1"]).sum()), "of", lab[None].numel(), flush=True)
for rnd in range(3):
    for k in (None, "1"):
        os.environ.pop("PPNET_UPER_UNFUSED_RESIZE", None)
        if k: os.environ["PPNET_UPER_UNFUSED_RESIZE"] = k
        print(f"round {rnd} {'framework resizes (PPNET_UPER_UNFUSED_RESIZE=1)' if k else 'own FPN resize kernels (default)':52s} NAT-Base + UPerHead {run(8)[0]:7.3f} ms per batch of {B}", flush=True)
