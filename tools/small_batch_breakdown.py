"""Per-kernel GPU time of one PPNet segment+heatmap pass at a small batch, default against PPNET_LIBRARY_GEMM (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from ppnet_amd import edage
from ppnet_amd.ppnet import PPNet
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
pb = edage.generate_paths(1, 256, 50, 3, seed=0, device=dev)
mb = edage.generate_maps(pb, max(B, 1), 5, 20, seed=0)
g = mb.grid[:B].contiguous()
torch.manual_seed(0)
model = PPNet(resolution=256).to(dev).eval()
for knob in (None, "PPNET_LIBRARY_GEMM"):
    os.environ.pop("PPNET_LIBRARY_GEMM", None)
    if knob: os.environ[knob] = "1"
    for _ in range(3):
        model.heatmap(model.segment_u8(g))
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(4):
            model.heatmap(model.segment_u8(g))
        torch.cuda.synchronize()
    rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
    tot = sum(e.device_time_total for e in rows)
    print(f"== {knob or 'default'}: kernel time per pass {tot / 4e3:.3f} ms, {sum(e.count for e in rows) // 4} launches")
    for e in rows[:int(os.environ.get('TOP', '14'))]:
        print("%5d  %8.3f ms  %s" % (e.count // 4, e.device_time_total / 4e3, e.key[:110]))
