import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppnet_amd import edage
from ppnet_amd.ppnet import PPNet
dev = torch.device("cuda:0")
pb = edage.generate_paths(3, 256, 50, 3, seed=0, device=dev)
mb = edage.generate_maps(pb, 100, 5, 20, seed=0)
torch.manual_seed(0)
model = PPNet(resolution=256).to(dev).eval()
for B in [int(a) for a in sys.argv[1:]]:
    g = mb.grid[:B].contiguous()
    for it in range(3):
        print("batch", B, "pass", it, flush=True)
        m = model.segment_u8(g); torch.cuda.synchronize()
        h = model.heatmap(m); torch.cuda.synchronize()
    print("ok", B, float(h.float().mean()), flush=True)
