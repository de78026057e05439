"""Profile target: a few PPNet plan() calls at batch 256 (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from ppnet_amd import edage
from ppnet_amd.ppnet import PPNet
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pb = edage.generate_paths(16, 256, 50, 3, seed=0, device=dev)
mb = edage.generate_maps(pb, B // 16, 5, 20, seed=0)
torch.manual_seed(0)
import os
if os.environ.get("PPN_CUDNN_BENCHMARK") == "1":
    torch.backends.cudnn.benchmark = True
m = PPNet(256).to(dev).eval()
args = (mb.grid, mb.segpoint[:, 0].contiguous(), mb.segpoint[:, 10].contiguous(), mb.obstacles, mb.n_obstacles[:, 0].contiguous(), 1 / 50 * 224)
for _ in range(2):
    m.plan(*args)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(3):
    m.plan(*args)
torch.cuda.synchronize()
print("ms per batch", (time.perf_counter() - t) / 3 * 1e3)
if os.environ.get("PPNET_TUNE_GEMMS"):
    import torch.cuda.tunable as tn
    res = tn.get_results()
    print("tunable results:", len(res))
    with open(os.environ["PPNET_TUNE_GEMMS"], "w") as f:
        for k, v in tn.get_validators():
            f.write(f"Validator,{k},{v}\n")
        for r in res:
            f.write(",".join(str(t) for t in r) + "\n")
