"""Per-kernel GPU time of one steady-state SegNet (DiNAT-B + SETR-UP) training step at 8 images (diagnostic).
TRAIN_MIOPEN_FIND=1 lets MIOpen search its convolution algorithms (torch.backends.cudnn.benchmark)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from ppnet_amd import edage, train
from ppnet_amd.segnet import SegNet

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = bool(os.environ.get("TRAIN_MIOPEN_FIND"))
pb = edage.generate_paths(1, 256, 50, 3, seed=2, device=dev)
mb = edage.generate_maps(pb, 8, 5, 20, seed=2)
grid, space, path = train.generator_pairs(pb, mb, 8)
seg = SegNet().to(dev)
trainer = train.segnet_trainer(seg)
opt = train.segnet_optimizer(trainer)
it = [0]
def step():
    it[0] += 1
    return train.segnet_train_step(trainer, opt, it[0], 160000, grid, space)
for _ in range(4):
    step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5):
    step()
torch.cuda.synchronize()
print("step %.1f ms" % ((time.perf_counter() - t) / 5 * 1e3))
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(2):
        step()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
tot = sum(e.device_time_total for e in rows)
print("kernel time per step %.2f ms" % (tot / 2e3))
for e in rows[:int(os.environ.get("TOP", "28"))]:
    print("%5d  %8.3f ms/step  %s" % (e.count // 2, e.device_time_total / 2e3, e.key[:120]))
