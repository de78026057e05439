"""Training-step timings on one GPU (config sizes of the reference: GenNet batch 8 at 224 -> here 256, SegNet 8 images per GPU).
Data: generator kernels (stage A/B + label masks), built once outside the timed loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppnet_amd import edage, train
from ppnet_amd.gennet import AEViT
from ppnet_amd.segnet import SegNet

dev = torch.device("cuda:0")
R = 256


def pairs(n_paths, placements, seed):
    pb = edage.generate_paths(n_paths, R, 50, 3, seed=seed, device=dev)
    mb = edage.generate_maps(pb, placements, 5, 20, seed=seed)
    return train.generator_pairs(pb, mb, placements)


def timeit(step, n):
    for _ in range(3):
        step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


ONLY = os.environ.get("TRAIN_TIMING_ONLY", "")                      # "segnet" / "gennet": one model only (profiler passes)
for batch, amp in (() if ONLY == "segnet" else ((8, None), (64, None), (64, torch.bfloat16))):
    grid, space, path = pairs(batch // 8, 8, 1)
    net = AEViT(1, 1, img_resolution=R, dim=24).to(dev)
    opt = train.gennet_optimizer(net); sch = train.PolyLR(opt, 1000)
    dt = timeit(lambda: train.gennet_train_step(net, opt, sch, space, path, amp_dtype=amp), 10)
    print("GenNet train step: batch %3d %s  %.1f ms  %.0f maps/s" % (batch, "bf16 autocast" if amp else "fp32", dt * 1e3, batch / dt))

for batch in (() if ONLY == "gennet" else (8,)):
    grid, space, path = pairs(1, batch, 2)
    seg = SegNet().to(dev)
    trainer = train.segnet_trainer(seg)
    opt = train.segnet_optimizer(trainer)
    it = [0]
    def step():
        it[0] += 1
        return train.segnet_train_step(trainer, opt, it[0], 160000, grid, space)
    dt = timeit(step, 5)
    print("SegNet (DiNAT-B + SETR-UP) train step: batch %d fp32  %.1f ms  %.1f images/s" % (batch, dt * 1e3, batch / dt))
