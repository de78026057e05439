#!/usr/bin/env python3
"""Time a batch of both networks with the TunableOp table in use (PPNET_TUNED_TABLE overrides the shipped one), or — with
`fresh OUT.csv` — record a table from scratch on the CURRENT call pattern (addmm_ with beta = 1, _addmm_activation with the
GELU epilogue, virtual padding's token counts) and time a batch with it.
    python tools/retune_gemms.py [R]             python tools/retune_gemms.py fresh OUT.csv [R] [fp32]  (R = 256 or 512, batch 256;
fp32: the float32 model of the reference-precision leg)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.cuda.tunable as tn
from ppnet_amd import edage
import ppnet_amd.ppnet as PP

fresh = len(sys.argv) > 2 and sys.argv[1] == "fresh"
R = int(sys.argv[3]) if fresh and len(sys.argv) > 3 else (int(sys.argv[1]) if not fresh and len(sys.argv) > 1 else 256)
dev = torch.device("cuda:0")
pb = edage.generate_paths(16, R, 50, 3, seed=0, device=dev)
mb = edage.generate_maps(pb, 16, 5, 20, seed=0)
g = mb.grid[:256].contiguous()


def bench(m, n=5):
    for _ in range(2):
        m.heatmap(m.segment_u8(g))
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        m.heatmap(m.segment_u8(g))
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


if fresh:
    def tune_from_scratch():
        tn.enable(True)
        tn.tuning_enable(True)
        tn.set_max_tuning_duration(30)
        tn.set_max_tuning_iterations(20)
        tn.set_filename(sys.argv[2] + ".tunableop")
    PP._use_tuned_gemms = tune_from_scratch
torch.manual_seed(0)
FP32 = "fp32" in sys.argv
m = (PP.PPNet(R, weights_dtype=None) if FP32 else PP.PPNet(R)).to(dev).eval()
if fresh:
    t0 = time.perf_counter()
    m.heatmap(m.segment_u8(g))
    torch.cuda.synchronize()
    print(f"tuning pass: {time.perf_counter() - t0:.1f} s", flush=True)
    tn.tuning_enable(False)
print(f"{'fresh' if fresh else os.environ.get('PPNET_TUNED_TABLE', 'shipped')} table: {bench(m):.3f} ms per batch (both networks)", flush=True)
if fresh:
    res = tn.get_results()
    with open(sys.argv[2], "w") as f:
        for k, v in tn.get_validators():
            f.write(f"Validator,{k},{v}\n")
        for r in res:
            f.write(",".join(str(t) for t in r) + "\n")
    print("entries:", len(res))
