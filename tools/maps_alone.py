"""Stage B alone (no stage A beside it): ms per launch of ppn_edage_maps at the bench configuration, and the share of maps whose
corridor-compose pass ran (diagnostic; PPNET_HIP_LIB selects the build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppnet_amd import edage, _lib
dev = torch.device("cuda:0")
pb = edage.generate_paths(100, 256, 50, 3, seed=0, device=dev)
mb = edage.MapsBatch(10000, 256, 20, dev)
for it in range(300):
    edage.generate_maps(pb, 100, 5, 20, seed=0, first_map_id=it * 10000, out=mb)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for it in range(500):
    edage.generate_maps(pb, 100, 5, 20, seed=0, first_map_id=(300 + it) * 10000, out=mb)
e1.record(); torch.cuda.synchronize()
f = ((mb.flags & _lib.FLAG_CORRIDOR_PASS) != 0).float().mean().item()
print(f"{os.environ.get('PPNET_HIP_LIB', 'default').split('/')[-1]}: {e0.elapsed_time(e1) / 500:.4f} ms per launch, compose on {100 * f:.2f} % of maps, mean obstacles {mb.n_obstacles[:, 0].float().mean().item():.2f}")
