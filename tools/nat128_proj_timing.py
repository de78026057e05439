import sys, torch
sys.path.insert(0, "/root/repo")
from ppnet_amd import fused
dev = torch.device("cuda:0")
T = 256 * 64 * 64
s = torch.randn(T, 128, device=dev, dtype=torch.bfloat16); a = torch.randn_like(s)
lin = torch.nn.Linear(128, 128, bias=False).to(dev).to(torch.bfloat16)
for _ in range(5): fused.nat128_proj_add_(s, a, lin)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): fused.nat128_proj_add_(s, a, lin)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"proj_add 1M tokens: {ms:.4f} ms  {T * 768 / ms / 1e9:.2f} TB/s")
