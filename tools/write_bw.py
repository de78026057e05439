"""Diagnostic: achievable pure-store bandwidth (torch fill of 823 MB, the bytes one maps launch writes)."""
import torch
dev = torch.device("cuda:0")
x = torch.empty(823 * 1024 * 1024 // 4, dtype=torch.int32, device=dev)
for _ in range(3): x.fill_(1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(20): x.fill_(i)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"fill 823 MiB: {ms:.4f} ms -> {x.numel() * 4 / ms / 1e9:.2f} TB/s")
y = torch.empty_like(x)
e0.record()
for i in range(20): y.copy_(x)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"copy 823 MiB: {ms:.4f} ms -> {2 * x.numel() * 4 / ms / 1e9:.2f} TB/s (read+write)")
