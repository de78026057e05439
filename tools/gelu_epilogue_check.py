"""Diagnostic: what GELU does torch._addmm_activation(use_gelu=True) compute on this stack (erf or tanh form)?"""
import torch
torch.manual_seed(0)
x = torch.randn(4096, 512, device="cuda", dtype=torch.bfloat16)
w = torch.randn(2048, 512, device="cuda", dtype=torch.bfloat16) * 0.05
b = torch.randn(2048, device="cuda", dtype=torch.bfloat16)
y = torch._addmm_activation(b, x, w.t(), use_gelu=True)
pre = torch.addmm(b.float(), x.float(), w.float().t())
for name, ref in (("erf", torch.nn.functional.gelu(pre)), ("tanh", torch.nn.functional.gelu(pre, approximate="tanh"))):
    print(name, "max abs diff", (y.float() - ref).abs().max().item(), "mean", (y.float() - ref).abs().mean().item())
sep = torch.nn.functional.gelu(torch.nn.functional.linear(x, w, b))
print("separate bf16 linear+gelu vs erf ref", (sep.float() - torch.nn.functional.gelu(pre)).abs().max().item(), (sep.float() - torch.nn.functional.gelu(pre)).abs().mean().item())
import time
for f, n in ((lambda: torch._addmm_activation(b, x, w.t(), use_gelu=True), "fused"), (lambda: torch.nn.functional.gelu(torch.nn.functional.linear(x, w, b)), "separate")):
    for _ in range(5): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): f()
    torch.cuda.synchronize(); print(n, (time.perf_counter() - t) / 50 * 1e3, "ms")
