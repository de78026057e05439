"""Times the 128-channel token-streaming kernels alone (1 M tokens = a batch of 256 at level 0)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from ppnet_amd import fused

tokens = 1 << 20
ln = nn.LayerNorm(128).cuda().bfloat16()
qkv, fc1, fc2 = nn.Linear(128, 384).cuda().bfloat16(), nn.Linear(128, 256).cuda().bfloat16(), nn.Linear(256, 128).cuda().bfloat16()
s = torch.randn(tokens, 128, device="cuda").bfloat16()
off = torch.zeros(128, device="cuda")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


print("ln+qkv  fused %.3f ms" % timeit(lambda: fused.nat128_ln_qkv(s, off, ln, qkv)))
print("ln+mlp  fused %.3f ms" % timeit(lambda: fused.nat128_ln_mlp_(s, off, ln, fc1, fc2)))
y = fused.layer_norm(s, ln, offset=off)
print("ln      alone %.3f ms" % timeit(lambda: fused.layer_norm(s, ln, offset=off)))
print("qkv     lib   %.3f ms" % timeit(lambda: qkv(y)))
h = torch._addmm_activation(fc1.bias, y, fc1.weight.t(), use_gelu=True)
print("fc1gelu lib   %.3f ms" % timeit(lambda: torch._addmm_activation(fc1.bias, y, fc1.weight.t(), use_gelu=True)))
print("fc2     lib   %.3f ms" % timeit(lambda: s.addmm_(h, fc2.weight.t())))
