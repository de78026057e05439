"""Diagnostic: where a workgroup of the MFMA neighbourhood-attention kernel spends its cycles (build: make -C ppnet_amd/csrc timing)."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["PPNET_HIP_LIB"] = os.path.join(ROOT, "ppnet_amd", "libppnet_hip_timing.so")
from ppnet_amd import _lib as L
from ppnet_amd.na import na2d_forward
dev = torch.device("cuda", 0)
buf = (C.c_ulonglong * 8)()
for side, Cc, heads in ((64, 128, 4), (32, 256, 8), (16, 512, 16)):
    qkv = torch.randn(256, side, side, 3 * Cc, device=dev, dtype=torch.bfloat16)
    rpb = torch.randn(heads, 13, 13, device=dev)
    for _ in range(3):
        na2d_forward(qkv, rpb, heads, 1, 32 ** -0.5)
    torch.cuda.synchronize()
    L.lib.ppn_debug_na_phase_cycles(buf, 1)
    na2d_forward(qkv, rpb, heads, 1, 32 ** -0.5)
    torch.cuda.synchronize()
    L.lib.ppn_debug_na_phase_cycles(buf, 1)
    n = 256 * (side // 16) ** 2 * heads
    names = ["setup + load issue", "loads return", "LDS writes + barrier", "block 0", "block 1"]
    tot = sum(buf[:5])
    print(f"side {side}: " + ", ".join(f"{nm} {buf[i] / n:.0f}" for i, nm in enumerate(names)) + f"  = {tot / n:.0f} cycles per workgroup")
