"""Build-time guards on the generated gfx950 code of kernels whose correctness or speed rests on what the compiler emitted (CPU suite:
hipcc cross-compiles here; nothing runs).

nat128_ln_qkv_kernel (csrc/nat_c128.hip) requests the NEXT group's rows from inline assembly and waits for them with a COUNTED
`s_waitcnt vmcnt(12)`: the count is the number of vector-memory instructions the wave issues between those loads and the wait, and the
sixteen destination registers must not be touched in between (the compiler does not know the loads are pending).  Both are properties
of the emitted code, so they are checked on the emitted code.  nat128_proj_add_kernel lost 45 % of its speed to loop-invariant LDS reads
the compiler hoisted and spilled: no kernel of this file may use scratch."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ppnet_amd", "csrc")
HIPCC = os.environ.get("HIPCC") or "/opt/rocm/bin/hipcc"


def _device_asm(tmp_path, src):
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("hipcc not present")
    out = tmp_path / (src + ".s")
    flags = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize --cuda-device-only -S".split()
    subprocess.run([HIPCC, *flags, os.path.join(CSRC, src), "-o", str(out)], check=True, cwd=CSRC, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out.read_text()


def _kernel(asm, name):
    m = re.search(r"^(_ZN3ppn\d+" + name + r"\w*):[^\n]*\n(.*?)s_endpgm", asm, re.S | re.M)
    assert m, name
    return m.group(2)


def test_nat128_kernels_use_no_scratch_and_qkv_wait_count_matches(tmp_path):
    asm = _device_asm(tmp_path, "nat_c128.hip")
    scratch = dict(re.findall(r"\.amdhsa_kernel (\S+).*?; ScratchSize: (\d+)", asm, re.S))
    assert len(scratch) == 3 and all(int(v) == 0 for v in scratch.values()), scratch
    body = [l.split(";")[0].strip() for l in _kernel(asm, "nat128_ln_qkv_kernel").splitlines()]
    body = [l for l in body if l]
    head = max(i for i, l in enumerate(body) if l.startswith("s_waitcnt vmcnt(12)"))          # the token loop's head
    back = next(i for i in range(head, len(body)) if body[i].startswith("s_cbranch") and i > head + 50)
    loop = body[head:back]
    loads = [i for i, l in enumerate(loop) if l.startswith("global_load_dwordx4")]
    assert len(loads) == 4 and loads[-1] - loads[0] == 3, loads                              # the asm block: four loads back to back
    after = loop[loads[-1] + 1:]
    vmem = [l for l in after if re.match(r"(global|buffer|scratch|flat)_", l)]
    assert len(vmem) == 12 and all(l.startswith("global_store_dwordx4") for l in vmem), vmem  # what vmcnt(12) leaves in flight
    dest = set()
    for l in loop[loads[0]:loads[-1] + 1]:
        lo, hi = map(int, re.match(r"global_load_dwordx4 v\[(\d+):(\d+)\]", l).groups())
        dest |= set(range(lo, hi + 1))
    assert len(dest) == 16
    for l in after:                                                                           # nothing reads or writes them before the wait
        regs = set(int(r) for r in re.findall(r"\bv(\d+)\b", l))
        for lo, hi in re.findall(r"v\[(\d+):(\d+)\]", l):
            regs |= set(range(int(lo), int(hi) + 1))
        assert not (regs & dest), l
    # the first group waits for everything: nothing is behind its loads to count
    pre = body[:head]
    first = max(i for i, l in enumerate(pre) if l.startswith("global_load_dwordx4"))
    assert any(l.startswith("s_waitcnt vmcnt(0)") for l in pre[first:]), pre[first:]
