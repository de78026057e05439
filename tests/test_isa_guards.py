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


def test_nat_mlp_no_lds_write_reaches_a_barrier_unwaited(tmp_path):
    """nat_mlp_kernel (csrc/nat_mlp.hip) publishes a pair's GELU(P) halves with a ds_write that the partner reads right behind the
    next s_barrier.  gfx950's barrier does not wait for the LDS queue and hipcc adds no wait of its own, so on every straight-line
    run of emitted code that ends in an s_barrier, a ds_write must be followed by `s_waitcnt lgkmcnt(0)` before the barrier
    (the per-chunk barrier had only a vmcnt wait in front)."""
    asm = _device_asm(tmp_path, "nat_mlp.hip")
    m = re.search(r"^(_ZN3ppn4nmlp\d+nat_mlp_kernel\w*):[^\n]*\n(.*?)s_endpgm", asm, re.S | re.M)
    assert m, "nat_mlp_kernel"
    body = [l.split(";")[0].strip() for l in m.group(2).splitlines()]
    body = [l for l in body if l]
    barriers = [i for i, l in enumerate(body) if l.startswith("s_barrier")]
    assert len(barriers) >= 8, len(barriers)                       # four bodies x (chunk barrier + block-end barrier), + the prologue
    def waits_lgkm0(l):
        # `s_waitcnt lgkmcnt(0)` alone or combined with a vmcnt field
        return l.startswith("s_waitcnt") and re.search(r"lgkmcnt\(0\)", l) is not None
    labels = {l[:-1]: i for i, l in enumerate(body) if l.endswith(":")}
    jumps = {}
    for i, l in enumerate(body):
        m2 = re.match(r"s_c?branch\w*\s+(\S+)", l)
        if m2:
            jumps.setdefault(m2.group(1), []).append(i)

    def unwaited_write_reaches(end, seen):
        """True if some path into body[end] carries a ds_write with no lgkmcnt(0) wait behind it (paths followed backwards through
        fall-through and through every branch that targets a label on the way; an earlier barrier's own check covers what is in front of it)."""
        for i in range(end - 1, -1, -1):
            l = body[i]
            if waits_lgkm0(l) or l.startswith("s_barrier"):
                return False
            if l.startswith("ds_write") or l.startswith("ds_store"):
                return True
            if l.startswith("s_branch"):                        # unconditional: nothing falls through from above
                return False
            if l.endswith(":"):
                for j in jumps.get(l[:-1], []):
                    if j not in seen:
                        seen.add(j)
                        if unwaited_write_reaches(j, seen):
                            return True
        return False

    for b in barriers:
        assert not unwaited_write_reaches(b, set()), f"a ds_write reaches the s_barrier at instruction {b} without s_waitcnt lgkmcnt(0)"
    heads = [i for i in barriers if i > 0 and waits_lgkm0(body[i - 1])]
    assert len(heads) >= 8, (len(heads), len(barriers))


@pytest.mark.parametrize("src,kernel,instances", [("na2d_halo16.hip", r"_ZN3ppn18na2d_halo16_kernel", 2), ("na2d_dense7.hip", r"_ZN3ppn18na2d_dense7_kernel", 2)])
def test_attention_inline_asm_never_reads_a_fresh_mfma_result(tmp_path, src, kernel, instances):
    """na2d_halo16_kernel's and na2d_dense7_kernel's row maxima are inline assembly (`v_max3_f32`, which unlike fmaxf does not canonicalise its operands
    first).  The compiler's hazard recogniser does not look inside inline assembly, and gfx950 has no hardware interlock between an
    XDL write and a VALU read: a v_max3 issued fewer than 11 wait states behind the v_mfma_f32_16x16x32_bf16 that writes one of its
    sources reads the register's OLD content (round 5: hipcc had interleaved them, the maxima were K-fragment bit patterns — harmless
    to the softmax only while the logits are small).  The kernel waits explicitly; this checks the emitted code, both block shapes."""
    asm = _device_asm(tmp_path, src)
    kernels = re.findall(r"^(" + kernel + r"\w*):[^\n]*\n(.*?)s_endpgm", asm, re.S | re.M)
    assert len(kernels) == instances, [k for k, _ in kernels]

    def regs(tok):
        tok = tok.strip().rstrip(",")
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if m:
            return set(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.fullmatch(r"v(\d+)", tok)
        return {int(m.group(1))} if m else set()

    for name, text in kernels:
        since = {}                                                  # register -> wait states since the MFMA that last wrote it
        checked = 0
        for line in text.splitlines():
            l = line.split(";")[0].strip()
            if not l or l.endswith(":") or l.startswith("."):
                continue
            op, _, rest = l.partition(" ")
            ops = [t for t in rest.split(",")] if rest else []
            if op.startswith("v_max3_f32"):
                for t in ops[1:]:
                    for r in regs(t):
                        assert since.get(r, 99) >= 11, (name, l, r, since.get(r))
                checked += 1
            step = int(rest) + 1 if op == "s_nop" else 1
            for r in since:
                since[r] += step
            if op.startswith("v_mfma"):
                for r in regs(ops[0]):
                    since[r] = 0
            elif op.startswith("v_") and ops:
                for r in regs(ops[0]):
                    since.pop(r, None)                              # rewritten by an ordinary instruction (the compiler's own hazards)
        assert checked >= 6, (name, checked)
