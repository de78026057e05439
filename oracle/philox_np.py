"""TEST INFRASTRUCTURE — CPU oracle, not product code.

Philox4x32-10 counter-based generator (Salmon et al., "Parallel random numbers: as easy as
1, 2, 3", SC'11; the Random123 reference constants), restated in NumPy so that the oracle
and the HIP kernels (ppnet_amd/csrc/ppn_philox.h) draw identical numbers in throughput mode.

The reference itself draws from the global MT19937 stream (`np.random.random`,
PathSeg.py:19,23,32; MapGenerate.py:63-64,128-130) and from torch's global generator
(`torch.rand`, Path.py:479-485); a data-dependent global stream cannot be consumed in
parallel, so the MI355X path keys every draw by (seed, stream, instance, index) instead.
The *algorithm* is pinned against the reference with MT-fed draws (tests/test_oracle_golden.py);
the *kernels* are pinned against this oracle with Philox draws.

Draw recipes (same bit recipes NumPy/torch apply to their MT words):
  double : ((w0 >> 5) * 2**26 + (w1 >> 6)) / 2**53        (numpy random_sample recipe)
  float32: (w & 0xFFFFFF) / 2**24                          (torch.rand CPU recipe)
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)

# stream tags (counter word 3)
STREAM_PATH = 1      # stage A numpy-style doubles, instance = path id
STREAM_POCKET = 2    # stage A torch-style float32 (set_obstacles), instance = path id
STREAM_PLACE = 3     # stage B placement attempts, instance = map id
STREAM_OBST = 4      # stage B obstacle draws, instance = map id


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10. All inputs broadcastable integer arrays; returns 4 uint32 arrays."""
    c0 = np.asarray(c0, dtype=np.uint64) & MASK32
    c1 = np.asarray(c1, dtype=np.uint64) & MASK32
    c2 = np.asarray(c2, dtype=np.uint64) & MASK32
    c3 = np.asarray(c3, dtype=np.uint64) & MASK32
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for r in range(10):
        if r:
            k0 = (k0 + W0) & 0xFFFFFFFF
            k1 = (k1 + W1) & 0xFFFFFFFF
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)), lo1, (hi0 ^ c3 ^ np.uint64(k1)), lo0
    return (c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32))


def _key(seed):
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return seed & 0xFFFFFFFF, seed >> 32


def doubles(seed, stream, instance, first, count):
    """`count` uniform doubles in [0,1): draw index d = first..first+count-1 of (stream, instance).

    Draw d uses Philox block d>>1 (counter = (d>>1, 0, instance, stream)); d even -> words (0,1),
    d odd -> words (2,3)."""
    d = np.arange(first, first + count, dtype=np.uint64)
    k0, k1 = _key(seed)
    inst = np.uint64(int(instance) & 0xFFFFFFFF)
    o0, o1, o2, o3 = philox4x32_10(d >> np.uint64(1), int(instance) >> 32, inst, stream, k0, k1)
    odd = (d & np.uint64(1)).astype(bool)
    a = np.where(odd, o2, o0).astype(np.uint64) >> np.uint64(5)
    b = np.where(odd, o3, o1).astype(np.uint64) >> np.uint64(6)
    return (a * np.uint64(67108864) + b).astype(np.float64) / 9007199254740992.0


def floats(seed, stream, instance, first, count):
    """`count` uniform float32 in [0,1): draw d uses block d>>2, word d&3."""
    d = np.arange(first, first + count, dtype=np.uint64)
    k0, k1 = _key(seed)
    inst = np.uint64(int(instance) & 0xFFFFFFFF)
    o = philox4x32_10(d >> np.uint64(2), int(instance) >> 32, inst, stream, k0, k1)
    w = np.choose((d & np.uint64(3)).astype(np.int64), o)
    return ((w & np.uint32(0xFFFFFF)).astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(np.float32)
