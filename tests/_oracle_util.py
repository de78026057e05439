"""Helpers shared by the parity tests: run the CPU oracle and lay its results out like the HIP
library's output tensors.  Test infrastructure only."""
import numpy as np

from oracle import edage_np as E


def fixed_layout(compact, straight):
    """reference consumption order (golden fixtures) -> the library's fixed draw layout."""
    full = np.ones(E.DRAWS_PER_PATH)
    full[0] = 0.0 if straight else 1.0
    pos = 0
    for s in range(E.PATHSEGNUM):
        b = 1 + s * E.DRAWS_PER_SEG
        if not straight:
            full[b] = compact[pos]
            pos += 1
        full[b + 1:b + 1 + E.N_FIT] = compact[pos:pos + E.N_FIT]
        pos += E.N_FIT
        full[b + 1 + E.N_FIT] = compact[pos]
        pos += 1
    assert pos == len(compact)
    return full


def bits_to_mask(words, h, w):
    words = np.asarray(words).astype(np.uint32).reshape(-1)
    bits = ((words[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).astype(bool)
    return bits.reshape(h, w)


def cyclic_equal(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return False
    for k in range(len(a)):
        if np.array_equal(np.roll(a, -k, axis=0), b):
            return True
    return False


def oracle_paths(seed, n, R, map_size, clearance, first_path_id=0):
    return E.generate_paths(E.PhiloxSource(seed), n, R, map_size, clearance, first_path_id=first_path_id)


def oracle_maps(seed, precs, R, map_size, obstacles_size, K, clearance, placements, first_map_id=0):
    return E.generate_maps(E.PhiloxSource(seed), precs, R, map_size, obstacles_size, K, clearance, placements,
                           first_map_id=first_map_id)


def wiring_weights(keys, shapes, seed):
    """The weights of the NAT-wiring golden (tests/golden/g16_nat_wiring.npz): one deterministic array per state-dict key, from the
    key's own legacy MT19937 stream (np.random.RandomState: frozen across NumPy versions), scaled by kind so that every branch of
    the network carries signal — LayerScale 0.2..1.2 (the default 1e-5 would hide the residual branches), norms' weights around 1,
    biases +-0.2, relative-position biases N(0, 0.5), matrices N(0, fan_in^-1/2).  The fixture stores the key list, the shapes and
    a (sum, sum of squares) checksum per key instead of ~5 MB of random numbers; tests/golden/make_fixtures.py loads exactly these
    arrays into the REFERENCE's own module."""
    import zlib
    out = {}
    for k, shp in zip(keys, shapes):
        rs = np.random.RandomState((zlib.crc32(k.encode()) + 7919 * seed) & 0x7FFFFFFF)
        shp = tuple(int(v) for v in shp)
        if k.endswith("gamma1") or k.endswith("gamma2"):
            a = rs.uniform(0.2, 1.2, shp)
        elif k.endswith("rpb"):
            a = rs.normal(0.0, 0.5, shp)
        elif len(shp) == 1 and k.endswith("weight"):
            a = rs.uniform(0.7, 1.3, shp)
        elif len(shp) == 1:
            a = rs.uniform(-0.2, 0.2, shp)
        else:
            fan_in = int(np.prod(shp[1:]))
            a = rs.normal(0.0, fan_in ** -0.5, shp)
        out[k] = np.ascontiguousarray(a, dtype=np.float64)
    return out
