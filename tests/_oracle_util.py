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
