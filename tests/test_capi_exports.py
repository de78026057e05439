"""CPU-side checks of the C-ABI shared library: it loads, exports every function that
include/ppnet_hip.h declares, rejects bad arguments without touching a GPU, and its host-side
polyfit operator reproduces np.polyfit (the per-call LAPACK solve at PathSeg.py:24)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "ppnet_amd", "libppnet_hip.so")


def _ensure_built():
    if not os.path.exists(LIB):
        import __graft_entry__ as g
        g.build()
    return C.CDLL(LIB)


def _declared():
    src = open(os.path.join(ROOT, "include", "ppnet_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ppn_[a-z_0-9]+)\s*\(", src)))


def test_header_functions_are_exported():
    lib = _ensure_built()
    names = _declared()
    assert {"ppn_edage_paths", "ppn_edage_maps", "ppn_boundary_check", "ppn_disc_raster", "ppn_collision_segments",
            "ppn_extract_paths", "ppn_polyfit_table", "ppn_version"} <= set(names)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ppnet_hip.h but not exported"


def test_binding_matches_header():
    from ppnet_amd import _lib
    assert sorted(_lib.EXPORTS) == _declared()
    header = open(os.path.join(ROOT, "include", "ppnet_hip.h")).read()
    abi = int(re.search(r"#define\s+PPN_ABI_VERSION\s+(\d+)", header).group(1))
    assert _lib.lib.ppn_version() == abi == _lib.ABI_VERSION          # library, header and bindings agree
    assert _lib.lib.ppn_error_string(-1) == b"invalid argument"
    assert C.sizeof(_lib.PathsStruct) == 28 * C.sizeof(C.c_void_p)
    assert C.sizeof(_lib.MapsStruct) == 11 * C.sizeof(C.c_void_p)


def test_invalid_arguments_return_codes_without_gpu():
    from ppnet_amd import _lib
    L = _lib.lib
    assert L.ppn_polyfit_table(None) == -1
    ps, ms = _lib.PathsStruct(), _lib.MapsStruct()
    assert L.ppn_edage_paths(1, 0, 100, 50.0, 3.0, 0, None, None, 0, C.byref(ps), None) == -1     # R % 32
    assert L.ppn_edage_paths(1, 0, 64, 50.0, 3.0, 0, None, None, 0, C.byref(ps), None) == -1      # NULL outputs
    assert L.ppn_edage_maps(C.byref(ps), 1, 1, 0, 64, 50.0, 5.0, 300, 3.0, 0, None, None, C.byref(ms), None) == -1
    assert L.ppn_boundary_check(None, 0, None, None, 0, 64, None, None) == -1
    assert L.ppn_extract_paths(None, 1, 8, 8, None, None, 16, None, None, None, None) == -1


def test_network_entry_points_reject_bad_arguments_without_gpu():
    """The round-4 entries of the SegNet heads / NAT levels: argument checks come before any HIP call."""
    from ppnet_amd import _lib
    L = _lib.lib
    one = C.c_void_p(0x1000)                                  # a non-null pointer that is never dereferenced on these paths
    hw8 = (C.c_int32 * 8)(16, 16, 8, 8, 4, 4, 2, 2)
    assert L.ppn_resize_concat4_nhwc(None, one, one, one, hw8, one, 1, 64, 1, None) == -1
    assert L.ppn_resize_concat4_nhwc(one, one, one, one, hw8, one, 1, 60, 1, None) == -1          # C % 8
    assert L.ppn_resize_concat4_nhwc(one, one, one, one, hw8, one, 1, 64, 2, None) == -1          # dtype
    ptrs = (C.c_void_p * 2)(0x1000, 0x1000)
    assert L.ppn_resize_concat_nhwc(ptrs, (C.c_int32 * 4)(8, 8, 0, 2), (C.c_int32 * 2)(64, 64), 2, one, 1, 1, None) == -1     # empty level
    assert L.ppn_resize_concat_nhwc(ptrs, (C.c_int32 * 4)(8, 8, 2, 2), (C.c_int32 * 2)(64, 12), 2, one, 1, 1, None) == -1     # channels % 8
    assert L.ppn_resize_concat_nhwc(ptrs, (C.c_int32 * 4)(8, 8, 2, 2), (C.c_int32 * 2)(64, 64), 9, one, 1, 1, None) == -1     # > 8 levels
    assert L.ppn_adaptive_pools_nhwc(one, ptrs, (C.c_int32 * 2)(1, 0), 2, 1, 8, 8, 64, 1, None) == -1                        # scale 0
    assert L.ppn_adaptive_pools_nhwc(one, ptrs, (C.c_int32 * 2)(1, 2), 5, 1, 8, 8, 64, 1, None) == -1                        # > 4 scales
    assert L.ppn_upsample2x_add_nhwc(one, None, one, 1, 8, 8, 64, 1, None) == -1
    assert L.ppn_upsample2x_add_nhwc(one, one, one, 1, 8, 8, 60, 1, None) == -1
    assert L.ppn_nat_mlp_bf16(None, one, one, one, None, 256, 256, 512, C.c_float(1e-5), None) == -1
    assert L.ppn_nat_mlp_supported(256, 512, 1024) == 0 and L.ppn_nat_mlp_supported(256, 256, 512) == 1


def test_polyfit_operator_matches_numpy():
    from ppnet_amd import _lib
    W = np.zeros((4, 1000))
    assert _lib.lib.ppn_polyfit_table(W.ctypes.data_as(C.c_void_p)) == 0
    x = np.arange(1000) / 100
    rng = np.random.RandomState(0)
    for _ in range(5):
        y = rng.random_sample(1000) * 10 - 5
        ref = np.polyfit(x, y, 4)
        got = W @ y
        assert np.abs(got - ref[:4]).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    # exactness on polynomials: fitting p(x) returns p
    p = np.array([0.01, -0.2, 0.5, 1.5, 0.0])
    assert np.abs(W @ np.polyval(p, x) - p[:4]).max() < 1e-11
