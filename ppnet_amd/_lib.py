"""ctypes binding of libppnet_hip.so (C ABI in include/ppnet_hip.h).

The HIP library is the product path: there is no CPU fallback.  If the shared object is missing
the import fails loudly with the build command.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PPNET_HIP_LIB points at another build of the same library (A/B runs of kernel variants); there is no non-HIP fallback
LIB_PATH = os.environ.get("PPNET_HIP_LIB") or os.path.join(_HERE, "libppnet_hip.so")

PPN_OK = 0
SEGS = 10
PATH_POINTS = 1000
BOUNDARY_POINTS = 1100
DRAWS_PER_SEG = 1002
DRAWS_PER_PATH = 1 + SEGS * DRAWS_PER_SEG
MAX_HULL = 64
MAX_ISLES = 16
MAX_POCKET = 64
POCKET_TRY_CAP = 256
PLACE_TRY_CAP = 4096
MAX_WAYPOINTS = 2048
GRID_OBST, GRID_FREE, GRID_MARK = 0, 255, 128
FLAG_POCKET_CAP, FLAG_PLACE_CAP, FLAG_EMPTY_ISLE, FLAG_HULL_CAP, FLAG_ISLE_CAP, FLAG_POCKET_FULL = 1, 2, 4, 8, 16, 32
FLAG_CORRIDOR_PASS = 64   # informational (maps only): the raster kernel ran its corridor compose pass
FLAG_POCKET_DRAWS = 128   # the caller-fed pocket draws ran out (ppn_edage_paths: pocket_stride too small)

_p = C.c_void_p


class PathsStruct(C.Structure):
    """ppn_paths_t"""
    _fields_ = [(n, _p) for n in (
        "seg_poly", "seg_endpoint", "seg_rotation", "seg_translation", "seg_length", "seg_straight",
        "segpoint_world", "pathpoint_world", "boundary_world", "canvas_bits", "hull_raw", "hull", "hull_n",
        "rotation", "trans_rc", "segpoint_image", "pathpoint_image", "space_bits", "isles", "n_isles",
        "obstacles", "n_obstacles", "length", "straight", "flags", "max_step_px", "seg_grad", "pocket_draws_used")]


class MapsStruct(C.Structure):
    """ppn_maps_t"""
    _fields_ = [(n, _p) for n in (
        "grid", "angle", "translation", "attempts", "segpoint", "pathpoint", "accept", "obstacles",
        "n_obstacles", "flags", "records")]


ABI_VERSION = 105          # include/ppnet_hip.h: PPN_ABI_VERSION (tests/test_capi_exports.py keeps the two equal)

EXPORTS = ("ppn_version", "ppn_error_string", "ppn_last_hip_error", "ppn_polyfit_table", "ppn_edage_paths",
           "ppn_edage_paths_ex", "ppn_edage_paths_ex2", "ppn_edage_maps", "ppn_edage_maps_place", "ppn_edage_maps_raster",
           "ppn_label_masks", "ppn_boundary_check", "ppn_boundary_check_ex",
           "ppn_obstacle_filter", "ppn_paint_markers", "ppn_disc_raster", "ppn_collision_segments", "ppn_collision_segments_bound",
           "ppn_extract_paths", "ppn_resize_bilinear_u8", "ppn_philox_doubles", "ppn_na2d_fwd", "ppn_na2d_fwd_padded", "ppn_na2d_fwd_vpad", "ppn_na2d_bwd", "ppn_na2d_bwd_workspace", "ppn_residual_layernorm", "ppn_residual_layernorm_padded", "ppn_layernorm_offset", "ppn_upsample2x_nhwc", "ppn_resize_concat4_nhwc", "ppn_resize_concat_nhwc", "ppn_adaptive_pools_nhwc", "ppn_upsample2x_add_nhwc", "ppn_upsample2x_nhwc_bias", "ppn_bias_act_nhwc", "ppn_seg_labels_2class", "ppn_grid_to_image", "ppn_conv3x3_c1_nhwc", "ppn_conv3x3_to1_nhwc",
           "ppn_conv3x3_mfma_bf16", "ppn_conv3x3_relu_classify2_bf16", "ppn_conv3x3_relu_classify2_slots", "ppn_gemm_bf16", "ppn_nat_gemm_bf16", "ppn_nat_gemm_partials", "ppn_row_stats_bf16", "ppn_nat_mlp_supported", "ppn_nat_mlp_pack_bf16", "ppn_nat_mlp_bf16", "ppn_gennet_conv_s2_bf16", "ppn_gennet_trunk_bf16",
           "ppn_assemble_paths", "ppn_plan_collision", "ppn_gennet_first_enc_bf16", "ppn_gennet_dec_final_bf16", "ppn_heatmap_u8", "ppn_tokenizer_conv1_codes_bf16", "ppn_tokenizer_codes_bf16", "ppn_nat128_ln_qkv_bf16", "ppn_nat128_ln_mlp_bf16", "ppn_nat128_ln_mlp_add_bf16", "ppn_nat128_proj_add_bf16")


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension is the only implementation of this package "
            f"(no CPU fallback). Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C ppnet_amd/csrc`.")
    lib = C.CDLL(LIB_PATH)
    lib.ppn_version.restype = C.c_int
    if lib.ppn_version() != ABI_VERSION:
        raise ImportError(
            f"{LIB_PATH} reports ABI version {lib.ppn_version()}, these bindings are written against {ABI_VERSION} "
            f"(include/ppnet_hip.h: PPN_ABI_VERSION): argument lists differ, rebuild with `make -C ppnet_amd/csrc`.")
    lib.ppn_error_string.restype = C.c_char_p
    lib.ppn_error_string.argtypes = [C.c_int]
    lib.ppn_last_hip_error.restype = C.c_int
    lib.ppn_polyfit_table.argtypes = [_p]
    lib.ppn_edage_paths.argtypes = [C.c_int32, C.c_uint64, C.c_int32, C.c_double, C.c_double, C.c_uint64,
                                    _p, _p, C.c_int32, C.POINTER(PathsStruct), _p]
    lib.ppn_edage_paths_ex.argtypes = [C.c_int32, C.c_uint64, C.c_int32, C.c_double, C.c_double, C.c_uint64,
                                       _p, _p, C.c_int32, _p, C.POINTER(PathsStruct), _p]
    lib.ppn_edage_paths_ex2.argtypes = [C.c_int32, C.c_uint64, C.c_int32, C.c_double, C.c_double, C.c_uint64,
                                        _p, _p, C.c_int32, _p, _p, C.POINTER(PathsStruct), _p]
    lib.ppn_boundary_check_ex.argtypes = [_p, C.c_int32, _p, _p, C.c_int32, C.c_int32, _p, _p, _p]
    lib.ppn_obstacle_filter.argtypes = [_p, _p, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double,
                                        _p, _p, _p, _p]
    lib.ppn_paint_markers.argtypes = [_p, C.c_int32, C.c_int32, _p, _p, _p]
    lib.ppn_edage_maps.argtypes = [C.POINTER(PathsStruct), C.c_int32, C.c_int32, C.c_uint64, C.c_int32,
                                   C.c_double, C.c_double, C.c_int32, C.c_double, C.c_uint64, _p, _p,
                                   C.POINTER(MapsStruct), _p]
    lib.ppn_edage_maps_place.argtypes = lib.ppn_edage_maps.argtypes
    lib.ppn_edage_maps_raster.argtypes = [C.POINTER(PathsStruct), C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                          C.POINTER(MapsStruct), _p]
    lib.ppn_boundary_check.argtypes = [_p, C.c_int32, _p, _p, C.c_int32, C.c_int32, _p, _p]
    lib.ppn_disc_raster.argtypes = [_p, _p, C.c_int32, C.c_int32, C.c_int32, _p, _p]
    lib.ppn_collision_segments.argtypes = [_p, _p, _p, C.c_int32, _p, _p, C.c_float, _p, _p]
    lib.ppn_collision_segments_bound.argtypes = [_p, _p, _p, C.c_int32, _p, _p, C.c_float, C.c_float, _p, _p]
    lib.ppn_extract_paths.argtypes = [_p, C.c_int32, C.c_int32, C.c_int32, _p, _p, C.c_int32, _p, _p, _p, _p]
    lib.ppn_resize_bilinear_u8.argtypes = [_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _p, _p, _p]
    lib.ppn_philox_doubles.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_int32, _p, _p]
    lib.ppn_na2d_fwd.argtypes = [_p, _p, _p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, _p]
    lib.ppn_na2d_bwd.argtypes = [_p, _p, _p, _p, _p, _p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, _p]
    lib.ppn_na2d_bwd_workspace.argtypes = [C.c_int32] * 5
    lib.ppn_na2d_bwd_workspace.restype = C.c_int64
    lib.ppn_residual_layernorm.argtypes = [_p, _p, _p, _p, _p, _p, _p, C.c_int64, C.c_int32, C.c_float, C.c_int32, _p]
    lib.ppn_na2d_fwd_padded.argtypes = [_p, _p, _p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_float, C.c_int32, _p]
    lib.ppn_na2d_fwd_vpad.argtypes = [_p, _p, _p, _p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_float, C.c_int32, _p]
    lib.ppn_residual_layernorm_padded.argtypes = [_p, _p, _p, _p, _p, _p, _p, C.c_int64, C.c_int32, C.c_float, C.c_int32,
                                                  C.c_int32, C.c_int32, C.c_int32, C.c_int32, _p]
    lib.ppn_layernorm_offset.argtypes = [_p, _p, _p, _p, _p, C.c_int64, C.c_int32, C.c_float, C.c_int32, _p]
    lib.ppn_upsample2x_add_nhwc.argtypes = [_p, _p, _p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _p]
    lib.ppn_resize_concat_nhwc.argtypes = [C.POINTER(_p), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, _p, C.c_int32, C.c_int32, _p]
    lib.ppn_adaptive_pools_nhwc.argtypes = [_p, C.POINTER(_p), C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _p]
    lib.ppn_resize_concat4_nhwc.argtypes = [_p, _p, _p, _p, C.POINTER(C.c_int32), _p, C.c_int32, C.c_int32, C.c_int32, _p]
    lib.ppn_upsample2x_nhwc.argtypes = [_p, _p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _p]
    lib.ppn_upsample2x_nhwc_bias.argtypes = [_p, _p, _p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _p]
    lib.ppn_grid_to_image.argtypes = [_p, _p, C.c_int64, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int32, _p]
    lib.ppn_seg_labels_2class.argtypes = [_p, _p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _p]
    lib.ppn_bias_act_nhwc.argtypes = [_p, _p, C.c_int64, C.c_int32, C.c_float, C.c_int32, _p]
    lib.ppn_conv3x3_c1_nhwc.argtypes = [_p, _p, _p, _p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, _p]
    lib.ppn_conv3x3_to1_nhwc.argtypes = [_p, _p, C.c_float, _p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _p]
    lib.ppn_conv3x3_mfma_bf16.argtypes = [_p, _p, _p, _p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _p]
    lib.ppn_conv3x3_relu_classify2_bf16.argtypes = [_p, _p, _p, _p, _p, _p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _p]
    lib.ppn_conv3x3_relu_classify2_slots.argtypes = [C.c_int32]
    lib.ppn_gennet_conv_s2_bf16.argtypes = [_p, _p, _p, _p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, _p]
    lib.ppn_gennet_trunk_bf16.argtypes = [_p, _p, _p, C.c_int32, C.c_int32, C.c_int32, _p]
    lib.ppn_assemble_paths.argtypes = [_p, _p, _p, _p, _p, C.c_double, C.c_int32, C.c_int32, _p, _p, _p]
    lib.ppn_plan_collision.argtypes = [_p, _p, _p, C.c_int32, _p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, _p, _p]
    lib.ppn_gennet_first_enc_bf16.argtypes = [_p, _p, _p, _p, _p, _p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, _p]
    lib.ppn_gennet_dec_final_bf16.argtypes = [_p, _p, _p, C.c_float, _p, C.c_float, _p, C.c_int32, C.c_int32, C.c_int32, _p]
    lib.ppn_heatmap_u8.argtypes = [_p, _p, C.c_int32, C.c_int32, C.c_int32, _p]
    lib.ppn_tokenizer_conv1_codes_bf16.argtypes = [_p, _p, _p, C.c_int32, C.c_int32, C.c_int32, _p]
    lib.ppn_tokenizer_codes_bf16.argtypes = [_p, _p, _p, _p, _p, C.c_int32, C.c_int32, C.c_int32, C.c_float, _p]
    lib.ppn_nat128_ln_qkv_bf16.argtypes = [_p, _p, _p, _p, _p, _p, _p, C.c_int64, C.c_float, _p]
    lib.ppn_nat128_ln_mlp_bf16.argtypes = [_p, _p, _p, _p, _p, _p, _p, C.c_int64, C.c_float, _p]
    lib.ppn_nat128_ln_mlp_add_bf16.argtypes = [_p, _p, _p, _p, _p, _p, _p, _p, C.c_int64, C.c_float, _p]
    lib.ppn_nat128_proj_add_bf16.argtypes = [_p, _p, _p, C.c_int64, _p]
    lib.ppn_gemm_bf16.argtypes = [_p, _p, _p, _p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _p]
    lib.ppn_nat_gemm_bf16.argtypes = [_p, _p, _p, _p, _p, C.c_int32, _p, _p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_float, _p]
    lib.ppn_row_stats_bf16.argtypes = [_p, C.c_int64, C.c_int32, _p, _p]
    lib.ppn_nat_gemm_partials.argtypes = [C.c_int32]
    lib.ppn_nat_mlp_supported.argtypes = [C.c_int64, C.c_int32, C.c_int32]
    lib.ppn_nat_mlp_pack_bf16.argtypes = [_p, _p, _p, C.c_int32, C.c_int32, _p]
    lib.ppn_nat_mlp_bf16.argtypes = [_p, _p, _p, _p, _p, C.c_int64, C.c_int32, C.c_int32, C.c_float, _p]
    lib.ppn_label_masks.argtypes = [C.POINTER(PathsStruct), C.POINTER(MapsStruct), C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                    _p, _p, _p]
    for name in EXPORTS:
        getattr(lib, name)
        if name not in ("ppn_error_string", "ppn_na2d_bwd_workspace"):
            getattr(lib, name).restype = C.c_int
    return lib


lib = _load()


class PpnError(RuntimeError):
    pass


def check(rc, what):
    if rc != PPN_OK:
        msg = lib.ppn_error_string(rc).decode()
        extra = f" (hipError_t {lib.ppn_last_hip_error()})" if rc == -2 else ""
        raise PpnError(f"{what}: {msg}{extra}")
