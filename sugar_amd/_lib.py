"""ctypes loader for sugar_amd/libsugar_raster.so (the C ABI of include/sugar_raster.h).

There is NO fallback: if the HIP library is missing or a symbol of the ABI is absent, importing the
rasterizer fails loudly.  The CPU oracle under oracle/ is test infrastructure and is never used here.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SGR_LIB_PATH") or os.path.join(_HERE, "libsugar_raster.so")  # (SGR_LIB_PATH: A/B of a variant build)

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)

# symbol -> (restype, argtypes); must list every function declared in include/sugar_raster.h
_vp, _i, _f, _i64, _sz = C.c_void_p, C.c_int, C.c_float, C.c_int64, C.c_size_t
SIGNATURES = {
    "sgr_abi_version": (_i, []),
    "sgr_last_error": (C.c_char_p, []),
    "sgr_set_deep_min": (None, [_i]),
    "sgr_get_deep_min": (_i, []),
    "sgr_set_exact_alpha": (None, [_i]),
    "sgr_get_exact_alpha": (_i, []),
    "sgr_forward": (_i64, [ALLOC_FN, _vp, ALLOC_FN, _vp, ALLOC_FN, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp,
                           _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _vp, _i, _vp]),
    "sgr_forward_ex": (_i64, [ALLOC_FN, _vp, ALLOC_FN, _vp, ALLOC_FN, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp,
                              _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _vp, _i, _vp, _vp]),
    "sgr_backward": (_i, [_i, _i, _i, _i64, _vp, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp,
                          _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "sgr_backward_phase": (_i, [_i, _i, _i, _i, _i64, _vp, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp,
                                _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "sgr_backward_ex": (_i, [_i, _i, _i, _i, _i64, _vp, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp,
                             _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "sgr_mark_visible": (_i, [_i, _vp, _vp, _vp, _vp, _vp]),
    "sgr_trainer_create": (_vp, [_vp]),
    "sgr_trainer_destroy": (None, [_vp]),
    "sgr_trainer_set_binning": (_i, [_vp, _vp, _sz, _i64]),
    "sgr_trainer_step": (_i, [_vp, _vp, _i, _vp, _vp]),
    "sgr_trainer_forward_valid": (_i, [_vp, _vp]),
    "sgr_trainer_last_error": (C.c_char_p, []),
    "sgr_rccl_unique_id": (_i, [_vp]),
    "sgr_trainer_comm_init": (_i, [_vp, _vp, _i, _i, _vp, _sz]),
    "sgr_trainer_comm_destroy": (_i, [_vp]),
    "sgr_trainer_comm_abort": (_i, [_vp]),
    "sgr_trainer_step_exchange": (_i, [_vp, _vp, _i, _vp]),
    "sgr_trainer_set_exchange_chunks": (_i, [_vp, _i]),
    "sgr_trainer_last_exchange_wait_ms": (C.c_double, [_vp]),
    "sgr_bin2_bytes": (_sz, [_i, _i, _i]),
    "sgr_sh_grad_from_views": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _i64, _vp, _vp]),
    "sgr_sh_adam_from_views": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _f, _f, _f, _f, _f, _i, _f, _vp]),
    "sgr_sh_adam_from_views_ex": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _f, _f, _f, _f, _f, _i, _f, _vp, _vp]),
    "sgr_geom_bytes": (_sz, [_i]),
    "sgr_img_bytes": (_sz, [_i, _i]),
    "sgr_binning_bytes": (_sz, [_i64, _i, _i]),
    "sgr_geom_rec_offset": (_sz, [_i]),
    "sgr_img_final_T_offset": (_sz, [_i, _i]),
    "sgr_img_n_contrib_offset": (_sz, [_i, _i]),
    "sgr_img_tile_start_offset": (_sz, [_i, _i]),
    "sgr_img_tile_maxc_offset": (_sz, [_i, _i]),
    "sgr_img_tile_walked_offset": (_sz, [_i, _i]),
    "sgr_img_header_offset": (_sz, [_i, _i]),
    "sgr_binning_point_list_offset": (_sz, [_i64]),
    "sgr_profile_enable": (None, [_i]),
    "sgr_profile_read": (_i, [_vp, _vp, _i]),
    "sgr_l1_ssim_scratch_bytes": (_sz, [_i, _i, _i]),
    "sgr_l1_ssim_forward": (_i, [_i, _i, _i, _vp, _vp, _f, _vp, _vp, _vp]),
    "sgr_l1_ssim_backward": (_i, [_i, _i, _i, _vp, _vp, _f, _vp, _vp, _vp, _vp]),
    "sgr_l1_ssim_backward_ex": (_i, [_i, _i, _i, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp]),
    "sgr_adam_step": (_i, [C.c_longlong, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _i, _f, _vp]),
    "sgr_adam_step_multi": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sgr_adam_step_ex": (_i, [C.c_longlong, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _i, _f, _vp, C.c_longlong, _vp]),
    "sgr_density_field_forward": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp]),
    "sgr_density_field_backward": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sgr_density_field_backward_scratch_bytes": (_sz, [_i, _i, _i]),
    "sgr_scatter_add_rows_scratch_bytes": (_sz, [C.c_longlong, _i]),
    "sgr_scatter_add_rows": (_i, [C.c_longlong, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "sgr_density_field_backward_gather": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sgr_scaled_rotation_forward": (_i, [_i, _vp, _vp, _i, _vp, _vp]),
    "sgr_scaled_rotation_backward": (_i, [_i, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "sgr_pack_gaussians": (_i, [_i, _vp, _vp, _vp, _vp, _vp]),
    "sgr_view_std": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sgr_view_depth_rgb": (_i, [_i, _vp, _vp, _vp, _vp]),
    "sgr_unproject_pixels": (_i, [_i, _vp, _vp, _i, _i, _f, _f, _vp, _vp, _vp]),
    "sgr_pick_pixels_scratch_bytes": (_sz, [_i]),
    "sgr_pick_pixels": (_i, [_i, _vp, _i, C.c_uint32, _vp, _vp, _vp, _vp, _vp]),
    "sgr_compact_level_rows": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sgr_compact_level_rows_scratch_bytes": (_sz, [_i, _i]),
    "sgr_level_set_points": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "sgr_sh_to_rgb_forward": (_i, [_i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "sgr_sh_to_rgb_backward": (_i, [_i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sgr_activations_forward": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sgr_activations_backward": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sgr_dist2": (_i, [_i, _vp, _vp, _vp]),
    "sgr_knn": (_i, [_i, _vp, _i, _vp, _i, _vp, _vp, _vp]),
    "sgr_knn_grid_scratch_bytes": (_sz, [_i]),
    "sgr_knn_grid": (_i, [_i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "sgr_dist2_grid": (_i, [_i, _vp, _vp, _vp, _vp]),
    "sgr_rasterize_meshes_scratch_bytes": (_sz, [_i64, _i, _i]),
    "sgr_splat_mesh_face_verts": (_i, [_i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp]),
    "sgr_rasterize_meshes": (_i64, [_vp, _i64, _i64, _i, _i, _f, _i, _i, _i, _i, _vp, _sz, ALLOC_FN, _vp, _vp, _vp, _vp, _vp, _vp]),
}

ABI_VERSION = 4  # SGR_ABI_VERSION of include/sugar_raster.h these bindings were written for


# ---- structs of include/sugar_raster.h
class ForwardInfo(C.Structure):
    _fields_ = [("binning_mode", _i), ("sync_free", _i), ("speculation", _i)]


class ForwardOpts(C.Structure):
    _fields_ = [("binning_capacity", _i64), ("flags", _i), ("header_host", _vp), ("header_event", _vp), ("tile_need", _vp),
                ("tile_need_out", _vp), ("hint_margin", _f), ("chunk_grid", C.c_uint32), ("info", C.POINTER(ForwardInfo)),
                ("tile_order", _vp), ("tile_order_out", _vp)]


class BackwardOpts(C.Structure):
    _fields_ = [("max_radii2D", _vp), ("grad_accum", _vp), ("denom", _vp), ("campos_row", _vp), ("flags", _i)]


class TrainConfig(C.Structure):
    _fields_ = [("P", _i), ("D", _i), ("M", _i), ("width", _i), ("height", _i),
                ("flat", _vp), ("flat_grad", _vp), ("exp_avg", _vp), ("exp_avg_sq", _vp),
                ("off_xyz", C.c_longlong), ("off_opacity", C.c_longlong), ("off_scaling", C.c_longlong),
                ("off_rotation", C.c_longlong), ("off_features", C.c_longlong), ("n_small", C.c_longlong),
                ("lr_xyz", _f), ("lr_opacity", _f), ("lr_scaling", _f), ("lr_rotation", _f), ("lr_features_dc", _f),
                ("lr_features_rest", _f), ("beta1", _f), ("beta2", _f), ("eps", _f), ("lambda_dssim", _f),
                ("background", _vp), ("geom", _vp), ("geom_bytes", _sz), ("img", _vp), ("img_bytes", _sz), ("binning", _vp),
                ("binning_bytes", _sz), ("binning_capacity", _i64), ("loss_scratch", _vp), ("image", _vp), ("grad_image", _vp),
                ("loss_out", _vp), ("colors", _vp), ("radii", _vp), ("header_host", _vp), ("dL_dmean2D", _vp),
                ("max_radii2D", _vp), ("grad_accum", _vp), ("denom", _vp)]


class TrainView(C.Structure):
    _fields_ = [("viewmatrix", _vp), ("projmatrix", _vp), ("campos", _vp), ("tan_fovx", _f), ("tan_fovy", _f), ("gt_image", _vp),
                ("tile_need", _vp), ("tile_need_out", _vp), ("hint_margin", _f), ("chunk_grid", C.c_uint32),
                ("tile_order", _vp), ("tile_order_out", _vp), ("flags", C.c_uint32)]


class TrainExchange(C.Structure):
    _fields_ = [("n_views", _i), ("all_colors", _vp), ("view_stride", _i64), ("all_campos", _vp), ("grad_scale", _f), ("step", _i),
                ("g_begin", _i64), ("g_end", _i64), ("f_begin", _i64), ("f_end", _i64)]


SGR_FLAG_RAW_PARAMS, SGR_FLAG_SINGLE_LEVEL_BINNING, SGR_FLAG_SPECULATIVE = 1, 2, 8
SGR_BWD_TILE_ORDER_READY = 1
HDR_R, HDR_HINT_MISS, HDR_L1_OVERFLOW = 0, 3, 6

_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the gfx950 HIP rasterizer is not built. Run `python -m sugar_amd.build` "
            "(or __graft_entry__.build()). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError(f"{LIB_PATH} does not export `{name}` (ABI mismatch; rebuild)") from e
        fn.restype = res
        fn.argtypes = args
    if lib.sgr_abi_version() != ABI_VERSION:
        raise ImportError(f"sugar_raster ABI version mismatch: library {lib.sgr_abi_version()}, bindings {ABI_VERSION} (rebuild)")
    _lib = lib
    return lib


def last_error() -> str:
    return load().sgr_last_error().decode(errors="replace")
