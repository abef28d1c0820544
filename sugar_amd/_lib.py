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
    "sgr_forward": (_i64, [ALLOC_FN, _vp, ALLOC_FN, _vp, ALLOC_FN, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp,
                           _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _vp, _i, _vp]),
    "sgr_forward_ex": (_i64, [ALLOC_FN, _vp, ALLOC_FN, _vp, ALLOC_FN, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp,
                              _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _vp, _i, _vp, _i64, _i]),
    "sgr_backward": (_i, [_i, _i, _i, _i64, _vp, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp,
                          _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "sgr_backward_phase": (_i, [_i, _i, _i, _i, _i64, _vp, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp,
                                _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "sgr_mark_visible": (_i, [_i, _vp, _vp, _vp, _vp, _vp]),
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
    "sgr_adam_step": (_i, [C.c_longlong, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _i, _f, _vp]),
    "sgr_adam_step_ex": (_i, [C.c_longlong, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _i, _f, _vp, C.c_longlong, _vp]),
    "sgr_density_field_forward": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp]),
    "sgr_density_field_backward": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sgr_density_field_backward_scratch_bytes": (_sz, [_i, _i, _i]),
    "sgr_density_field_backward_gather": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sgr_scaled_rotation_forward": (_i, [_i, _vp, _vp, _i, _vp, _vp]),
    "sgr_scaled_rotation_backward": (_i, [_i, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "sgr_pack_gaussians": (_i, [_i, _vp, _vp, _vp, _vp, _vp]),
    "sgr_level_set_points": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "sgr_sh_to_rgb_forward": (_i, [_i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "sgr_sh_to_rgb_backward": (_i, [_i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sgr_activations_forward": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sgr_activations_backward": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sgr_set_binning_mode": (_i, [_i]),
    "sgr_last_binning_mode": (_i, []),
    "sgr_set_blend_variant": (_i, [_i]),
    "sgr_dist2": (_i, [_i, _vp, _vp, _vp]),
    "sgr_knn": (_i, [_i, _vp, _i, _vp, _i, _vp, _vp, _vp]),
    "sgr_knn_grid_scratch_bytes": (_sz, [_i]),
    "sgr_knn_grid": (_i, [_i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "sgr_dist2_grid": (_i, [_i, _vp, _vp, _vp, _vp]),
}

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
    if lib.sgr_abi_version() != 1:
        raise ImportError("sugar_raster ABI version mismatch")
    if os.environ.get("SGR_BLEND_VARIANT"):  # development switch, see sgr_set_blend_variant
        lib.sgr_set_blend_variant(int(os.environ["SGR_BLEND_VARIANT"]))
    _lib = lib
    return lib


def last_error() -> str:
    return load().sgr_last_error().decode(errors="replace")
