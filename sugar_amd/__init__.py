"""sugar_amd -- the MI355X-native splatting hot path of SuGaR.  Process-wide switches of the HIP library:"""


def set_exact_alpha(on: bool = True) -> None:
    """Exact-alpha mode of the blend kernels (include/sugar_raster.h: SGR_FLAG_EXACT_ALPHA): alpha, T, final_T and n_contrib
    bit-identical to the reference's kernels, gradients within the reference's own float-atomic noise; ~12 more vector
    instructions per (list entry, 8x8 block).  Off by default (the default meets north_star's 1e-4 bar); also SGR_EXACT_ALPHA=1."""
    from . import _lib
    _lib.load().sgr_set_exact_alpha(1 if on else 0)


def exact_alpha() -> bool:
    from . import _lib
    return bool(_lib.load().sgr_get_exact_alpha())
