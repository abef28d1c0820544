"""sugar_amd -- the MI355X-native splatting hot path of SuGaR.  Process-wide switches of the HIP library:"""


def set_exact_alpha(on: bool = True) -> None:
    """Exact-alpha mode of the blend kernels (include/sugar_raster.h): alpha, T, final_T and n_contrib bit-identical to the
    reference's kernels, gradients within the reference's own float-atomic noise.  ON by default.  `set_exact_alpha(False)` (or
    SGR_EXACT_ALPHA=0 in the environment) selects the fast evaluation: ~6 % of a train step faster, gradients 3e-5 .. 1.4e-4
    norm-wise from the reference's -- around north_star's 1e-4 bar, not safely inside it."""
    from . import _lib
    _lib.load().sgr_set_exact_alpha(1 if on else 0)


def exact_alpha() -> bool:
    from . import _lib
    return bool(_lib.load().sgr_get_exact_alpha())


def set_deep_min(entries: int = 1024) -> None:
    """Hinted list length above which a tile's 8x8 blocks are blended by the eight-wave kernel for long lists
    (include/sugar_raster.h: sgr_set_deep_min); 0 disables.  Results are bit-identical either way."""
    from . import _lib
    _lib.load().sgr_set_deep_min(int(entries))
