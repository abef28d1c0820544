"""Drop-in module name of the reference's rasterizer package
(gaussian_splatting/submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py):
`from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer` resolves to the
MI355X-native implementation when the repository root is on sys.path."""
from sugar_amd.diff_gaussian_rasterization import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    _RasterizeGaussians,
    rasterize_gaussians,
    cpu_deep_copy_tuple,
    _C,
)
