"""TEST-ONLY stand-in for the rasterizer module, backed by the CPU oracle.  Lets the CPU test-suite exercise the
host logic that sits above the C ABI (API contract, train step, view-sharded data parallelism over gloo) without a
GPU.  Never imported by the product."""
from __future__ import annotations

import numpy as np
import torch

from oracle import cpu_oracle as orc
from sugar_amd.diff_gaussian_rasterization import GaussianRasterizationSettings  # noqa: F401  (same settings tuple)


def _np(t):
    return None if t is None or t.numel() == 0 else t.detach().cpu().numpy()


class _OracleRasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs):
        st = orc.forward(_np(means3D), _np(opacities), shs=_np(sh), colors_precomp=_np(colors_precomp), scales=_np(scales),
                         rotations=_np(rotations), cov3D_precomp=_np(cov3Ds_precomp), viewmatrix=_np(rs.viewmatrix),
                         projmatrix=_np(rs.projmatrix), campos=_np(rs.campos), bg=_np(rs.bg), W=rs.image_width,
                         H=rs.image_height, tanfovx=rs.tanfovx, tanfovy=rs.tanfovy, sh_degree=rs.sh_degree,
                         scale_modifier=rs.scale_modifier)
        ctx.st = st
        radii = torch.from_numpy(st["radii"].copy())
        ctx.mark_non_differentiable(radii)
        return torch.from_numpy(st["color"].copy()), radii

    @staticmethod
    def backward(ctx, grad_color, _):
        g = orc.backward(ctx.st, grad_color.contiguous().numpy())
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        return (t(g["dL_dmeans3D"]), t(g["dL_dmeans2D"]), t(g["dL_dsh"]), t(g["dL_dcolors"]), t(g["dL_dopacity"]),
                t(g["dL_dscales"]), t(g["dL_drotations"]), t(g["dL_dcov3D"]), None)


class GaussianRasterizer(torch.nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        e = torch.Tensor([])
        return _OracleRasterize.apply(means3D, means2D, e if shs is None else shs,
                                      e if colors_precomp is None else colors_precomp, opacities,
                                      e if scales is None else scales, e if rotations is None else rotations,
                                      e if cov3D_precomp is None else cov3D_precomp, self.raster_settings)
