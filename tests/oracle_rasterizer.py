"""TEST-ONLY stand-in for the rasterizer module, backed by the CPU oracle.  Lets the CPU test-suite exercise the
host logic that sits above the C ABI (API contract, train step, view-sharded data parallelism over gloo) without a
GPU.  Never imported by the product."""
from __future__ import annotations

import numpy as np
import torch

from oracle import cpu_oracle as orc
import contextlib

from sugar_amd.diff_gaussian_rasterization import GaussianRasterizationSettings  # noqa: F401  (same settings tuple)

_SINK: dict = {}


@contextlib.contextmanager
def grad_sink(**buffers):
    """Same protocol as sugar_amd.diff_gaussian_rasterization.grad_sink (only compact_sh / out are honoured here)."""
    global _SINK
    old, _SINK = _SINK, dict(buffers)
    try:
        yield
    finally:
        _SINK = old


def sh_grad_from_views_torch(means3D, campos_all, dcolor_all, sh_degree, out):
    """Torch restatement of sgr_sh_grad_from_views for the CPU rehearsal (per-view SH backward, backward.cu:47-97)."""
    from oracle.torch_cpu_rasterizer import SH_C0, SH_C1, SH_C2, SH_C3
    acc = torch.zeros_like(out)
    for v in range(dcolor_all.shape[0]):
        d = means3D - campos_all[v][None]
        d = d / d.norm(dim=1, keepdim=True)
        x, y, z = d[:, 0], d[:, 1], d[:, 2]
        b = [torch.full_like(x, SH_C0)]
        if sh_degree > 0:
            b += [-SH_C1 * y, SH_C1 * z, -SH_C1 * x]
        if sh_degree > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            b += [SH_C2[0] * xy, SH_C2[1] * yz, SH_C2[2] * (2 * zz - xx - yy), SH_C2[3] * xz, SH_C2[4] * (xx - yy)]
        if sh_degree > 2:
            b += [SH_C3[0] * y * (3 * xx - yy), SH_C3[1] * xy * z, SH_C3[2] * y * (4 * zz - xx - yy),
                  SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy), SH_C3[4] * x * (4 * zz - xx - yy), SH_C3[5] * z * (xx - yy),
                  SH_C3[6] * x * (xx - 3 * yy)]
        B = torch.stack(b, dim=1)  # [P, nb]
        acc[:, : B.shape[1]] += B[:, :, None] * dcolor_all[v][:, None, :]
    out.copy_(acc)
    return out


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
        ctx.sink = dict(_SINK)
        radii = torch.from_numpy(st["radii"].copy())
        ctx.mark_non_differentiable(radii)
        return torch.from_numpy(st["color"].copy()), radii

    @staticmethod
    def backward(ctx, grad_color, _):
        g = orc.backward(ctx.st, grad_color.contiguous().numpy())
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        if ctx.sink.get("compact_sh") and g["dL_dsh"].size:
            masked = g["dL_dcolors"] * (1 - ctx.st["clamped"].astype(np.float32))  # dL_dRGB of backward.cu:31-34
            ctx.sink["out"]["masked_colors"] = t(masked)
            return (t(g["dL_dmeans3D"]), t(g["dL_dmeans2D"]), None, t(g["dL_dcolors"]), t(g["dL_dopacity"]),
                    t(g["dL_dscales"]), t(g["dL_drotations"]), t(g["dL_dcov3D"]), None)
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
