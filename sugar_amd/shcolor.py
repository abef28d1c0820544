"""Host side of the fused SH -> RGB op (C ABI: sgr_sh_to_rgb_forward / sgr_sh_to_rgb_backward).

`get_points_rgb` mirrors SuGaR.get_points_rgb (sugar_scene/sugar_model.py:839-883): same arguments, same result
(`clamp_min(eval_sh(sh_levels - 1, sh, dir) + 0.5, 0)`), differentiable w.r.t. the SH coefficients and the positions
(or directions).  SuGaR's trainers call it every step to feed `colors_precomp` (coarse_sdf.py:51, sugar_model.py:2187-2200);
the reference evaluates it with ~30 elementwise launches plus their autograd twins.  To use it with the unmodified
reference:  `SuGaR.get_points_rgb = sugar_amd.shcolor.get_points_rgb`.  GPU tensors only; the HIP library must be built.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _rows(sh):
    """(tensor to pass, M = row stride in coefficients) for sh[P, n, 3], accepting a leading-coefficient slice of a wider
    contiguous [P, M, 3] tensor without a copy."""
    P, n, _ = sh.shape
    st = sh.stride()
    if P > 0 and st[2] == 1 and st[1] == 3 and st[0] % 3 == 0 and st[0] // 3 >= n and st[0] // 3 <= 16:
        return sh, st[0] // 3
    return sh.contiguous(), n


class _ShToRgb(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sh, positions, centers, directions, degree):
        lib = _lib.load()
        if not sh.is_cuda:
            raise RuntimeError("the HIP SH->RGB op needs tensors on a ROCm device; there is no CPU fallback")
        sh_f = sh.float()
        sh_rows, M = _rows(sh_f)
        P = sh.shape[0]
        dev = sh.device
        pos = positions.contiguous().float() if positions is not None else None
        cen = centers.reshape(-1, 3).contiguous().float() if centers is not None else None
        dirs = directions.contiguous().float() if directions is not None else None
        colors = torch.empty(P, 3, device=dev)
        with torch.cuda.device(dev):
            rc = lib.sgr_sh_to_rgb_forward(P, int(degree), M, _p(sh_rows), _p(pos), _p(cen), 0 if cen is None else cen.shape[0],
                                           _p(dirs), _p(colors), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc < 0:
            raise RuntimeError(f"sgr_sh_to_rgb_forward failed ({rc})")
        ctx.save_for_backward(sh_rows, pos, cen, dirs)
        ctx.meta = (int(degree), M, sh.shape)
        return colors

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        sh_rows, pos, cen, dirs = ctx.saved_tensors
        degree, M, shape = ctx.meta
        P, n = shape[0], shape[1]
        dev = sh_rows.device
        need_sh, need_pos, _, need_dir, _ = ctx.needs_input_grad
        dsh = torch.empty(P, M, 3, device=dev) if need_sh else None
        dpos = torch.empty(P, 3, device=dev) if (need_pos and pos is not None) else None
        ddir = torch.empty(P, 3, device=dev) if (need_dir and dirs is not None) else None
        g = g.contiguous().float()  # (kept referenced until the call is enqueued)
        with torch.cuda.device(dev):
            rc = lib.sgr_sh_to_rgb_backward(P, degree, M, _p(sh_rows), _p(pos), _p(cen), 0 if cen is None else cen.shape[0],
                                            _p(dirs), _p(g), _p(dsh), _p(dpos), _p(ddir),
                                            C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc < 0:
            raise RuntimeError(f"sgr_sh_to_rgb_backward failed ({rc})")
        if dsh is not None and M != n:
            dsh = dsh[:, :n]  # gradient of the slice that was passed in; autograd pads it back into the wide tensor
        return dsh, dpos, None, ddir, None


def sh_to_rgb(sh_coordinates, sh_levels, positions=None, camera_centers=None, directions=None):
    """colors[P,3] = clamp_min(eval_sh(sh_levels-1, sh_coordinates[:, :sh_levels**2], dir) + 0.5, 0)"""
    if camera_centers is not None:
        if positions is None:
            raise ValueError("positions must be given together with camera_centers")
        cen = camera_centers.reshape(-1, 3)
        if cen.shape[0] not in (1, positions.shape[0]):
            raise ValueError("camera_centers must have shape (n_pts, 3) or (1, 3)")
        directions = None
    elif directions is not None:
        positions, cen = None, None
    else:
        raise ValueError("Either camera_centers or directions must be provided.")
    sh = sh_coordinates[:, :sh_levels ** 2]
    return _ShToRgb.apply(sh, positions, cen, directions, sh_levels - 1)


def get_points_rgb(self, positions=None, camera_centers=None, directions=None, sh_levels=None, sh_coordinates=None):
    """Drop-in for SuGaR.get_points_rgb (sugar_scene/sugar_model.py:839-883); `self` is the SuGaR model (needs `.points`
    and `.sh_coordinates`)."""
    if positions is None:
        positions = self.points
    if camera_centers is None and directions is None:
        raise ValueError("Either camera_centers or directions must be provided.")
    if sh_coordinates is None:
        sh_coordinates = self.sh_coordinates
    if sh_levels is None:
        # the reference indexes sh_levels**2 with None here and fails; use every coefficient that was passed
        sh_levels = int(round(sh_coordinates.shape[1] ** 0.5))
    return sh_to_rgb(sh_coordinates, sh_levels, positions=positions, camera_centers=camera_centers,
                     directions=None if camera_centers is not None else directions)
