"""Host side of SuGaR's density field and level-set sampler (C ABI: sgr_density_field_*, sgr_level_set_points).

`density_field(x, nbr_idx, centers, inv_scaled_rot, strengths)` is the differentiable core of
SuGaR.get_field_values (sugar_scene/sugar_model.py:1247-1281); `level_set_points(...)` is the per-pixel part of
SuGaR.compute_level_surface_points_from_camera_fast (:1971-2079).  Both read the same tensors SuGaR already holds
(`self.points`, `self.get_covariance(return_full_matrix=True, return_sqrt=True, inverse_scales=True)`, `self.strengths`,
`self.knn_idx`).  GPU tensors only; the HIP library must be built.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _pack(lib, ce, Bm, st):
    """one 64-byte record per Gaussian for the kernels' neighbour gathers (sgr_pack_gaussians)"""
    P, dev = ce.shape[0], ce.device
    packed = torch.empty(P, 16, device=dev)
    with torch.cuda.device(dev):
        rc = lib.sgr_pack_gaussians(P, _p(ce), _p(Bm), _p(st), _p(packed), _stream(dev))
    if rc < 0:
        raise RuntimeError(f"sgr_pack_gaussians failed ({rc})")
    return packed


def _prep(x, nbr_idx, centers, B, strengths):
    if not x.is_cuda:
        raise RuntimeError("the HIP density field needs tensors on a ROCm device; there is no CPU fallback")
    P = centers.shape[0]
    return (x.contiguous().float(), nbr_idx.contiguous().to(torch.int64), centers.contiguous().float(),
            B.reshape(P, 9).contiguous().float(), strengths.reshape(P).contiguous().float())


class _DensityField(torch.autograd.Function):
    use_gather = True  # False: the 13-float-atomics-per-pair kernel (kept for comparison)

    @staticmethod
    def forward(ctx, x, nbr_idx, centers, inv_scaled_rot, strengths, density_factor):
        lib = _lib.load()
        xs, nb, ce, Bm, st = _prep(x, nbr_idx, centers, inv_scaled_rot, strengths)
        N, K = nb.shape
        dev = xs.device
        opac = torch.empty(N, K, device=dev)
        dens = torch.empty(N, device=dev)
        packed = _pack(lib, ce, Bm, st)
        with torch.cuda.device(dev):
            rc = lib.sgr_density_field_forward(N, K, _p(xs), _p(nb), _p(ce), _p(Bm), _p(st), float(density_factor), _p(opac),
                                               _p(dens), _p(packed), _stream(dev))
        if rc < 0:
            raise RuntimeError(f"sgr_density_field_forward failed ({rc})")
        ctx.save_for_backward(xs, nb, ce, Bm, st, packed)
        ctx.factor = float(density_factor)
        ctx.shapes = (inv_scaled_rot.shape, strengths.shape)
        return opac, dens

    @staticmethod
    def backward(ctx, g_opac, g_dens):
        lib = _lib.load()
        xs, nb, ce, Bm, st, packed = ctx.saved_tensors
        N, K = nb.shape
        dev = xs.device
        P = ce.shape[0]
        dx = torch.empty(N, 3, device=dev)
        go = g_opac.contiguous().float() if g_opac is not None else None
        gd = g_dens.contiguous().float() if g_dens is not None else None
        if _DensityField.use_gather:
            # one integer atomic per pair + a per-Gaussian gather (every output row is written)
            dce = torch.empty(P, 3, device=dev); dB = torch.empty(P, 9, device=dev); dst = torch.empty(P, device=dev)
            scratch = torch.empty(lib.sgr_density_field_backward_scratch_bytes(N, K, P), dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                rc = lib.sgr_density_field_backward_gather(N, K, P, _p(xs), _p(nb), _p(ce), _p(Bm), _p(st), ctx.factor, _p(go), _p(gd),
                                                           _p(dx), _p(dce), _p(dB), _p(dst), _p(scratch), _p(packed), _stream(dev))
        else:
            dce = torch.zeros(P, 3, device=dev); dB = torch.zeros(P, 9, device=dev); dst = torch.zeros(P, device=dev)
            with torch.cuda.device(dev):
                rc = lib.sgr_density_field_backward(N, K, _p(xs), _p(nb), _p(ce), _p(Bm), _p(st), ctx.factor, _p(go), _p(gd), _p(dx),
                                                    _p(dce), _p(dB), _p(dst), _p(packed), _stream(dev))
        if rc < 0:
            raise RuntimeError(f"sgr_density_field_backward failed ({rc})")
        Bshape, sshape = ctx.shapes
        return dx, None, dce, dB.reshape(Bshape), dst.reshape(sshape), None


class _ScaledRotation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, quaternions, scaling, inverse_scales):
        lib = _lib.load()
        if not quaternions.is_cuda:
            raise RuntimeError("the HIP scaled-rotation op needs tensors on a ROCm device; there is no CPU fallback")
        q, s = quaternions.contiguous().float(), scaling.contiguous().float()
        P, dev = q.shape[0], q.device
        out = torch.empty(P, 3, 3, device=dev)
        with torch.cuda.device(dev):
            rc = lib.sgr_scaled_rotation_forward(P, _p(q), _p(s), int(bool(inverse_scales)), _p(out), _stream(dev))
        if rc < 0:
            raise RuntimeError(f"sgr_scaled_rotation_forward failed ({rc})")
        ctx.save_for_backward(q, s)
        ctx.inverse = int(bool(inverse_scales))
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        q, s = ctx.saved_tensors
        P, dev = q.shape[0], q.device
        dq, ds = torch.empty_like(q), torch.empty_like(s)
        g = g.contiguous().float()  # (kept referenced until the call is enqueued: a temporary's block goes back to the allocator at once)
        with torch.cuda.device(dev):
            rc = lib.sgr_scaled_rotation_backward(P, _p(q), _p(s), ctx.inverse, _p(g), _p(dq), _p(ds), _stream(dev))
        if rc < 0:
            raise RuntimeError(f"sgr_scaled_rotation_backward failed ({rc})")
        return dq, ds, None


def scaled_rotation(quaternions, scaling, inverse_scales: bool = False):
    """SuGaR.get_covariance(return_sqrt=True, inverse_scales=...) (sugar_scene/sugar_model.py:730-736):
    quaternion_to_matrix(quaternions) * s[:, None] with s = scaling or 1 / scaling.clamp(min=1e-8); differentiable."""
    return _ScaledRotation.apply(quaternions, scaling, inverse_scales)


def density_field(x, nbr_idx, centers, inv_scaled_rot, strengths, density_factor: float = 1.0):
    """Returns (neighbor_opacities[N,K], densities[N]) exactly as sugar_model.py:1270-1276, differentiable w.r.t. x,
    centers, inv_scaled_rot and strengths."""
    return _DensityField.apply(x, nbr_idx, centers, inv_scaled_rot, strengths, density_factor)


def level_set_points(world_points, nbr_idx, cam_center, centers, inv_scaled_rot, strengths, gaussian_std,
                     surface_levels=(0.1, 0.3, 0.5), n_points_in_range: int = 21, range_size: float = 3.0,
                     density_factor: float = 1.0, return_normals: bool = True, raw: bool = False):
    """Per level: dict(valid=bool[N], valid_idx=int64[n_valid], intersection_points=[n_valid,3], normals=[n_valid,3]) -- the `outputs` of
    sugar_model.py:2013-2081 (rows where the reference's empty_pixels is False).
    `raw=True`: no gathering (and no host round trip): returns (valid uint8[L,N], points[L,N,3], normals[L,N,3] or None) as the kernel
    wrote them -- sugar_amd.sampler compacts them on the device."""
    lib = _lib.load()
    wp, nb, ce, Bm, st = _prep(world_points, nbr_idx, centers, inv_scaled_rot, strengths)
    dev = wp.device
    N, K = nb.shape
    L = len(surface_levels)
    cam = cam_center.reshape(3).contiguous().float().to(dev)
    gstd = gaussian_std.reshape(-1).contiguous().float()
    valid = torch.empty(L, N, dtype=torch.uint8, device=dev)
    pts = torch.empty(L, N, 3, device=dev)
    nrm = torch.empty(L, N, 3, device=dev) if return_normals else None
    lv = (C.c_float * L)(*[float(v) for v in surface_levels])
    packed = _pack(lib, ce, Bm, st)
    with torch.cuda.device(dev):
        rc = lib.sgr_level_set_points(N, K, _p(wp), _p(nb), _p(cam), _p(ce), _p(Bm), _p(st), _p(gstd), L, lv,
                                      int(n_points_in_range), float(range_size), float(density_factor), _p(valid), _p(pts),
                                      _p(nrm), _p(packed), _stream(dev))
    if rc < 0:
        raise RuntimeError(f"sgr_level_set_points failed ({rc})")
    if raw:
        return valid, pts, nrm
    out = {}
    for i, level in enumerate(surface_levels):
        m = valid[i].bool()
        rows = m.nonzero(as_tuple=True)[0]      # one host round trip per level; every per-level output is gathered with it
        out[level] = dict(valid=m, valid_idx=rows, intersection_points=pts[i].index_select(0, rows),
                          normals=(nrm[i].index_select(0, rows) if return_normals else None))
    return out
