"""Helpers shared by the parity tests: run the CPU oracle and the HIP path on the same seeded inputs."""
from __future__ import annotations

import numpy as np
import torch

from oracle import cpu_oracle as orc


def scene_kwargs(scene, cam, bg, *, use_sh=True, use_cov=False, sh_degree=3, scale_modifier=1.0):
    """numpy kwargs for oracle.cpu_oracle.forward"""
    kw = dict(viewmatrix=cam.viewmatrix.numpy(), projmatrix=cam.projmatrix.numpy(), campos=cam.campos.numpy(),
              bg=bg.numpy(), W=cam.image_width, H=cam.image_height, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
              sh_degree=sh_degree, scale_modifier=scale_modifier)
    if use_sh:
        kw["shs"] = scene.shs.numpy()
    else:
        kw["colors_precomp"] = precomputed_colors(scene).numpy()
    if use_cov:
        kw["cov3D_precomp"] = precomputed_cov(scene, scale_modifier).numpy()
    else:
        kw["scales"] = scene.scales.numpy()
        kw["rotations"] = scene.rotations.numpy()
    return kw


def precomputed_colors(scene):
    g = torch.Generator().manual_seed(99)
    return torch.rand(scene.means3D.shape[0], 3, generator=g)


def precomputed_cov(scene, mod=1.0):
    """Sigma = R S^2 R^T packed as (00,01,02,11,12,22) -- sugar_scene/sugar_model.py:2222-2239"""
    from oracle.torch_cpu_rasterizer import cov3d_from_scale_rot
    S = cov3d_from_scale_rot(scene.scales.double(), mod, scene.rotations.double())
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=1).float().contiguous()


def run_oracle(scene, cam, bg, **opts):
    kw = scene_kwargs(scene, cam, bg, **opts)
    return orc.forward(scene.means3D.numpy(), scene.opacities.numpy(), **kw)


def run_hip(scene, cam, bg, *, use_sh=True, use_cov=False, sh_degree=3, scale_modifier=1.0, grad_out=None,
            device="cuda:0", debug=False):
    """Runs the product path through the reference-shaped Python API. Returns dict of numpy arrays."""
    from sugar_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from sugar_amd import _lib
    dev = torch.device(device)
    settings = GaussianRasterizationSettings(
        image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
        bg=bg.to(dev), scale_modifier=scale_modifier, viewmatrix=cam.viewmatrix.to(dev),
        projmatrix=cam.projmatrix.to(dev), sh_degree=sh_degree, campos=cam.campos.to(dev), prefiltered=False,
        debug=debug)
    rast = GaussianRasterizer(settings)
    P = scene.means3D.shape[0]
    means3D = scene.means3D.to(dev).requires_grad_(True)
    means2D = torch.zeros(P, 3, device=dev, requires_grad=True)
    opac = scene.opacities.to(dev).requires_grad_(True)
    kw = {}
    leaves = dict(means3D=means3D, means2D=means2D, opacities=opac)
    if use_sh:
        leaves["shs"] = kw["shs"] = scene.shs.to(dev).requires_grad_(True)
    else:
        leaves["colors_precomp"] = kw["colors_precomp"] = precomputed_colors(scene).to(dev).requires_grad_(True)
    if use_cov:
        leaves["cov3D_precomp"] = kw["cov3D_precomp"] = precomputed_cov(scene, scale_modifier).to(dev).requires_grad_(True)
    else:
        leaves["scales"] = kw["scales"] = scene.scales.to(dev).requires_grad_(True)
        leaves["rotations"] = kw["rotations"] = scene.rotations.to(dev).requires_grad_(True)
    color, radii = rast(means3D=means3D, means2D=means2D, opacities=opac, **kw)
    out = dict(color=color.detach().cpu().numpy(), radii=radii.cpu().numpy())
    # private scratch, through the introspection ABI
    fn = color.grad_fn
    saved = fn.saved_tensors
    geom, binning, img = saved[7], saved[8], saved[9]
    lib = _lib.load()
    W, H = cam.image_width, cam.image_height
    R = fn.num_rendered
    out["num_rendered"] = R
    rec = geom.cpu().numpy()[: P * 48].view(np.float32).reshape(P, 12)
    out["rec"] = rec  # columns: REC_* below (GeomRec, sugar_amd/csrc/sgr_common.h)
    imgb = img.cpu().numpy()
    T = ((W + 15) // 16) * ((H + 15) // 16)
    o = lib.sgr_img_final_T_offset(W, H); out["final_T"] = imgb[o:o + W * H * 4].view(np.float32)
    o = lib.sgr_img_n_contrib_offset(W, H); out["n_contrib"] = imgb[o:o + W * H * 4].view(np.uint32)
    o = lib.sgr_img_tile_start_offset(W, H); out["tile_start"] = imgb[o:o + (T + 1) * 4].view(np.uint32)
    o = lib.sgr_img_tile_maxc_offset(W, H); out["tile_maxc"] = imgb[o:o + T * 4].view(np.uint32)
    o = lib.sgr_binning_point_list_offset(R); out["point_list"] = binning.cpu().numpy()[o:o + R * 4].view(np.uint32)
    if grad_out is not None:
        color.backward(torch.as_tensor(grad_out).to(dev))
        out["grads"] = {k: (v.grad.detach().cpu().numpy() if v.grad is not None else None) for k, v in leaves.items()}
    return out


# float columns of the private 48-byte geometry record (GeomRec, sugar_amd/csrc/sgr_common.h)
REC_XY, REC_CONIC, REC_OPACITY, REC_DEPTH, REC_RADIUS, REC_RGB, REC_CLAMPED = slice(0, 2), [2, 3, 4], 5, 6, 7, slice(8, 11), 11


def rel_stats(a, b, floor_frac=1e-3):
    """Per-element relative error with a floor of floor_frac * max|b| (so exact zeros do not blow up)."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    scale = np.abs(b).max() if b.size else 0.0
    floor = max(scale * floor_frac, 1e-30)
    rel = np.abs(a - b) / (np.abs(b) + floor)
    nrm = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
    return dict(max_rel=float(rel.max()) if rel.size else 0.0, frac_gt_1e4=float((rel > 1e-4).mean()) if rel.size else 0.0,
                norm_rel=float(nrm), max_abs=float(np.abs(a - b).max()) if a.size else 0.0, scale=float(scale))
