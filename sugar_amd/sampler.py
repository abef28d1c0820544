"""The coarse-mesh extractor's per-view sampling pass on raw Gaussian buffers (no SuGaR object, no pytorch3d camera classes).

Restates `SuGaR.compute_level_surface_points_from_camera_fast(use_gaussian_depth=True)` (sugar_scene/sugar_model.py:1848-2083;
called once per training view by sugar_extractors/coarse_mesh.py:243-327) with every device-side piece on this package's HIP
kernels sharing the Gaussian buffers of the rasterizer:

    depth map            a render of the Gaussian rasterizer with the view-space depth as colour, background -1    :1901-1911
    pixels               those with a depth, a random subset of `n_surface_points` of them                          :1929-1957
    back-projection      the reference's NDC pixel tables (:1934-1941) through a pinhole camera                     :1958-1959
    neighbours           exact K nearest Gaussians of every back-projected pixel (HIP grid k-NN)                    :1962-1964
    level crossings      21 ray samples x 16 neighbours x levels + normals in ONE kernel (csrc/field.hip)           :1971-2079

`sugar_amd.sugar_patch` binds the same kernels to the reference's class through its own camera objects (pinned by the fixtures
the reference's method wrote); this module is what bench.py's config-4 line and callers without the reference's Python use.
The camera is a `sugar_amd.synthetic.Camera`-shaped tuple with DEVICE tensors (row-vector matrices, as the rasterizer takes
them).  GPU tensors only; there is no CPU path."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .field import level_set_points, scaled_rotation
from .knn import knn_points


def view_depth(means3D: torch.Tensor, viewmatrix: torch.Tensor) -> torch.Tensor:
    """z of every centre in the camera's frame, [P,1] (`get_world_to_view_transform().transform_points(points)[..., 2:]`)"""
    return means3D @ viewmatrix[:3, 2:3] + viewmatrix[3, 2]


def render_depth(means3D, scales, rotations, opacities, cam, bg_value: float = -1.0):
    """[H,W] depth image: the Gaussian rasterizer with depth as the colour of every Gaussian (sugar_model.py:1901-1911)"""
    from .diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = means3D.device
    ce = means3D.to(torch.float32).contiguous()
    vm = cam.viewmatrix.to(device=dev, dtype=torch.float32).contiguous()
    depth_rgb = torch.empty(ce.shape[0], 3, dtype=torch.float32, device=dev)   # (one launch: csrc/field.hip, k_view_depth_rgb)
    with torch.cuda.device(dev):
        rc = _lib.load().sgr_view_depth_rgb(int(ce.shape[0]), _p(ce), _p(vm), _p(depth_rgb), _stream(dev))
    if rc < 0:
        raise RuntimeError(f"sgr_view_depth_rgb failed ({rc})")
    st = GaussianRasterizationSettings(int(cam.image_height), int(cam.image_width), cam.tanfovx, cam.tanfovy,
                                       torch.full((3,), float(bg_value), device=dev), 1.0, cam.viewmatrix, cam.projmatrix, 0,
                                       cam.campos, False, False)
    img, _ = GaussianRasterizer(st)(means3D, torch.zeros_like(means3D), opacities, colors_precomp=depth_rgb, scales=scales,
                                    rotations=rotations)
    return img[0]


def _p(t):
    return C.c_void_p(t.data_ptr())


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def unproject_pixels(picked: torch.Tensor, depth_flat: torch.Tensor, cam) -> torch.Tensor:
    """World points of the picked pixels, one launch (csrc/field.hip: k_unproject_pixels).  The reference lays an NDC grid over the
    image (x = W/m - 2 col / (m - 1), +x to the LEFT, +y UP, m = min(W, H); sugar_model.py:1934-1941) and un-projects through a
    pytorch3d camera whose focal length in NDC units is 2 fx / m; the camera frame of the rasterizer (COLMAP: +x right, +y down) is
    that frame with x and y negated.  The view matrix is inverted inside the kernel (no `torch.linalg.inv`: that one checks its
    factorisation on the host)."""
    dev = depth_flat.device
    picked = picked.to(torch.int64).contiguous()
    depth_flat = depth_flat.to(torch.float32).contiguous()
    vm = cam.viewmatrix.to(device=dev, dtype=torch.float32).contiguous()
    out = torch.empty(picked.shape[0], 3, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.load().sgr_unproject_pixels(int(picked.shape[0]), _p(picked), _p(depth_flat), int(cam.image_width), int(cam.image_height),
                                              float(cam.tanfovx), float(cam.tanfovy), _p(vm), _p(out), _stream(dev))
    if rc < 0:
        raise RuntimeError(f"sgr_unproject_pixels failed ({rc})")
    return out


def view_std(means3D: torch.Tensor, rotations: torch.Tensor, scales: torch.Tensor, campos: torch.Tensor) -> torch.Tensor:
    """[P]: the extent of every Gaussian along its direction to the camera, | scales (.) R(q)^T normalize(campos - mean) |
    (sugar_model.py:1971-1972; unit quaternions, real part first) -- one launch (csrc/field.hip: k_view_std)"""
    dev = means3D.device
    ce, q, sc = (t.to(torch.float32).contiguous() for t in (means3D, rotations, scales))
    if q.data_ptr() % 16:
        q = q.clone()
    cc = campos.to(device=dev, dtype=torch.float32).reshape(3).contiguous()
    out = torch.empty(ce.shape[0], dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.load().sgr_view_std(int(ce.shape[0]), _p(ce), _p(q), _p(sc), _p(cc), _p(out), _stream(dev))
    if rc < 0:
        raise RuntimeError(f"sgr_view_std failed ({rc})")
    return out


@torch.no_grad()
def sample_level_sets(means3D, scales, rotations, opacities, cam, *, n_surface_points: int = 124_000,
                      surface_levels=(0.1, 0.3, 0.5), n_points_in_range: int = 21, range_size: float = 3.0,
                      density_factor: float = 1.0, K: int = 16, return_normals: bool = True, cpu_randperm: bool = False,
                      depth: torch.Tensor | None = None, sync_free: bool = False, seed: int | None = None):
    """One sampling pass for one view.  Returns {level: dict(intersection_points[n,3], pixel_idx[n], gaussian_idx[n], normals[n,3])}.
    `scales` / `rotations` / `opacities` are the ACTIVATED values ([P,3], unit [P,4], [P,1]); `cpu_randperm` draws the pixel subset on
    the CPU as the reference does (:1955), so that a seeded run picks the reference's pixels.

    `sync_free=True` (round 6): nothing in the pass waits for the GPU.  The pixel subset is chosen on the device into
    `n_surface_points` rows (sgr_pick_pixels: a uniformly random subset of the valid pixels, in raster order), every stage runs on
    that fixed size, and the valid rows of every level are moved to the front on the device (sgr_compact_level_rows).  The tensors of
    a level then have `n_surface_points` rows of which the first `count` -- a 0-d int32 DEVICE tensor in the level's dict -- are
    meaningful; `trim(result)` slices them (one host wait for all levels)."""
    if not means3D.is_cuda:
        raise RuntimeError("sample_level_sets needs tensors on a ROCm device; there is no CPU fallback")
    from .sugar_patch import random_prefix_of_permutation
    dev = means3D.device
    W = int(cam.image_width)
    if depth is None:
        depth = render_depth(means3D, scales, rotations, opacities, cam)
    depth_flat = depth.reshape(-1)
    if sync_free:
        if n_surface_points <= 0 or cpu_randperm:
            raise ValueError("sync_free needs a positive n_surface_points and the device-side pixel subset")
        return _sample_sync_free(means3D, scales, rotations, opacities, cam, depth_flat.to(torch.float32).contiguous(), int(n_surface_points),
                                 tuple(surface_levels), n_points_in_range, range_size, density_factor, K, return_normals, seed)
    valid_pix = torch.logical_not(depth_flat < 0.).nonzero(as_tuple=True)[0]
    n_valid = valid_pix.shape[0]
    if n_surface_points == -1:
        picked = valid_pix
    else:
        n = min(int(n_surface_points), n_valid)
        idx = torch.randperm(n_valid)[:n].to(dev) if cpu_randperm else random_prefix_of_permutation(n_valid, n, dev)
        picked = valid_pix[idx]
    world = unproject_pixels(picked, depth_flat, cam)
    nbr = knn_points(world[None], means3D[None], K=K).idx[0]
    gaussian_idx = nbr[:, 0]
    cam_center = cam.campos.reshape(1, 3)
    stds = view_std(means3D, rotations, scales, cam.campos)                                            # :1971-1972
    B = scaled_rotation(rotations, scales, inverse_scales=True)                                        # :730-736
    res = level_set_points(world, nbr, cam_center, means3D, B, opacities.reshape(-1, 1), stds, surface_levels=tuple(surface_levels),
                           n_points_in_range=n_points_in_range, range_size=range_size, density_factor=density_factor,
                           return_normals=return_normals)
    out = {}
    for lv in surface_levels:
        r = res[lv]
        rows = r["valid_idx"]
        out[lv] = dict(intersection_points=r["intersection_points"], pixel_idx=picked[rows], gaussian_idx=gaussian_idx[rows],
                       normals=r["normals"])
    return out


def _sample_sync_free(means3D, scales, rotations, opacities, cam, depth_flat, n, surface_levels, n_points_in_range, range_size,
                      density_factor, K, return_normals, seed):
    lib = _lib.load()
    dev = means3D.device
    if seed is None:
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())   # (the CPU generator: seeded by torch.manual_seed, no device wait)
    picked = torch.empty(n, dtype=torch.int64, device=dev)
    words = torch.empty(2, dtype=torch.int32, device=dev)        # [count, n_valid]
    scratch = torch.empty(int(lib.sgr_pick_pixels_scratch_bytes(depth_flat.numel())), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.sgr_pick_pixels(int(depth_flat.numel()), _p(depth_flat), n, C.c_uint32(seed & 0xFFFFFFFF), _p(picked), _p(words),
                                 C.c_void_p(words.data_ptr() + 4), _p(scratch), _stream(dev))
    if rc < 0:
        raise RuntimeError(f"sgr_pick_pixels failed ({rc})")
    world = unproject_pixels(picked, depth_flat, cam)
    nbr = knn_points(world[None], means3D[None], K=K).idx[0]
    gaussian_idx = nbr[:, 0].contiguous()
    stds = view_std(means3D, rotations, scales, cam.campos)
    B = scaled_rotation(rotations, scales, inverse_scales=True)
    valid, pts, nrm = level_set_points(world, nbr, cam.campos.reshape(1, 3), means3D, B, opacities.reshape(-1, 1), stds,
                                       surface_levels=surface_levels, n_points_in_range=n_points_in_range, range_size=range_size,
                                       density_factor=density_factor, return_normals=return_normals, raw=True)
    L = len(surface_levels)
    rows = torch.empty(L, n, dtype=torch.int64, device=dev)
    pts_c = torch.empty(L, n, 3, device=dev)
    nrm_c = torch.empty(L, n, 3, device=dev) if return_normals else None
    pix_c = torch.empty(L, n, dtype=torch.int64, device=dev)
    gid_c = torch.empty(L, n, dtype=torch.int64, device=dev)
    counts = torch.empty(L, dtype=torch.int32, device=dev)
    scr2 = torch.empty(int(lib.sgr_compact_level_rows_scratch_bytes(n, L)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.sgr_compact_level_rows(n, L, _p(valid), _p(words), _p(pts), _p(nrm) if nrm is not None else None, _p(picked),
                                        _p(gaussian_idx), _p(rows), _p(pts_c), _p(nrm_c) if nrm_c is not None else None, _p(pix_c),
                                        _p(gid_c), _p(counts), _p(scr2), _stream(dev))
    if rc < 0:
        raise RuntimeError(f"sgr_compact_level_rows failed ({rc})")
    out = {}
    for i, lv in enumerate(surface_levels):
        out[lv] = dict(intersection_points=pts_c[i], pixel_idx=pix_c[i], gaussian_idx=gid_c[i], normals=nrm_c[i] if return_normals else None,
                       count=counts[i], n_picked=words[0], n_valid_pixels=words[1], picked=picked)
    return out


def trim(result):
    """a `sync_free` result with every level's tensors cut to their counts (ONE host wait for all levels)"""
    levels = list(result)
    counts = torch.stack([result[lv]["count"] for lv in levels]).cpu().tolist()
    out = {}
    for lv, c in zip(levels, counts):
        r = result[lv]
        out[lv] = dict(intersection_points=r["intersection_points"][:c], pixel_idx=r["pixel_idx"][:c], gaussian_idx=r["gaussian_idx"][:c],
                       normals=(r["normals"][:c] if r["normals"] is not None else None))
    return out
