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

import torch

from .field import level_set_points, scaled_rotation
from .knn import knn_points


def view_depth(means3D: torch.Tensor, viewmatrix: torch.Tensor) -> torch.Tensor:
    """z of every centre in the camera's frame, [P,1] (`get_world_to_view_transform().transform_points(points)[..., 2:]`)"""
    return means3D @ viewmatrix[:3, 2:3] + viewmatrix[3, 2]


def render_depth(means3D, scales, rotations, opacities, cam, bg_value: float = -1.0):
    """[H,W] depth image: the Gaussian rasterizer with depth as the colour of every Gaussian (sugar_model.py:1901-1911)"""
    from .diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = means3D.device
    depth_rgb = view_depth(means3D, cam.viewmatrix).expand(-1, 3).contiguous()
    st = GaussianRasterizationSettings(int(cam.image_height), int(cam.image_width), cam.tanfovx, cam.tanfovy,
                                       torch.full((3,), float(bg_value), device=dev), 1.0, cam.viewmatrix, cam.projmatrix, 0,
                                       cam.campos, False, False)
    img, _ = GaussianRasterizer(st)(means3D, torch.zeros_like(means3D), opacities, colors_precomp=depth_rgb, scales=scales,
                                    rotations=rotations)
    return img[0]


def unproject_pixels(picked: torch.Tensor, depth_flat: torch.Tensor, cam) -> torch.Tensor:
    """World points of the picked pixels.  The reference lays an NDC grid over the image (x = W/m - 2 col / (m - 1), +x to the LEFT,
    +y UP, m = min(W, H); sugar_model.py:1934-1941) and un-projects through a pytorch3d camera whose focal length in NDC units is
    2 fx / m; the camera frame of the rasterizer (COLMAP: +x right, +y down) is that frame with x and y negated."""
    H, W = int(cam.image_height), int(cam.image_width)
    m = min(W, H)
    rows = torch.div(picked, W, rounding_mode="floor")
    cols = picked - rows * W
    ndc_x = W / m - (cols.to(torch.float32) / (m - 1)) * 2
    ndc_y = H / m - (rows.to(torch.float32) / (m - 1)) * 2
    z = depth_flat[picked]
    f_ndc_x = (W / (2.0 * cam.tanfovx)) * 2.0 / m
    f_ndc_y = (H / (2.0 * cam.tanfovy)) * 2.0 / m
    xc = -ndc_x * z / f_ndc_x
    yc = -ndc_y * z / f_ndc_y
    c2w = torch.linalg.inv(cam.viewmatrix)  # row-vector convention: [x y z 1] @ inverse(W2C^T)
    pts = torch.stack([xc, yc, z, torch.ones_like(z)], dim=-1) @ c2w
    return pts[:, :3].contiguous()


def _rotate_inverse(q: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """R(q)^T v for unit quaternions (real part first): `quaternion_apply(quaternion_invert(q), v)`"""
    r, x, y, z = q.unbind(-1)
    vx, vy, vz = v.unbind(-1)
    # rows of R^T = columns of R
    ox = (1 - 2 * (y * y + z * z)) * vx + 2 * (x * y + r * z) * vy + 2 * (x * z - r * y) * vz
    oy = 2 * (x * y - r * z) * vx + (1 - 2 * (x * x + z * z)) * vy + 2 * (y * z + r * x) * vz
    oz = 2 * (x * z + r * y) * vx + 2 * (y * z - r * x) * vy + (1 - 2 * (x * x + y * y)) * vz
    return torch.stack([ox, oy, oz], dim=-1)


@torch.no_grad()
def sample_level_sets(means3D, scales, rotations, opacities, cam, *, n_surface_points: int = 124_000,
                      surface_levels=(0.1, 0.3, 0.5), n_points_in_range: int = 21, range_size: float = 3.0,
                      density_factor: float = 1.0, K: int = 16, return_normals: bool = True, cpu_randperm: bool = False,
                      depth: torch.Tensor | None = None):
    """One sampling pass for one view.  Returns {level: dict(intersection_points[n,3], pixel_idx[n], gaussian_idx[n], normals[n,3])}.
    `scales` / `rotations` / `opacities` are the ACTIVATED values ([P,3], unit [P,4], [P,1]); `cpu_randperm` draws the pixel subset on
    the CPU as the reference does (:1955), so that a seeded run picks the reference's pixels."""
    if not means3D.is_cuda:
        raise RuntimeError("sample_level_sets needs tensors on a ROCm device; there is no CPU fallback")
    from .sugar_patch import random_prefix_of_permutation
    dev = means3D.device
    W = int(cam.image_width)
    if depth is None:
        depth = render_depth(means3D, scales, rotations, opacities, cam)
    depth_flat = depth.reshape(-1)
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
    to_cam = torch.nn.functional.normalize(cam_center - means3D, dim=-1)
    stds = (scales * _rotate_inverse(rotations, to_cam)).norm(dim=-1)                                  # :1971-1972
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
