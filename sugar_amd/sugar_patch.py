"""Routes SuGaR's Gaussian-buffer-sharing tensor code to the HIP kernels of this package WITHOUT touching the reference's
files: `install(sm)` replaces four methods of the `SuGaR` class of an imported `sugar_scene.sugar_model` module

    SuGaR.get_points_rgb                                   sugar_model.py:839-883    -> sgr_sh_to_rgb_*        (shcolor.py)
    SuGaR.get_covariance(return_sqrt=True, ...)            sugar_model.py:729-736    -> sgr_scaled_rotation_*  (field.py)
    SuGaR.get_field_values                                 sugar_model.py:1247-1316  -> sgr_density_field_*    (field.py)
    SuGaR.compute_level_surface_points_from_camera_fast    sugar_model.py:1848-2083  -> sgr_level_set_points + the rasterizer
                                                                                        + the HIP k-NN (use_gaussian_depth=True)

with functions of the same signature and return values.  Each replacement takes over only what it implements -- tensors on a
ROCm device and the argument combinations listed below -- and hands every other call to the reference's own method, which
stays reachable as `SuGaR._sugar_amd_original[name]`.  `uninstall(sm)` puts the originals back.

The replacements are written against the attributes SuGaR exposes (`points`, `scaling`, `quaternions`, `strengths`,
`sh_coordinates`, `knn_idx`, `knn_to_track`, `image_height`, `image_width`, `get_beta`, `render_image_gaussian_rasterizer`,
`get_gaussians_closest_to_samples`), so the GPU tests can drive them with a small stand-in object holding a fixture of the
reference model's state (the reference tree does not exist on the GPU box).
"""
from __future__ import annotations

import functools

import numpy as np
import torch

PATCHED = ("get_points_rgb", "get_covariance", "get_field_values", "compute_level_surface_points_from_camera_fast")


def _on_gpu(*ts):
    return all(t is None or (torch.is_tensor(t) and t.is_cuda) for t in ts) and any(t is not None for t in ts)


# ----------------------------------------------------------------------------------------------- get_points_rgb
def get_points_rgb(self, positions=None, camera_centers=None, directions=None, sh_levels=None, sh_coordinates=None, _orig=None):
    pos = self.points if positions is None else positions
    if _orig is not None and (sh_levels is None or not _on_gpu(pos) or (camera_centers is None and directions is None)):
        return _orig(self, positions=positions, camera_centers=camera_centers, directions=directions, sh_levels=sh_levels,
                     sh_coordinates=sh_coordinates)
    from .shcolor import get_points_rgb as hip
    return hip(self, positions=positions, camera_centers=camera_centers, directions=directions, sh_levels=sh_levels,
               sh_coordinates=sh_coordinates)


# ----------------------------------------------------------------------------------------------- get_covariance
def get_covariance(self, return_full_matrix=False, return_sqrt=False, inverse_scales=False, _orig=None):
    q = self.quaternions
    if return_sqrt and q.is_cuda:
        from .field import scaled_rotation
        return scaled_rotation(q, self.scaling, inverse_scales)  # sugar_model.py:730-736
    return _orig(self, return_full_matrix=return_full_matrix, return_sqrt=return_sqrt, inverse_scales=inverse_scales)


# ----------------------------------------------------------------------------------------------- get_field_values
def get_field_values(self, x, gaussian_idx=None, closest_gaussians_idx=None, gaussian_strengths=None, gaussian_centers=None,
                     gaussian_inv_scaled_rotation=None, return_sdf=True, density_threshold=1., density_factor=1.,
                     return_sdf_grad=False, sdf_grad_max_value=10., opacity_min_clamp=1e-16,
                     return_closest_gaussian_opacities=False, return_beta=False, _orig=None, _density_field=None):
    """sugar_model.py:1247-1316.  The neighbour gather, the warp, the exponentials and their sum (and, in the backward, the
    scatter of their gradients) are the fused kernels; everything after the densities is the reference's own arithmetic on
    [N] / [N,K] tensors.  `return_sdf_grad=True` needs the per-neighbour warped shifts and goes to the reference's code."""
    if _orig is not None and (return_sdf_grad or not x.is_cuda):
        return _orig(self, x, gaussian_idx=gaussian_idx, closest_gaussians_idx=closest_gaussians_idx,
                     gaussian_strengths=gaussian_strengths, gaussian_centers=gaussian_centers,
                     gaussian_inv_scaled_rotation=gaussian_inv_scaled_rotation, return_sdf=return_sdf,
                     density_threshold=density_threshold, density_factor=density_factor, return_sdf_grad=return_sdf_grad,
                     sdf_grad_max_value=sdf_grad_max_value, opacity_min_clamp=opacity_min_clamp,
                     return_closest_gaussian_opacities=return_closest_gaussian_opacities, return_beta=return_beta)
    if return_sdf_grad:
        raise NotImplementedError("return_sdf_grad=True is served by the reference's own method")
    if _density_field is None:
        from .field import density_field as _density_field
    if gaussian_strengths is None:
        gaussian_strengths = self.strengths
    if gaussian_centers is None:
        gaussian_centers = self.points
    if gaussian_inv_scaled_rotation is None:
        gaussian_inv_scaled_rotation = self.get_covariance(return_full_matrix=True, return_sqrt=True, inverse_scales=True)
    if closest_gaussians_idx is None:
        closest_gaussians_idx = self.knn_idx[gaussian_idx]
    fields = {}
    neighbor_opacities, densities = _density_field(x, closest_gaussians_idx, gaussian_centers, gaussian_inv_scaled_rotation,
                                                   gaussian_strengths, density_factor)                       # :1266-1276
    fields['density'] = densities.clone()
    density_mask = densities >= 1.
    densities = torch.where(density_mask, densities / (densities.detach() + 1e-12), densities)               # :1278-1279
    if return_closest_gaussian_opacities:
        fields['closest_gaussian_opacities'] = neighbor_opacities
    if return_sdf or return_sdf_grad or return_beta:
        beta = self.get_beta(x, closest_gaussians_idx=closest_gaussians_idx, closest_gaussians_opacities=neighbor_opacities,
                             densities=densities, opacity_min_clamp=opacity_min_clamp)
        clamped_densities = densities.clamp(min=opacity_min_clamp)
    if return_beta:
        fields['beta'] = beta
    if return_sdf:
        fields['sdf'] = beta * (torch.sqrt(-2. * torch.log(clamped_densities))
                                - np.sqrt(-2. * np.log(min(density_threshold, 1.))))                         # :1301-1306
    return fields


def random_prefix_of_permutation(n: int, k: int, device) -> torch.Tensor:
    """`torch.randperm(n, device=device)[:k]` in distribution -- k distinct indices of range(n), every ordered k-tuple equally likely --
    without permuting all n when k is a small part of it (the extractor keeps 124k of ~2M pixels per view, coarse_mesh.py:243,276:
    a device randperm of 2M is a 0.5 ms sort).  Uniform draws WITH replacement, repeated values dropped after their first occurrence,
    first k kept: sequential sampling without replacement."""
    if k * 8 > n:
        return torch.randperm(n, device=device)[:k]
    m = k + k // 4 + 64
    while True:
        draws = torch.randint(n, (m,), device=device)
        vals, order = torch.sort(draws, stable=True)
        first = torch.ones(m, dtype=torch.bool, device=device)
        first[1:] = vals[1:] != vals[:-1]               # in a STABLE sort the first of a run of equal values is the earliest draw
        keep = torch.empty(m, dtype=torch.bool, device=device)
        keep[order] = first
        kept = draws[keep]                               # first occurrences, in draw order
        if kept.shape[0] >= k:
            return kept[:k]
        m *= 2


# ------------------------------------------------------------------------- compute_level_surface_points_from_camera_fast
def compute_level_surface_points_from_camera_fast(
        self, nerf_cameras=None, cam_idx=0, rasterizer=None, surface_levels=[0.1, 0.3, 0.5], n_surface_points=-1,
        primitive_types=None, triangle_scale=None, splat_mesh=True, n_points_in_range=21, range_size=3.,
        n_points_per_pass=2_000_000, density_factor=1., return_pixel_idx=False, return_gaussian_idx=False, return_normals=False,
        compute_flat_normals=False, compute_intersection_for_flat_gaussian=False, use_gaussian_depth=False,
        just_use_depth_as_level=False, _orig=None, _level_set_points=None):
    """sugar_model.py:1848-2083.  The depth map and the front Gaussian of every pixel come from
      * `use_gaussian_depth=False` (what coarse_mesh.py:26 hard-codes): the splat mesh's nearest-face z-buffer -- the reference's
        own `splat_mesh` (:695-727) rasterized by `rasterizer` (the stand-in `pytorch3d.renderer.MeshRasterizer` on the HIP
        z-buffer kernel, sgr_rasterize_meshes), `depth = fragments.zbuf[0, ..., 0]` (:1927-1928), front Gaussian =
        `pix_to_face // n_triangles_per_gaussian`, neighbours = its row of `knn_idx` (:1966-1968); the view-dependent texture the
        reference attaches to the mesh first (:1914-1925) is not computed: rasterization does not read it;
      * `use_gaussian_depth=True` (:1901-1911, :1962-1964): a render of the Gaussian rasterizer with the view-space depth as
        colour, neighbours from the k-NN query of the unprojected pixel.
    The pixels are unprojected with the camera, and the 21 ray samples x 16 neighbours per pixel, the level crossings and the
    normals are ONE kernel (the reference materialises [n * 21, 16, 3, 3] tensors in passes of 2M samples).  The flat-Gaussian
    variants go to the reference's own method."""
    on_gpu = self.points.is_cuda
    if _orig is not None and (not on_gpu or compute_flat_normals or compute_intersection_for_flat_gaussian or just_use_depth_as_level):
        return _orig(self, nerf_cameras=nerf_cameras, cam_idx=cam_idx, rasterizer=rasterizer, surface_levels=surface_levels,
                     n_surface_points=n_surface_points, primitive_types=primitive_types, triangle_scale=triangle_scale,
                     splat_mesh=splat_mesh, n_points_in_range=n_points_in_range, range_size=range_size,
                     n_points_per_pass=n_points_per_pass, density_factor=density_factor, return_pixel_idx=return_pixel_idx,
                     return_gaussian_idx=return_gaussian_idx, return_normals=return_normals,
                     compute_flat_normals=compute_flat_normals,
                     compute_intersection_for_flat_gaussian=compute_intersection_for_flat_gaussian,
                     use_gaussian_depth=use_gaussian_depth, just_use_depth_as_level=just_use_depth_as_level)
    if compute_flat_normals or compute_intersection_for_flat_gaussian or just_use_depth_as_level:
        raise NotImplementedError("the flat-Gaussian variants run the reference's own method")
    if _level_set_points is None:
        from .field import level_set_points as _level_set_points
    from pytorch3d.transforms import quaternion_apply, quaternion_invert
    if nerf_cameras is None:
        nerf_cameras = self.nerfmodel.training_cameras
    if primitive_types is not None:
        self.primitive_types = primitive_types
    if triangle_scale is not None:
        self.triangle_scale = triangle_scale
    p3d_cameras = nerf_cameras.p3d_cameras[cam_idx]
    device = self.points.device
    H, W = self.image_height, self.image_width
    fragments = None
    if use_gaussian_depth:  # splatted depth (:1901-1911)
        point_depth = p3d_cameras.get_world_to_view_transform().transform_points(self.points)[..., 2:].expand(-1, 3)
        depth = self.render_image_gaussian_rasterizer(camera_indices=cam_idx, bg_color=torch.Tensor([-1., -1., -1.]).to(device),
                                                      sh_deg=0, compute_covariance_in_rasterizer=True, return_2d_radii=False,
                                                      use_same_scale_in_all_directions=False,
                                                      point_colors=point_depth).contiguous()[..., 0]
    else:                   # nearest face of the splat mesh (:1880-1893, :1912-1928)
        if rasterizer is None:
            from pytorch3d.renderer import MeshRasterizer, RasterizationSettings
            rasterizer = MeshRasterizer(cameras=p3d_cameras, raster_settings=RasterizationSettings(
                image_size=(H, W), blur_radius=0.0, faces_per_pixel=10, max_faces_per_bin=50_000))
        fragments = _splat_fragments(self, p3d_cameras, rasterizer) if splat_mesh else None
        if fragments is None:
            mesh = self.splat_mesh(p3d_cameras) if splat_mesh else self.mesh
            fragments = rasterizer(mesh, cameras=p3d_cameras)
        depth = fragments.zbuf[0, ..., 0]
    # Pixels with a depth, in raster order -- the rows `x[~no_proj_mask]` keeps (:1929-1931, :1942-1946) -- found ONCE; the random
    # subset (:1948-1957), the back-projection (:1932-1959) and the front Gaussians are then computed for the picked pixels only.
    # (The reference builds [H*W, 3] NDC tables, masks them three times -- each a host round trip -- and indexes the result; same
    # values, same order, same permutation.)
    depth_flat = depth.reshape(-1)
    valid_pix = torch.logical_not(depth_flat < 0.).nonzero(as_tuple=True)[0]
    n_valid = valid_pix.shape[0]
    if n_surface_points == -1:
        picked = valid_pix
    else:
        n_surface_points = min(n_surface_points, n_valid)
        # (the reference draws this permutation on the CPU, :1955: ~15 ms at 1080p plus the copy; same distribution on the device.
        # `_sugar_amd_cpu_randperm = True` on the model draws it on the CPU as the reference does, so that a seeded run picks the
        # reference's pixels: tests/test_gpu_reference_sugar.py)
        if getattr(self, "_sugar_amd_cpu_randperm", False):
            ndc_points_idx = torch.randperm(n_valid)[:n_surface_points].to(device)
        else:
            ndc_points_idx = random_prefix_of_permutation(n_valid, n_surface_points, device)
        picked = valid_pix[ndc_points_idx]
    m = min(W, H)
    rows = torch.div(picked, W, rounding_mode="floor")
    cols = picked - rows * W
    ndc_x = W / m - (cols.to(torch.float32) / (m - 1)) * 2          # the pixel tables of :1934-1941 as index arithmetic
    ndc_y = H / m - (rows.to(torch.float32) / (m - 1)) * 2
    ndc_points = torch.stack((ndc_x, ndc_y, depth_flat[picked]), dim=-1)[None]
    all_world_points = p3d_cameras.unproject_points(ndc_points, scaled_depth_input=False).view(-1, 3)
    if use_gaussian_depth:
        closest_gaussians_idx = self.get_gaussians_closest_to_samples(all_world_points)                # :1963 (HIP k-NN)
        gaussian_idx = closest_gaussians_idx[..., 0]
    else:                                                                                              # :1966-1968
        gaussian_idx = fragments.pix_to_face[0, ..., 0].reshape(-1)[picked] // self.n_triangles_per_gaussian
        closest_gaussians_idx = self.knn_idx[gaussian_idx]
    cam_center = p3d_cameras.get_camera_center()
    if self.points.is_cuda:                                                                            # :1971-1972, one launch
        from .sampler import view_std
        gaussian_standard_deviations = view_std(self.points.detach(), self.quaternions.detach(), self.scaling.detach(), cam_center.detach())
    else:
        gaussian_to_camera = torch.nn.functional.normalize(cam_center - self.points, dim=-1)
        gaussian_standard_deviations = (self.scaling * quaternion_apply(quaternion_invert(self.quaternions), gaussian_to_camera)).norm(dim=-1)
    B = self.get_covariance(return_full_matrix=True, return_sqrt=True, inverse_scales=True)
    with torch.no_grad():
        res = _level_set_points(all_world_points.detach(), closest_gaussians_idx, cam_center.detach(), self.points.detach(),
                                B.detach(), self.strengths.detach(), gaussian_standard_deviations.detach(),
                                surface_levels=tuple(surface_levels), n_points_in_range=n_points_in_range, range_size=range_size,
                                density_factor=density_factor, return_normals=return_normals)
    all_outputs = {}
    for surface_level in surface_levels:
        r = res[surface_level]
        outputs = {'intersection_points': r['intersection_points']}
        rows = r.get('valid_idx')
        if rows is None and (return_pixel_idx or return_gaussian_idx):
            rows = r['valid'].nonzero(as_tuple=True)[0]
        if return_pixel_idx:
            outputs['pixel_idx'] = picked[rows]
        if return_gaussian_idx:
            outputs['gaussian_idx'] = gaussian_idx[rows]
        if return_normals:
            outputs['normals'] = r['normals']
        all_outputs[surface_level] = outputs
    return all_outputs


def _splat_fragments(self, p3d_cameras, rasterizer):
    """The fragments `rasterizer(self.splat_mesh(camera), cameras=camera)` returns, without the 4 P-vertex tensors in between: the
    splat mesh's faces come out of ONE kernel that reads the Gaussian buffers (sgr_splat_mesh_face_verts: triangle_vertices :481-514,
    splat_mesh :695-716 and the rasterizer's vertex transform), then clipping + z-buffer as the stand-in MeshRasterizer does.  Only
    with the stand-in rasterizer of sugar_amd.shims (a real pytorch3d rasterizer gets the reference's mesh: returns None); the
    interpolation weights and distances nobody reads here are not produced (None in the tuple)."""
    import importlib
    try:
        fn = getattr(importlib.import_module("pytorch3d.renderer.mesh.rasterize_meshes"), "rasterize_face_verts_ndc", None)
        from pytorch3d.renderer.mesh.rasterizer import Fragments, MeshRasterizer
    except ImportError:
        return None
    if fn is None or not isinstance(rasterizer, MeshRasterizer) or not hasattr(MeshRasterizer, "resolved_settings"):
        return None
    if not self.points.is_cuda:
        return None  # (CPU tensors: the host-logic tests run the reference's own splat_mesh on the oracle backend)
    from .mesh_raster import splat_face_verts
    rs = rasterizer.raster_settings
    if rs.blur_radius != 0.0 or self.primitive_types not in ("diamond", "square"):
        return None
    prim = self._diamond_verts if self.primitive_types == "diamond" else self._square_verts
    face_verts = splat_face_verts(self.points, self.scaling, self.quaternions, prim, self.triangle_scale,
                                  p3d_cameras.get_world_to_view_transform().get_matrix(), p3d_cameras.get_projection_transform().get_matrix())
    clip_bary, persp, z_clip = MeshRasterizer.resolved_settings(rs, p3d_cameras)
    n = face_verts.shape[0]
    dev = face_verts.device
    # (faces_per_pixel = 1 whatever the caller's settings say -- the extractor asks for 10, coarse_mesh.py:216-221: the sampler reads slot
    # 0 only (:1928, :1966) and slot 0 of the K nearest faces is the nearest face; these fragments never leave this module)
    p2f, zbuf, bary, dists = fn(face_verts, torch.zeros(1, dtype=torch.int64, device=dev), torch.full((1,), n, dtype=torch.int64, device=dev),
                                rs.image_size, rs.blur_radius, 1, persp, clip_bary, rs.cull_backfaces, z_clip,
                                rs.cull_to_frustum, want_bary=False, want_dists=False)
    return Fragments(pix_to_face=p2f, zbuf=zbuf, bary_coords=bary, dists=dists)


_IMPL = dict(get_points_rgb=get_points_rgb, get_covariance=get_covariance, get_field_values=get_field_values,
             compute_level_surface_points_from_camera_fast=compute_level_surface_points_from_camera_fast)


def install(sugar_model_module, names=PATCHED):
    """Patch `sugar_model_module.SuGaR` (the imported reference module).  Idempotent; returns the patched names."""
    cls = sugar_model_module.SuGaR
    saved = cls.__dict__.get("_sugar_amd_original")
    if saved is None:
        saved = {}
        cls._sugar_amd_original = saved
    done = []
    for name in names:
        if name in saved:
            done.append(name)
            continue
        orig = getattr(cls, name)
        impl = _IMPL[name]
        wrapped = functools.partialmethod(impl, _orig=orig)
        saved[name] = orig
        setattr(cls, name, wrapped)
        done.append(name)
    return done


ROW_GATHER_PROPERTIES = ("points", "scaling", "quaternions")
ROW_GATHER_METHODS = ("get_normals",)


def install_row_gathers(sugar_model_module):
    """`SuGaR.points` / `.scaling` / `.quaternions` (properties, sugar_model.py:383-479) and `SuGaR.get_normals()` (:946-968) return
    the tensor they always returned, viewed as a `sugar_amd.row_gather.RowGatherTensor`: indexing it with an int64 CUDA tensor --
    what the regulariser does by the million (:922-925, coarse_sdf.py:690-692) -- takes the HIP scatter-add for its backward.
    CPU tensors pass through untouched.  Idempotent; `uninstall_row_gathers` restores the class."""
    from .row_gather import as_row_gather
    cls = sugar_model_module.SuGaR
    saved = cls.__dict__.get("_sugar_amd_row_gather_original")
    if saved is None:
        saved = {}
        cls._sugar_amd_row_gather_original = saved
    for name in ROW_GATHER_PROPERTIES:
        if name in saved:
            continue
        prop = cls.__dict__[name]
        saved[name] = prop
        setattr(cls, name, property(lambda self, _get=prop.fget: as_row_gather(_get(self)), prop.fset, prop.fdel, prop.__doc__))
    for name in ROW_GATHER_METHODS:
        if name in saved:
            continue
        orig = cls.__dict__[name]
        saved[name] = orig

        def wrapped(self, *a, _orig=orig, **k):
            return as_row_gather(_orig(self, *a, **k))
        wrapped.__name__ = name
        wrapped.__doc__ = orig.__doc__
        setattr(cls, name, wrapped)
    return list(saved)


def uninstall_row_gathers(sugar_model_module):
    cls = sugar_model_module.SuGaR
    for name, orig in list(cls.__dict__.get("_sugar_amd_row_gather_original", {}).items()):
        setattr(cls, name, orig)
    if "_sugar_amd_row_gather_original" in cls.__dict__:
        delattr(cls, "_sugar_amd_row_gather_original")


def uninstall(sugar_model_module):
    uninstall_row_gathers(sugar_model_module)
    cls = sugar_model_module.SuGaR
    for name, orig in list(cls.__dict__.get("_sugar_amd_original", {}).items()):
        setattr(cls, name, orig)
    cls._sugar_amd_original = {}
