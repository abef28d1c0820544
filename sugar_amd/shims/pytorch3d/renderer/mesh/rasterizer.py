"""`pytorch3d.renderer.mesh.rasterizer`: RasterizationSettings, Fragments, MeshRasterizer (pytorch3d 0.7.4's classes restated
for the arguments SuGaR uses: sugar_scene/sugar_model.py:1880-1893, sugar_extractors/coarse_mesh.py:216-225)."""
from __future__ import annotations

from typing import NamedTuple, Optional

import torch

from .rasterize_meshes import rasterize_meshes


class Fragments(NamedTuple):
    pix_to_face: torch.Tensor            # (N, H, W, K) int64, index into the packed faces, -1 = no face
    zbuf: torch.Tensor                   # (N, H, W, K) view-space depth of the face at the pixel centre, -1 = no face
    bary_coords: Optional[torch.Tensor]  # (N, H, W, K, 3)
    dists: Optional[torch.Tensor]        # (N, H, W, K) signed squared NDC distance to the face's outline (negative inside)


class RasterizationSettings:
    def __init__(self, image_size=256, blur_radius: float = 0.0, faces_per_pixel: int = 1, bin_size=None, max_faces_per_bin=None,
                 perspective_correct=None, clip_barycentric_coords=None, cull_backfaces: bool = False, z_clip_value=None,
                 cull_to_frustum: bool = False):
        self.image_size, self.blur_radius, self.faces_per_pixel = image_size, blur_radius, faces_per_pixel
        self.bin_size, self.max_faces_per_bin = bin_size, max_faces_per_bin
        self.perspective_correct, self.clip_barycentric_coords = perspective_correct, clip_barycentric_coords
        self.cull_backfaces, self.z_clip_value, self.cull_to_frustum = cull_backfaces, z_clip_value, cull_to_frustum


class MeshRasterizer(torch.nn.Module):
    def __init__(self, cameras=None, raster_settings=None):
        super().__init__()
        self.cameras = cameras
        self.raster_settings = RasterizationSettings() if raster_settings is None else raster_settings

    def to(self, device):
        if self.cameras is not None:
            self.cameras = self.cameras.to(device)
        return self

    def transform(self, meshes_world, **kwargs):
        """world -> view -> NDC x, y; z keeps the view-space depth (MeshRasterizer.transform)"""
        cameras = kwargs.get("cameras", self.cameras)
        if cameras is None:
            raise ValueError("Cameras must be specified either at initialization or in the forward pass of MeshRasterizer")
        if len(cameras) != 1 and len(cameras) != len(meshes_world):
            raise ValueError(f"Wrong number ({len(cameras)}) of cameras for {len(meshes_world)} meshes")
        eps = kwargs.get("eps", None)
        from pytorch3d.structures import Meshes
        out_verts = []
        for i, verts_world in enumerate(meshes_world.verts_list()):
            cam = cameras if len(cameras) == 1 else cameras[i]
            verts_view = cam.get_world_to_view_transform().transform_points(verts_world, eps=eps)
            verts_ndc = cam.get_projection_transform().transform_points(verts_view, eps=eps)  # FoV cameras project straight to NDC
            out_verts.append(torch.cat([verts_ndc[..., :2], verts_view[..., 2:3]], dim=-1))
        return Meshes(out_verts, meshes_world.faces_list(), textures=meshes_world.textures)

    @staticmethod
    def resolved_settings(rs, cameras):
        """(clip_barycentric_coords, perspective_correct, z_clip_value) as MeshRasterizer.forward derives them from the settings and
        the camera: hard rasterization keeps unclipped barycentrics, perspective cameras interpolate perspective-correctly and clip
        at znear / 2"""
        clip_barycentric_coords = rs.clip_barycentric_coords
        if clip_barycentric_coords is None:
            clip_barycentric_coords = rs.blur_radius > 0.0
        perspective_correct = rs.perspective_correct
        if perspective_correct is None:
            perspective_correct = cameras.is_perspective()
        z_clip = rs.z_clip_value
        if z_clip is None and cameras.is_perspective():
            znear = cameras.get_znear()
            if torch.is_tensor(znear):
                znear = znear.min().item()
            z_clip = None if znear is None else znear / 2
        return clip_barycentric_coords, perspective_correct, z_clip

    def forward(self, meshes_world, **kwargs) -> Fragments:
        meshes_proj = self.transform(meshes_world, **kwargs)
        rs = kwargs.get("raster_settings", self.raster_settings)
        cameras = kwargs.get("cameras", self.cameras)
        clip_barycentric_coords, perspective_correct, z_clip = self.resolved_settings(rs, cameras)
        pix_to_face, zbuf, bary_coords, dists = rasterize_meshes(
            meshes_proj, image_size=rs.image_size, blur_radius=rs.blur_radius, faces_per_pixel=rs.faces_per_pixel, bin_size=rs.bin_size,
            max_faces_per_bin=rs.max_faces_per_bin, clip_barycentric_coords=clip_barycentric_coords,
            perspective_correct=perspective_correct, cull_backfaces=rs.cull_backfaces, z_clip_value=z_clip,
            cull_to_frustum=rs.cull_to_frustum)
        return Fragments(pix_to_face=pix_to_face, zbuf=zbuf, bary_coords=bary_coords, dists=dists)
