"""`pytorch3d.renderer.mesh.rasterize_meshes.rasterize_meshes` for hard rasterization (blur_radius = 0): the Python half of
pytorch3d 0.7.4's function (face gathering, near-plane clipping, conversion back) restated, the extension call replaced by
the HIP z-buffer (`sugar_amd.mesh_raster.rasterize_face_verts` -> `sgr_rasterize_meshes`).  `bin_size` and `max_faces_per_bin`
are accepted and have no effect: there are no coarse bins to overflow (pytorch3d drops the faces of a bin beyond
`max_faces_per_bin`; SuGaR raises it to 50 000 to avoid exactly that, sugar_model.py:1882)."""
from __future__ import annotations

import torch

from .clip import ClipFrustum, clip_faces, convert_clipped_rasterization_to_original_faces


def rasterize_face_verts_ndc(face_verts, first_t, count_t, image_size, blur_radius=0.0, faces_per_pixel=8, perspective_correct=False,
                             clip_barycentric_coords=False, cull_backfaces=False, z_clip_value=None, cull_to_frustum=False,
                             want_bary=True, want_dists=True):
    """everything `rasterize_meshes` does behind the face gather: near-plane clipping, the z-buffer, the conversion back.  Also
    the entry of callers that produce face_verts themselves (sugar_amd.sugar_patch: the fused splat-mesh kernel)."""
    from sugar_amd.mesh_raster import rasterize_face_verts
    if isinstance(image_size, int):
        image_size = (image_size, image_size)
    clipped = None
    if z_clip_value is not None or cull_to_frustum:
        frustum = ClipFrustum(left=-1, right=1, top=-1, bottom=1, perspective_correct=perspective_correct, cull=cull_to_frustum,
                              z_clip_value=z_clip_value)
        clipped = clip_faces(face_verts, first_t, count_t, frustum)
        face_verts, first_t, count_t = clipped.face_verts, clipped.mesh_to_face_first_idx, clipped.num_faces_per_mesh
    pix_to_face, zbuf, bary, dists = rasterize_face_verts(face_verts, first_t, count_t, image_size, blur_radius, faces_per_pixel,
                                                          perspective_correct, clip_barycentric_coords, cull_backfaces,
                                                          want_bary=want_bary, want_dists=want_dists)
    if clipped is not None:
        pix_to_face, bary = convert_clipped_rasterization_to_original_faces(pix_to_face, bary, clipped)
    return pix_to_face, zbuf, bary, dists


def rasterize_meshes(meshes, image_size=256, blur_radius: float = 0.0, faces_per_pixel: int = 8, bin_size=None, max_faces_per_bin=None,
                     perspective_correct: bool = False, clip_barycentric_coords: bool = False, cull_backfaces: bool = False,
                     z_clip_value=None, cull_to_frustum: bool = False):
    verts_packed = meshes.verts_packed()
    faces_packed = meshes.faces_packed()
    face_verts = verts_packed[faces_packed]
    counts = [len(f) for f in meshes.faces_list()]
    first = [sum(counts[:i]) for i in range(len(counts))]
    first_t = torch.tensor(first, dtype=torch.int64, device=face_verts.device)
    count_t = torch.tensor(counts, dtype=torch.int64, device=face_verts.device)
    return rasterize_face_verts_ndc(face_verts, first_t, count_t, image_size, blur_radius, faces_per_pixel, perspective_correct,
                                    clip_barycentric_coords, cull_backfaces, z_clip_value, cull_to_frustum)
