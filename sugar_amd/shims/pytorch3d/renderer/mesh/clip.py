"""Near-plane clipping of faces before rasterization -- the Python-side step of pytorch3d 0.7.4's `rasterize_meshes`
(`pytorch3d/renderer/mesh/clip.py`: ClipFrustum, clip_faces, convert_clipped_rasterization_to_original_faces), restated from
the published algorithm; pytorch3d is absent here, so this is PARITY-UNPINNED against it.

`MeshRasterizer` clips at z = znear / 2 for perspective cameras (SuGaR: znear = 1e-4, sugar_scene/cameras.py:244,324): a face
entirely behind the plane is dropped, one with two vertices behind is cut to a smaller triangle, one with one vertex behind
to a quadrilateral = two triangles.  The cut is made in view space (x, y of the NDC vertices multiplied back by their depth
when `perspective_correct`), the barycentric coordinates of the new corners w.r.t. the original face are kept so that the
rasterizer's output can be mapped back.  Plain torch on the device the faces live on; the common case -- nothing to clip --
costs one reduction and returns the input."""
from __future__ import annotations

from typing import NamedTuple, Optional

import torch


class ClipFrustum:
    def __init__(self, left=None, right=None, top=None, bottom=None, znear=None, zfar=None, perspective_correct: bool = False,
                 cull: bool = True, z_clip_value=None):
        self.left, self.right, self.top, self.bottom, self.znear, self.zfar = left, right, top, bottom, znear, zfar
        self.perspective_correct, self.cull, self.z_clip_value = perspective_correct, cull, z_clip_value


class ClippedFaces(NamedTuple):
    face_verts: torch.Tensor                                 # (F_clipped, 3, 3)
    mesh_to_face_first_idx: torch.Tensor                     # (N,)
    num_faces_per_mesh: torch.Tensor                         # (N,)
    faces_clipped_to_unclipped_idx: Optional[torch.Tensor] = None   # (F_clipped,) original face of every clipped face
    barycentric_conversion: Optional[torch.Tensor] = None    # (T, 3, 3): columns = the new corners in the original face's barycentrics
    faces_clipped_to_conversion_idx: Optional[torch.Tensor] = None  # (F_clipped,) row of barycentric_conversion or -1
    clipped_faces_neighbor_idx: Optional[torch.Tensor] = None       # (F_clipped,) the other half of a split face or -1


def _culled(face_verts: torch.Tensor, frustum: ClipFrustum) -> torch.Tensor:
    """faces entirely outside one plane of the view frustum (only when frustum.cull)"""
    out = torch.zeros(face_verts.shape[0], dtype=torch.bool, device=face_verts.device)
    if not frustum.cull:
        return out
    for value, axis, below in ((frustum.left, 0, True), (frustum.right, 0, False), (frustum.top, 1, True), (frustum.bottom, 1, False),
                               (frustum.znear, 2, True), (frustum.zfar, 2, False)):
        if value is None:
            continue
        c = face_verts[:, :, axis]
        out |= ((c < value) if below else (c > value)).sum(1) == 3
    return out


def _intersections(faces: torch.Tensor, p1_ind: torch.Tensor, clip_value: float, perspective_correct: bool):
    """p1 is alone on its side of the plane; p4 / p5 = where the edges p1-p2 / p1-p3 cross z = clip_value.  Returns the five
    points and their barycentric coordinates in the original face."""
    T = faces.shape[0]
    p2_ind, p3_ind = (p1_ind + 1) % 3, (p1_ind + 2) % 3
    pick = lambda ind: faces.gather(1, ind[:, None, None].expand(-1, -1, 3)).squeeze(1)
    p1, p2, p3 = pick(p1_ind), pick(p2_ind), pick(p3_ind)

    def cut(pa, pb):
        w = (pa[:, 2] - clip_value) / (pa[:, 2] - pb[:, 2])
        p = pa * (1 - w[:, None]) + pb * w[:, None]
        if perspective_correct:  # interpolate x, y in view space, project back at the plane's depth
            a, b = pa[:, :2] * pa[:, 2:3], pb[:, :2] * pb[:, 2:3]
            p = torch.cat([(a * (1 - w[:, None]) + b * w[:, None]) / clip_value, p[:, 2:3]], dim=1)
        return p, w

    p4, w2 = cut(p1, p2)
    p5, w3 = cut(p1, p3)
    w3 = w3.detach()
    rows = torch.arange(T, device=faces.device)
    bary = [torch.zeros(T, 3, dtype=faces.dtype, device=faces.device) for _ in range(5)]
    bary[0][rows, p1_ind] = 1
    bary[1][rows, p2_ind] = 1
    bary[2][rows, p3_ind] = 1
    bary[3][rows, p1_ind] = 1 - w2
    bary[3][rows, p2_ind] = w2
    bary[4][rows, p1_ind] = 1 - w3
    bary[4][rows, p3_ind] = w3
    return (p1, p2, p3, p4, p5), bary


def clip_faces(face_verts_unclipped: torch.Tensor, mesh_to_face_first_idx: torch.Tensor, num_faces_per_mesh: torch.Tensor,
               frustum: ClipFrustum) -> ClippedFaces:
    F = face_verts_unclipped.shape[0]
    device = face_verts_unclipped.device
    culled = _culled(face_verts_unclipped, frustum)
    z_clip = frustum.z_clip_value
    if z_clip is not None:
        behind = face_verts_unclipped[:, :, 2] < z_clip
        n_behind = behind.sum(1)
    else:
        behind = None
        n_behind = torch.zeros(F, dtype=torch.int64, device=device)
    if int(n_behind.sum().item()) == 0 and int(culled.sum().item()) == 0:  # nothing to do (the usual case)
        return ClippedFaces(face_verts_unclipped, mesh_to_face_first_idx, num_faces_per_mesh)

    keep = ~culled
    case1 = (n_behind == 0) & keep                # untouched
    case2 = (n_behind == 3) | culled              # dropped
    case3 = (n_behind == 2) & keep                # -> one smaller triangle
    case4 = (n_behind == 1) & keep                # -> two triangles
    idx1, idx3, idx4 = (c.nonzero(as_tuple=True)[0] for c in (case1, case3, case4))
    # position of every original face in the clipped array: case 2 faces vanish, case 4 faces take two consecutive slots
    slots = 1 + case4.long() - case2.long()
    to_clipped = slots.cumsum(0) - slots
    F_clipped = int(slots.sum().item())
    # per mesh: first face and count after clipping
    first = mesh_to_face_first_idx.long()
    end = first + num_faces_per_mesh.long()
    csum = torch.cat([slots.new_zeros(1), slots.cumsum(0)])
    first_clipped = csum[first]
    count_clipped = csum[end] - csum[first]

    out = torch.zeros(F_clipped, 3, 3, dtype=face_verts_unclipped.dtype, device=device)
    to_unclipped = torch.full((F_clipped,), -1, dtype=torch.int64, device=device)
    to_conv = torch.full((F_clipped,), -1, dtype=torch.int64, device=device)
    neighbor = torch.full((F_clipped,), -1, dtype=torch.int64, device=device)
    T3, T4 = idx3.numel(), idx4.numel()
    conv = torch.zeros(T3 + 2 * T4, 3, 3, dtype=face_verts_unclipped.dtype, device=device)

    out[to_clipped[idx1]] = face_verts_unclipped[idx1]
    to_unclipped[to_clipped[idx1]] = idx1
    if T3:
        f3 = face_verts_unclipped[idx3]
        p1_ind = torch.where(~behind[idx3])[1]     # the one vertex in front
        (p1, _, _, p4, p5), (b1, _, _, b4, b5) = _intersections(f3, p1_ind, z_clip, frustum.perspective_correct)
        c3 = to_clipped[idx3]
        out[c3] = torch.stack((p4, p5, p1), 1)
        conv[:T3] = torch.stack((b4, b5, b1), 2)
        to_unclipped[c3] = idx3
        to_conv[c3] = torch.arange(T3, device=device)
    if T4:
        f4 = face_verts_unclipped[idx4]
        p1_ind = torch.where(behind[idx4])[1]      # the one vertex behind
        (_, p2, p3, p4, p5), (_, b2, b3, b4, b5) = _intersections(f4, p1_ind, z_clip, frustum.perspective_correct)
        c4 = to_clipped[idx4]
        out[c4] = torch.stack((p4, p2, p5), 1)
        out[c4 + 1] = torch.stack((p5, p2, p3), 1)
        conv[T3:T3 + T4] = torch.stack((b4, b2, b5), 2)
        conv[T3 + T4:] = torch.stack((b5, b2, b3), 2)
        to_unclipped[c4] = idx4
        to_unclipped[c4 + 1] = idx4
        to_conv[c4] = torch.arange(T3, T3 + T4, device=device)
        to_conv[c4 + 1] = torch.arange(T3 + T4, T3 + 2 * T4, device=device)
        neighbor[c4] = c4 + 1
        neighbor[c4 + 1] = c4
    return ClippedFaces(out, first_clipped, count_clipped, to_unclipped, conv, to_conv, neighbor)


def convert_clipped_rasterization_to_original_faces(pix_to_face_clipped: torch.Tensor, bary_coords_clipped: torch.Tensor,
                                                    clipped_faces: ClippedFaces):
    """pix_to_face back to the indices of the faces as they were given, barycentric coordinates back to the original corners"""
    to_unclipped = clipped_faces.faces_clipped_to_unclipped_idx
    if to_unclipped is None:
        return pix_to_face_clipped, bary_coords_clipped
    empty = pix_to_face_clipped == -1
    safe = pix_to_face_clipped.clamp_min(0)
    pix_to_face = torch.where(empty, pix_to_face_clipped, to_unclipped[safe])
    bary = bary_coords_clipped
    conv = clipped_faces.barycentric_conversion
    if conv is not None and conv.shape[0] > 0 and bary is not None:
        ci = torch.where(empty, torch.full_like(safe, -1), clipped_faces.faces_clipped_to_conversion_idx[safe])
        m = ci != -1
        if bool(m.any()):
            bary = bary.clone()
            bary[m] = conv[ci[m]].bmm(bary[m].unsqueeze(-1)).squeeze(-1)
    return pix_to_face, bary
