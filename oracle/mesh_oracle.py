"""numpy driver for the mesh z-buffer oracle (oracle/mesh_rasterizer.c) -- TEST INFRASTRUCTURE, NOT PRODUCT.

`rasterize_meshes_naive` has the argument meaning of pytorch3d's `_C.rasterize_meshes` for one mesh (face_verts in NDC with
view-space z) and returns (pix_to_face, zbuf, bary_coords, dists) shaped [H,W,K] / [H,W,K,3].  PARITY UNPINNED against
pytorch3d (absent from this image and from /root/reference); see the header of mesh_rasterizer.c."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import cpu_oracle


def pix_to_ndc(i: int, S1: int, S2: int) -> float:
    f = cpu_oracle.lib().orc_mesh_pix_to_ndc
    f.restype = C.c_float
    return float(f(C.c_int(i), C.c_int(S1), C.c_int(S2)))


def rasterize_meshes_naive(face_verts, image_size, blur_radius=0.0, faces_per_pixel=1, perspective_correct=True,
                           clip_barycentric_coords=False, cull_backfaces=False, clipped_faces_neighbor_idx=None):
    fv = np.ascontiguousarray(np.asarray(face_verts, dtype=np.float32).reshape(-1, 3, 3))
    F = fv.shape[0]
    H, W = (image_size, image_size) if isinstance(image_size, int) else image_size
    K = int(faces_per_pixel)
    nb = None
    if clipped_faces_neighbor_idx is not None:
        nb = np.ascontiguousarray(np.asarray(clipped_faces_neighbor_idx, dtype=np.int64))
        assert nb.shape == (F,)
    p2f = np.empty((H, W, K), dtype=np.int64)
    zbuf = np.empty((H, W, K), dtype=np.float32)
    bary = np.empty((H, W, K, 3), dtype=np.float32)
    dists = np.empty((H, W, K), dtype=np.float32)
    vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    rc = cpu_oracle.lib().orc_rasterize_meshes_naive(vp(fv), C.c_int64(F), vp(nb), C.c_int(H), C.c_int(W), C.c_float(blur_radius),
                                                     C.c_int(K), C.c_int(int(perspective_correct)), C.c_int(int(clip_barycentric_coords)),
                                                     C.c_int(int(cull_backfaces)), vp(p2f), vp(zbuf), vp(bary), vp(dists))
    if rc != 0:
        raise ValueError("orc_rasterize_meshes_naive: bad arguments")
    return p2f, zbuf, bary, dists
