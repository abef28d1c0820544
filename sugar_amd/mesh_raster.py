"""Host side of the HIP triangle-mesh z-buffer (C ABI: sgr_rasterize_meshes in include/sugar_raster.h).

`rasterize_face_verts` has the argument meaning of pytorch3d 0.7.4's `_C.rasterize_meshes` (the extension function
`pytorch3d.renderer.mesh.rasterize_meshes` calls after its Python-side clipping) for the faces of one or several meshes:
face_verts[F,3,3] in NDC with view-space z -> (pix_to_face[N,H,W,K], zbuf, bary_coords[N,H,W,K,3], dists).  It is what the
stand-in `pytorch3d.renderer.MeshRasterizer` (sugar_amd/shims/pytorch3d/renderer/mesh) runs on, and through it SuGaR's
level-set sampler with `use_gaussian_depth=False` (sugar_scene/sugar_model.py:1912-1928,1966; coarse_mesh.py:26).

There is no CPU path: CPU tensors raise.  `_backend` exists for the CPU tests only (tests/ install the oracle there to pin the
host logic of the shim and to write fixtures with the reference's own sampler); the product never sets it.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib

MAX_FACES_PER_PIXEL = 16
_backend = None  # tests only: callable(face_verts[F,3,3] cpu tensor, image_size, K, perspective_correct, cull_backfaces) -> 4 tensors [H,W,K(,3)]


def rasterize_face_verts(face_verts: torch.Tensor, mesh_to_face_first_idx, num_faces_per_mesh, image_size, blur_radius: float = 0.0,
                         faces_per_pixel: int = 1, perspective_correct: bool = False, clip_barycentric_coords: bool = False,
                         cull_backfaces: bool = False, want_bary: bool = True, want_dists: bool = True):
    H, W = (int(image_size), int(image_size)) if isinstance(image_size, int) else (int(image_size[0]), int(image_size[1]))
    K = int(faces_per_pixel)
    if blur_radius != 0.0:
        raise NotImplementedError("the HIP mesh rasterizer implements hard rasterization only (blur_radius == 0, what SuGaR passes)")
    if clip_barycentric_coords:
        raise NotImplementedError("clip_barycentric_coords=True is not implemented (pytorch3d's default for blur_radius 0 is False)")
    if K < 1 or K > MAX_FACES_PER_PIXEL:
        raise ValueError(f"faces_per_pixel must be in 1..{MAX_FACES_PER_PIXEL}")
    if face_verts.dim() != 3 or face_verts.shape[1:] != (3, 3):
        raise ValueError("face_verts must have shape (F, 3, 3)")
    first = [int(x) for x in (mesh_to_face_first_idx.tolist() if torch.is_tensor(mesh_to_face_first_idx) else mesh_to_face_first_idx)]
    count = [int(x) for x in (num_faces_per_mesh.tolist() if torch.is_tensor(num_faces_per_mesh) else num_faces_per_mesh)]
    N = len(first)
    dev = face_verts.device
    fv = face_verts.detach()
    if fv.dtype != torch.float32:
        raise RuntimeError("face_verts must be float32")
    fv = fv.contiguous()
    p2f = torch.empty(N, H, W, K, dtype=torch.int64, device=dev)
    zbuf = torch.empty(N, H, W, K, dtype=torch.float32, device=dev)
    bary = torch.empty(N, H, W, K, 3, dtype=torch.float32, device=dev) if want_bary else None
    dists = torch.empty(N, H, W, K, dtype=torch.float32, device=dev) if want_dists else None
    if _backend is not None and not fv.is_cuda:
        for n in range(N):
            r = _backend(fv[first[n]:first[n] + count[n]], (H, W), K, bool(perspective_correct), bool(cull_backfaces))
            p2f[n] = torch.where(r[0] >= 0, r[0] + first[n], r[0])
            zbuf[n] = r[1]
            if bary is not None:
                bary[n] = r[2]
            if dists is not None:
                dists[n] = r[3]
        return p2f, zbuf, bary, dists
    if not fv.is_cuda:
        raise RuntimeError("the HIP mesh rasterizer needs tensors on a ROCm device (got CPU tensors); there is no CPU fallback")
    lib = _lib.load()
    keep = []  # scratch stays referenced until the work is enqueued; the caching allocator keeps it alive for the stream

    def alloc(_user, nbytes):
        t = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        keep.append(t)
        return t.data_ptr()

    cb = _lib.ALLOC_FN(alloc)
    vp = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        for n in range(N):
            F = count[n]
            nbytes = lib.sgr_rasterize_meshes_scratch_bytes(F, W, H)
            scratch = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)
            keep.append(scratch)
            sub = fv[first[n]:first[n] + F]
            rc = lib.sgr_rasterize_meshes(vp(sub) if F else None, F, first[n], W, H, float(blur_radius), K, int(bool(perspective_correct)),
                                          0, int(bool(cull_backfaces)), vp(scratch), scratch.numel(), cb, None, vp(p2f[n]), vp(zbuf[n]),
                                          None if bary is None else vp(bary[n]), None if dists is None else vp(dists[n]), stream)
            if rc < 0:
                raise RuntimeError(f"sgr_rasterize_meshes failed ({rc}): {_lib.last_error()}")
    return p2f, zbuf, bary, dists


def splat_face_verts(points: torch.Tensor, scaling: torch.Tensor, quaternions: torch.Tensor, primitive_verts: torch.Tensor,
                     triangle_scale: float, world_to_view: torch.Tensor, projection: torch.Tensor) -> torch.Tensor:
    """face_verts[2P,3,3] of SuGaR's splat mesh for one camera, straight from the Gaussian buffers (sgr_splat_mesh_face_verts):
    `SuGaR.triangle_vertices` + `SuGaR.splat_mesh(mode='perspective')` + `MeshRasterizer.transform` in one kernel.
    world_to_view / projection: the camera's 4x4 matrices in pytorch3d's row-vector convention ([1,4,4] or [4,4])."""
    if not points.is_cuda:
        raise RuntimeError("splat_face_verts needs tensors on a ROCm device; there is no CPU fallback")
    lib = _lib.load()
    dev = points.device
    f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
    pts, sc, q, prim = f32(points), f32(scaling), f32(quaternions), f32(primitive_verts)
    V, Pm = f32(world_to_view).reshape(-1), f32(projection).reshape(-1)
    P = pts.shape[0]
    if sc.shape != (P, 3) or q.shape != (P, 4) or prim.shape != (4, 3) or V.numel() != 16 or Pm.numel() != 16:
        raise ValueError("splat_face_verts: expected points[P,3], scaling[P,3], quaternions[P,4], primitive_verts[4,3] and two 4x4 matrices")
    out = torch.empty(2 * P, 3, 3, dtype=torch.float32, device=dev)
    vp = lambda t: C.c_void_p(t.data_ptr())
    with torch.cuda.device(dev):
        rc = lib.sgr_splat_mesh_face_verts(P, vp(pts), vp(sc), vp(q), vp(prim), float(triangle_scale), vp(V), vp(Pm), vp(out),
                                           C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc < 0:
        raise RuntimeError(f"sgr_splat_mesh_face_verts failed ({rc}): {_lib.last_error()}")
    return out

