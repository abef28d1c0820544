"""TEST INFRASTRUCTURE: runs the stand-in `pytorch3d.renderer.MeshRasterizer` on the CPU ORACLE (oracle/mesh_rasterizer.c) instead
of the HIP kernel, for the CPU tests of the shim's host logic and for writing fixtures with the reference's own sampler.  The
product never installs a backend (sugar_amd/mesh_raster.py raises on CPU tensors)."""
import contextlib

import torch


def oracle_backend(face_verts, image_size, K, perspective_correct, cull_backfaces):
    from oracle import mesh_oracle as mo
    r = mo.rasterize_meshes_naive(face_verts.detach().cpu().numpy(), image_size, 0.0, K, perspective_correct, False, cull_backfaces)
    return tuple(torch.from_numpy(a) for a in r)


@contextlib.contextmanager
def oracle_mesh_rasterizer():
    import sugar_amd.mesh_raster as mr
    old = mr._backend
    mr._backend = oracle_backend
    try:
        yield
    finally:
        mr._backend = old
