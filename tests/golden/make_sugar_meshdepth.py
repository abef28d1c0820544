#!/usr/bin/env python
"""Golden fixture for the level-set sampler's DEFAULT path -- `use_gaussian_depth=False`, what sugar_extractors/coarse_mesh.py:26
hard-codes -- written by the reference's OWN method on the CPU:

  SuGaR.compute_level_surface_points_from_camera_fast(...)  sugar_scene/sugar_model.py:1848-2083

called with the arguments of coarse_mesh.py:271-287 on the model of make_sugar_field.py (same state).  The method builds the
splat mesh (:695-727), rasterizes it with `pytorch3d.renderer.MeshRasterizer` (:1927) -- here the stand-in of sugar_amd.shims on
the CPU ORACLE backend (oracle/mesh_rasterizer.c, test infrastructure) -- and takes depth and front Gaussian from the fragments
(:1928, :1966).  Everything else is the reference's tensor code.  Two calls: every pixel (n_surface_points = -1), and a seeded
random subset as the extractor asks for (`n_surface_points = 2 * n_pts_per_frame`; the permutation is drawn on the CPU, :1955, so
the GPU run draws the same one).  The fragments of the splat mesh (nearest face and its depth) are stored as well.

The GPU test (tests/test_gpu_reference_sugar.py) runs the SAME unmodified method of the reference class on the MI355X with the HIP
kernels underneath (mesh z-buffer, k-NN, level-set kernel) and compares.

    python tests/golden/make_sugar_meshdepth.py      -> tests/golden/sugar_meshdepth.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import make_sugar_callsite as mk  # noqa: E402
import make_sugar_field as mf  # noqa: E402

LEVELS = [0.1, 0.3, 0.5]
CAM_IDX = 3
N_SUBSET = 700
SEED = 123


def sampler_kwargs(n_surface_points):
    """coarse_mesh.py:271-287"""
    return dict(cam_idx=CAM_IDX, surface_levels=LEVELS, n_surface_points=n_surface_points, primitive_types='diamond', triangle_scale=2.,
                splat_mesh=True, n_points_in_range=21, range_size=3., n_points_per_pass=2_000_000, density_factor=1.,
                return_pixel_idx=True, return_gaussian_idx=True, return_normals=True, compute_flat_normals=False,
                use_gaussian_depth=False)


def make_rasterizer(model):
    """coarse_mesh.py:216-225"""
    from pytorch3d.renderer import MeshRasterizer, RasterizationSettings
    settings = RasterizationSettings(image_size=(model.image_height, model.image_width), blur_radius=0.0, faces_per_pixel=10,
                                     max_faces_per_bin=50_000)
    return MeshRasterizer(cameras=model.nerfmodel.training_cameras.p3d_cameras[0], raster_settings=settings)


def run():
    from tests.mesh_backend import oracle_mesh_rasterizer
    sm = mk._import_reference_model()
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    sm.knn_points = mk._scipy_knn_points
    from tests.oracle_rasterizer import GaussianRasterizer as OracleRasterizer
    sm.GaussianRasterizer = OracleRasterizer
    try:
        model, cams = mf.build_model(sm)
        out = {"cam_idx": np.int32(CAM_IDX), "n_subset": np.int32(N_SUBSET), "seed": np.int32(SEED)}
        with torch.no_grad(), oracle_mesh_rasterizer():
            model.primitive_types = 'diamond'          # coarse_mesh.py:207-210
            model.triangle_scale = 2.
            model.update_texture_features()
            rasterizer = make_rasterizer(model)
            cam = model.nerfmodel.training_cameras.p3d_cameras[CAM_IDX]
            mesh = model.splat_mesh(cam)
            out["splat_verts"] = mesh.verts_list()[0].numpy().copy()
            fr = rasterizer(mesh, cameras=cam)
            out["frag_pix_to_face"] = fr.pix_to_face[0].numpy().astype(np.int32)
            out["frag_zbuf"] = fr.zbuf[0].numpy().copy()
            for tag, n in (("all", -1), ("sub", N_SUBSET)):
                torch.manual_seed(SEED)
                res = model.compute_level_surface_points_from_camera_fast(rasterizer=rasterizer, **sampler_kwargs(n))
                for lv in LEVELS:
                    t = f"{tag}_{int(round(lv * 10))}_"
                    out[t + "pixel_idx"] = res[lv]["pixel_idx"].numpy().copy()
                    out[t + "gaussian_idx"] = res[lv]["gaussian_idx"].numpy().copy()
                    out[t + "points"] = res[lv]["intersection_points"].numpy().copy()
                    out[t + "normals"] = res[lv]["normals"].numpy().copy()
        return out
    finally:
        torch.Tensor.cuda = real_cuda


def main():
    out = run()
    path = os.path.join(HERE, "sugar_meshdepth.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")
    for k in sorted(out):
        a = np.asarray(out[k])
        print(f"  {k:24s} {str(a.shape):16s} {a.dtype}  mean {float(a.astype(np.float64).mean()):.5g}")


if __name__ == "__main__":
    sys.exit(main())
