"""BASELINE config 4 at world size 1 with THE REFERENCE'S OWN SuGaR CLASS on the GPU (round 4): 1M Gaussians @ 1920x1080, per view
one level-set sampling pass of the coarse-mesh extractor as sugar_extractors/coarse_mesh.py:271-287 calls it --
`use_gaussian_depth=False`: splat mesh (2M triangles) -> MeshRasterizer (HIP z-buffer, faces_per_pixel 10) -> depth + front
Gaussian -> 124k pixels x 21 samples x 16 neighbours x 3 levels -- next to the Gaussian-depth path and one native train step.
The reference's Python comes from /root/reference or the staged oracle/_ref/pysrc; `shims.install(patch_sugar=True)` routes the
sampler to the fused kernels.  The UV atlas the reference builds once per extraction with a 1M-iteration Python loop
(`update_texture_features`, sugar_model.py:752-765, 2407-2461) is replaced by a 2x2 placeholder: rasterization never reads it.

    python scripts/config4_rehearsal_r4.py > gpurun_out/r4/config4_rehearsal.json
"""
import json, math, os, sys, time, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sugar_amd import shims, synthetic as syn
from sugar_amd.train_step import GaussianParams, NativeTrainer
from tests import ref_env
from tests.test_gpu_reference_sugar import _training_cameras

dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
scene, cams, bg = syn.make_config("metric", P=P)
W, H = cams[0].image_width, cams[0].image_height
sm = ref_env.import_sugar_model(patch_sugar=True)


def timed(fn, n, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


t0 = time.perf_counter()
nerf = types.SimpleNamespace(device=dev, training_cameras=_training_cameras(cams))
model = sm.SuGaR(nerfmodel=nerf, points=scene.means3D.to(dev), colors=torch.rand(P, 3).to(dev), initialize=True, sh_levels=4,
                 keep_track_of_knn=True, knn_to_track=16)
with torch.no_grad():
    o = scene.opacities.clamp(1e-6, 1 - 1e-6)
    model._scales.copy_(torch.log(scene.scales).to(dev)); model._quaternions.copy_(scene.rotations.to(dev))
    model.all_densities.copy_(torch.log(o / (1 - o)).to(dev))
    model._sh_coordinates_dc.copy_(scene.shs[:, :1].to(dev)); model._sh_coordinates_rest.copy_(scene.shs[:, 1:].to(dev))
torch.cuda.synchronize()
t_build = time.perf_counter() - t0
model.primitive_types, model.triangle_scale = 'diamond', 2.
# placeholder atlas (see the module docstring)
model.point_idx_per_pixel = torch.zeros(2, 2, dtype=torch.int32, device=dev)
model.verts_uv = torch.zeros(4 * P, 2, device=dev); model.faces_uv = model.triangles
model._texture_initialized = True
model.texture_size = 2
from pytorch3d.renderer import MeshRasterizer, RasterizationSettings
rasterizer = MeshRasterizer(cameras=nerf.training_cameras.p3d_cameras[0],
                            raster_settings=RasterizationSettings(image_size=(H, W), blur_radius=0.0, faces_per_pixel=10, max_faces_per_bin=50_000))
out = {"P": P, "W": W, "H": H, "model_build_s": t_build}
counts = {}
k = [0]


def sample(use_gaussian_depth, n=124_000, n_pass=2_000_000):
    with torch.no_grad():
        r = model.compute_level_surface_points_from_camera_fast(
            cam_idx=k[0] % len(cams), rasterizer=rasterizer, surface_levels=[0.1, 0.3, 0.5], n_surface_points=n, primitive_types='diamond',
            triangle_scale=2., splat_mesh=True, n_points_in_range=21, range_size=3., n_points_per_pass=n_pass, density_factor=1.,
            return_pixel_idx=True, return_gaussian_idx=True, return_normals=True, compute_flat_normals=False,
            use_gaussian_depth=use_gaussian_depth)
    k[0] += 1
    counts[str(use_gaussian_depth)] = {str(lv): int(r[lv]["intersection_points"].shape[0]) for lv in r}


print("leg: splat mesh + rasterizer", file=sys.stderr, flush=True)
with torch.no_grad():
    cam0 = nerf.training_cameras.p3d_cameras[0]
    out["ms_splat_mesh_reference_method"] = timed(lambda: model.splat_mesh(cam0), 5)
    mesh = model.splat_mesh(cam0)
    out["ms_mesh_rasterizer_forward"] = timed(lambda: rasterizer(mesh, cameras=cam0), 5)
    fr = rasterizer(mesh, cameras=cam0)
    out["mesh_pixels_covered"] = float((fr.pix_to_face[0, ..., 0] >= 0).float().mean())
    del fr, mesh
if not os.environ.get("ONLY_UNTOUCHED"):
    print("leg: patched mesh depth", file=sys.stderr, flush=True)
    out["ms_sampling_per_view_mesh_depth"] = timed(lambda: sample(False), 8)
    print("leg: patched gaussian depth", file=sys.stderr, flush=True)
    out["ms_sampling_per_view_gaussian_depth"] = timed(lambda: sample(True), 8)
out["level_set_points_per_view"] = counts
# the untouched reference method on the same inputs (its level sets are tensor code in passes of 2M samples)
from sugar_amd import sugar_patch
sugar_patch.uninstall(sm)
print("leg: untouched reference method", file=sys.stderr, flush=True)
try:
    # (n_points_per_pass 100k instead of the extractor's 2M: at 2M samples x 16 neighbours the reference's batched 3x3 matmul,
    # sugar_model.py:2000, takes a GPU memory fault inside the BLAS library of this ROCm stack -- 32M matrices in one batch)
    out["ms_sampling_per_view_mesh_depth_reference_tensor_code"] = timed(lambda: sample(False, n_pass=100_000), 3, warm=1)
    out["reference_tensor_code_note"] = "n_points_per_pass=100000 (2000000, the extractor's value, faults in the batched matmul of sugar_model.py:2000 on this ROCm stack)"
except Exception as e:  # (memory)
    out["ms_sampling_per_view_mesh_depth_reference_tensor_code"] = repr(e)
print(json.dumps(out))
