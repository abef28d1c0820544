"""Developer diagnostic (GPU box): torch profiler over one level-set sampling pass of the real reference class, both depth paths."""
import os, sys, types, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sugar_amd import synthetic as syn
from tests import ref_env
from tests.test_gpu_reference_sugar import _training_cameras
dev = torch.device("cuda:0")
P = 1_000_000
scene, cams, bg = syn.make_config("metric", P=P)
W, H = cams[0].image_width, cams[0].image_height
sm = ref_env.import_sugar_model(patch_sugar=True)
nerf = types.SimpleNamespace(device=dev, training_cameras=_training_cameras(cams))
model = sm.SuGaR(nerfmodel=nerf, points=scene.means3D.to(dev), colors=torch.rand(P, 3).to(dev), initialize=True, sh_levels=4, keep_track_of_knn=True, knn_to_track=16)
with torch.no_grad():
    o = scene.opacities.clamp(1e-6, 1 - 1e-6)
    model._scales.copy_(torch.log(scene.scales).to(dev)); model._quaternions.copy_(scene.rotations.to(dev)); model.all_densities.copy_(torch.log(o / (1 - o)).to(dev))
model.primitive_types, model.triangle_scale = 'diamond', 2.
model.point_idx_per_pixel = torch.zeros(2, 2, dtype=torch.int32, device=dev); model.verts_uv = torch.zeros(4 * P, 2, device=dev); model.faces_uv = model.triangles
model._texture_initialized = True
from pytorch3d.renderer import MeshRasterizer, RasterizationSettings
rasterizer = MeshRasterizer(cameras=nerf.training_cameras.p3d_cameras[0], raster_settings=RasterizationSettings(image_size=(H, W), blur_radius=0.0, faces_per_pixel=10, max_faces_per_bin=50_000))
def sample(ugd, cam=0):
    with torch.no_grad():
        return model.compute_level_surface_points_from_camera_fast(cam_idx=cam, rasterizer=rasterizer, surface_levels=[0.1, 0.3, 0.5], n_surface_points=124_000,
            primitive_types='diamond', triangle_scale=2., splat_mesh=True, n_points_in_range=21, range_size=3., n_points_per_pass=2_000_000, density_factor=1.,
            return_pixel_idx=True, return_gaussian_idx=True, return_normals=True, compute_flat_normals=False, use_gaussian_depth=ugd)
from torch.profiler import profile, ProfilerActivity
for ugd in (False, True):
    for _ in range(3): sample(ugd)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for c in range(4): sample(ugd, c)
        torch.cuda.synchronize()
    print("==== use_gaussian_depth =", ugd)
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=60))
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=12, max_name_column_width=60))
