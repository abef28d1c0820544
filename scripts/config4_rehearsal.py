"""BASELINE config 4 at world size 1 (the 8-GPU node is not reachable through gpurun): 1M surface-bound (flat) Gaussians @
1920x1080, per view ONE native train step plus ONE level-set sampling pass of the coarse-mesh extractor
(compute_level_surface_points_from_camera_fast on the Gaussian-depth path: depth render, unprojection, k-NN, 124k pixels x 21
samples x 16 neighbours x 3 levels), and the k-NN(16) rebuild over the Gaussians.  Everything runs on this repository's
kernels; the SuGaR object is the attribute-level stand-in of tests/sugar_standin.py (the reference tree does not exist on the
GPU box), driving the SAME patched methods `shims.install(patch_sugar=...)` puts on the reference class.

    python scripts/config4_rehearsal.py > gpurun_out/r03/config4_rehearsal.json
"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from sugar_amd import shims, synthetic as syn
from sugar_amd.knn import knn_points
from sugar_amd.train_step import GaussianParams, NativeTrainer

shims.install()
from tests.sugar_standin import StandInSuGaR
from tests.golden.make_sugar_field import p3d_cameras_like_the_reference

dev = torch.device("cuda:0")
scene, cams, bg = syn.make_config("metric")
P = scene.means3D.shape[0]
W, H = cams[0].image_width, cams[0].image_height
flat = scene.scales.clone()
flat[:, 0] = 1e-6 * 2.0  # thickness 1e-6 x scene extent (sugar_model.py:165-169,438-442): surface-aligned Gaussians
scene = scene._replace(scales=flat)
cams_d = [c._replace(viewmatrix=c.viewmatrix.to(dev), projmatrix=c.projmatrix.to(dev), campos=c.campos.to(dev)) for c in cams]
gts = [torch.rand(3, H, W, generator=torch.Generator().manual_seed(i)).to(dev) for i in range(len(cams))]


def timed(fn, n, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


# ---- the train step on the flat Gaussians
params = GaussianParams(scene, dev)
tr = NativeTrainer(params, bg, W, H)
k = [0]
def step():
    i = k[0] % len(cams); k[0] += 1
    tr.step(cams_d[i], gts[i], cam_key=i)
for _ in range(24):
    step()
tr.synchronize()
ms_step = timed(step, 40, warm=0)
tr.synchronize()

# ---- k-NN rebuild (sugar_model.py:1028-1030) and the sampler on the same Gaussians
pts = params.params["xyz"].detach()
ms_knn = timed(lambda: knn_points(pts[None], pts[None], K=16), 5)
knn_idx = knn_points(pts[None], pts[None], K=16).idx[0]
o = scene.opacities.clamp(1e-6, 1 - 1e-6)
fx = {"state_points": scene.means3D.numpy(), "state_scales": torch.log(scene.scales).numpy(), "state_quaternions": scene.rotations.numpy(),
      "stateall_densities": torch.log(o / (1 - o)).numpy(), "state_sh_coordinates_dc": scene.shs[:, :1].numpy(),
      "state_sh_coordinates_rest": scene.shs[:, 1:].numpy(), "state_knn_idx": knn_idx.cpu().numpy(), "W": W, "H": H}
model = StandInSuGaR(fx, dev, cams, p3d_cameras_like_the_reference(cams).to(dev))
out_counts = {}
def sample(cam_idx=[0]):
    with torch.no_grad():
        r = model.compute_level_surface_points_from_camera_fast(
            cam_idx=cam_idx[0] % len(cams), surface_levels=[0.1, 0.3, 0.5], n_surface_points=124_000, n_points_in_range=21, range_size=3.,
            density_factor=1., return_pixel_idx=True, return_gaussian_idx=True, return_normals=True, use_gaussian_depth=True)
    cam_idx[0] += 1
    out_counts.update({str(lv): int(r[lv]["intersection_points"].shape[0]) for lv in r})
ms_sample = timed(sample, 8)
print(json.dumps({"config": "BASELINE config 4 at world size 1: 1M flat Gaussians @ 1920x1080, per view one native train step + one level-set "
                            "sampling pass (124k pixels x 21 samples x 16 neighbours, 3 levels, Gaussian-depth path)",
                  "ms_train_step": ms_step, "ms_level_set_sampling_per_view": ms_sample, "ms_per_view_total": ms_step + ms_sample,
                  "views_per_sec_one_gpu": 1e3 / (ms_step + ms_sample), "ms_knn16_rebuild_1M": ms_knn,
                  "level_set_points_per_view": out_counts, "num_rendered": tr.last_num_rendered, "forwards_repeated": tr.redone}))
