"""Developer diagnostic (GPU box): where the time of one level-set sampling pass goes (config 4 shape)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from sugar_amd import shims, synthetic as syn
from sugar_amd.knn import knn_points
shims.install()
from tests.sugar_standin import StandInSuGaR
from tests.golden.make_sugar_field import p3d_cameras_like_the_reference

dev = torch.device("cuda:0")
scene, cams, bg = syn.make_config("metric")
flat = scene.scales.clone(); flat[:, 0] = 2e-6
scene = scene._replace(scales=flat)
W, H = cams[0].image_width, cams[0].image_height
pts = scene.means3D.to(dev)
knn_idx = knn_points(pts[None], pts[None], K=16).idx[0]
o = scene.opacities.clamp(1e-6, 1 - 1e-6)
fx = {"state_points": scene.means3D.numpy(), "state_scales": torch.log(scene.scales).numpy(), "state_quaternions": scene.rotations.numpy(),
      "stateall_densities": torch.log(o / (1 - o)).numpy(), "state_sh_coordinates_dc": scene.shs[:, :1].numpy(),
      "state_sh_coordinates_rest": scene.shs[:, 1:].numpy(), "state_knn_idx": knn_idx.cpu().numpy(), "W": W, "H": H}
model = StandInSuGaR(fx, dev, cams, p3d_cameras_like_the_reference(cams).to(dev))


def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n, r


p3 = model.nerfmodel.training_cameras.p3d_cameras[0]
with torch.no_grad():
    ms, depth_pts = t(lambda: p3.get_world_to_view_transform().transform_points(model.points)[..., 2:].expand(-1, 3))
    print("view-space depth of the Gaussians %.2f ms" % ms)
    ms, depth = t(lambda: model.render_image_gaussian_rasterizer(camera_indices=0, bg_color=torch.tensor([-1., -1., -1.], device=dev), sh_deg=0,
                                                                point_colors=depth_pts).contiguous()[..., 0])
    print("depth render %.2f ms" % ms)
    valid = depth >= 0
    print("pixels with depth >= 0: %.3f; depth quantiles of those:" % valid.float().mean().item(),
          torch.quantile(depth[valid][:: 37].float(), torch.tensor([0.001, 0.01, 0.1, 0.5, 0.9], device=dev)).cpu().numpy().round(3))
    m = min(W, H)
    rows = torch.arange(H, device=dev, dtype=torch.float32)[:, None].expand(H, W); cols = torch.arange(W, device=dev, dtype=torch.float32)[None, :].expand(H, W)
    ndc = torch.stack(((W / m - cols / (m - 1) * 2).reshape(-1), (H / m - rows / (m - 1) * 2).reshape(-1), depth.reshape(-1)), dim=-1)[valid.view(-1)]
    ndc = ndc[torch.randperm(ndc.shape[0], device=dev)[:124_000]][None]
    ms, world = t(lambda: p3.unproject_points(ndc, scaled_depth_input=False).view(-1, 3))
    print("unproject %.2f ms" % ms)
    ms, idx = t(lambda: knn_points(world[None], model.points[None], K=16).idx[0], n=3)
    print("k-NN of 124k unprojected pixels against 1M Gaussians %.2f ms" % ms)
    d0 = (world - model.points[idx[:, 0]]).norm(dim=-1)
    print("distance to the nearest Gaussian, quantiles:", torch.quantile(d0, torch.tensor([0.5, 0.9, 0.99, 0.999, 1.0], device=dev)).cpu().numpy().round(4))
    near = d0 < torch.quantile(d0, 0.9)
    ms, _ = t(lambda: knn_points(world[near][None], model.points[None], K=16).idx[0], n=3)
    print("k-NN of the nearest 90 %% of them %.2f ms" % ms)
