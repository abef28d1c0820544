#!/usr/bin/env python
"""Golden fixture from the reference's OWN methods for the SuGaR-side kernels:
sugar_scene/sugar_model.py is imported from /root/reference, untouched (see make_sugar_callsite.py for the test-only
substitutions: identity `.cuda()`, scipy k-NN, the CPU-oracle-backed rasterizer, empty open3d / plyfile modules), a SuGaR
model is built on the CPU, put into a mid-training state, and these methods of the reference are called:

  SuGaR.sample_points_in_gaussians   :885-928    (seeded; the samples feed the next call)
  SuGaR.get_field_values             :1247-1316  densities, neighbour opacities, beta, sdf + autograd gradients of a scalar
                                                 functional w.r.t. the samples and the model's raw parameters
  SuGaR.get_points_rgb               :839-883    colours + gradients w.r.t. SH coefficients and positions
  SuGaR.get_covariance(return_sqrt)  :729-736    the inverse-scaled rotations the field reads
  SuGaR.compute_level_surface_points_from_camera_fast(use_gaussian_depth=True)   :1848-2083
                                                 per level: pixel indices, front Gaussian, intersection points, normals

The camera is the stand-in FoVPerspectiveCameras of sugar_amd/shims (camera algebra only, built exactly as
sugar_scene/cameras.py:convert_camera_from_gs_to_pytorch3d builds it).  The model state is stored with the outputs so that the
GPU test (tests/test_gpu_sugar_field.py) can rebuild a stand-in object with the same attributes and drive the patched
methods of sugar_amd/sugar_patch.py with the HIP kernels underneath; the CPU test (tests/test_sugar_patch.py) re-runs `run()`
against the committed file and checks the patched methods' host logic against the reference's originals.

    python tests/golden/make_sugar_field.py      -> tests/golden/sugar_field.npz
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import make_sugar_callsite as mk  # noqa: E402

W, H, P, N_SAMPLES = 120, 88, 2500, 6000
LEVELS = [0.1, 0.3, 0.5]


def p3d_cameras_like_the_reference(cams):
    """sugar_scene/cameras.py:262-326 (convert_camera_from_gs_to_pytorch3d) for the synthetic look-at cameras"""
    from pytorch3d.renderer import FoVPerspectiveCameras
    from pytorch3d.renderer.cameras import _get_sfm_calibration_matrix
    Rs, Ts = [], []
    for c in cams:
        w2c = c.viewmatrix.t().double()
        flip = torch.tensor([-1.0, -1.0, 1.0], dtype=torch.float64)   # COLMAP frame -> pytorch3d frame (x left, y up)
        Rs.append((w2c[:3, :3].t() * flip).float())
        Ts.append((w2c[:3, 3] * flip).float())
    Wd, Hd = cams[0].image_width, cams[0].image_height
    scale = min(Wd, Hd) / 2.0
    fx, fy = Wd / (2 * cams[0].tanfovx), Hd / (2 * cams[0].tanfovy)
    K = _get_sfm_calibration_matrix(1, "cpu", torch.tensor([[fx / scale, fy / scale]]), torch.zeros(1, 2)).expand(len(cams), -1, -1)
    return FoVPerspectiveCameras(R=torch.stack(Rs), T=torch.stack(Ts), K=K, znear=0.0001)


class Cameras(mk._Cameras):
    def __init__(self, cams):
        super().__init__(cams)
        self.p3d_cameras = p3d_cameras_like_the_reference(cams)


def build_model(sm, seed=77):
    """the reference SuGaR model on the CPU in a mid-training state (also used by tests/test_sugar_patch.py)"""
    from sugar_amd import synthetic as syn
    cams = syn.orbit_cameras(W, H)
    nerf = types.SimpleNamespace(device=torch.device("cpu"), training_cameras=Cameras(cams))
    g = torch.Generator().manual_seed(seed)
    # a surface-like cloud: points on a blob with small noise, so that rays hit a density surface
    d = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1)
    pts = d * (0.55 + 0.12 * torch.sin(3 * d[:, :1]) * torch.cos(2 * d[:, 1:2])) + 0.01 * torch.randn(P, 3, generator=g)
    cols = torch.rand(P, 3, generator=g)
    model = sm.SuGaR(nerfmodel=nerf, points=pts, colors=cols, initialize=True, sh_levels=4, keep_track_of_knn=True,
                     knn_to_track=16)
    with torch.no_grad():
        model._scales += 0.3 * torch.randn(P, 3, generator=g) + 0.4
        model._quaternions += 0.8 * torch.randn(P, 4, generator=g)
        model.all_densities += 1.5 * torch.randn(P, 1, generator=g) + 2.0
        model._sh_coordinates_rest += 0.15 * torch.randn(P, 15, 3, generator=g)
    return model, cams


def run():
    sm = mk._import_reference_model()
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    sm.knn_points = mk._scipy_knn_points
    from tests.oracle_rasterizer import GaussianRasterizer as OracleRasterizer
    sm.GaussianRasterizer = OracleRasterizer
    try:
        model, cams = build_model(sm)
        out = {"W": np.int32(W), "H": np.int32(H)}
        for name in ("_points", "_scales", "_quaternions", "all_densities", "_sh_coordinates_dc", "_sh_coordinates_rest"):
            out["state" + name] = getattr(model, name).detach().numpy().copy()
        out["state_knn_idx"] = model.knn_idx.numpy().copy()
        g = torch.Generator().manual_seed(5)

        # ---- sample_points_in_gaussians (:885-928) + get_field_values (:1247-1316)
        torch.manual_seed(11)
        mask = torch.rand(P, generator=g) < 0.8
        samples, sample_idx = model.sample_points_in_gaussians(N_SAMPLES, sampling_scale_factor=1.5, mask=mask,
                                                               probabilities_proportional_to_volume=True)
        out["field_mask"] = mask.numpy()
        out["field_x"] = samples.detach().numpy().copy()
        out["field_gaussian_idx"] = sample_idx.numpy().copy()
        x = samples.detach().clone().requires_grad_(True)
        model.zero_grad(set_to_none=True)
        # (forward values with densities on both sides of 1 -- the `density >= 1 -> density / (density.detach() + 1e-12)` branch,
        # :1277-1279 -- first; the differentiated call below keeps every density below 1: at density >= 1 the reference's sdf is
        # sqrt(-2 log 1) = sqrt(0), whose autograd derivative is 0 / 0 -- NaN gradients in the reference itself)
        with torch.no_grad():
            hi = model.get_field_values(samples, sample_idx, return_sdf=True, density_threshold=1., density_factor=1.3,
                                        return_closest_gaussian_opacities=True, return_beta=True)
        assert float((hi["density"] >= 1).float().mean()) > 0.05
        for k in ("density", "sdf", "beta", "closest_gaussian_opacities"):
            out["field_hi_" + k] = hi[k].numpy().copy()
        fields = model.get_field_values(x, sample_idx, return_sdf=True, density_threshold=1., density_factor=0.2,
                                        return_sdf_grad=False, return_closest_gaussian_opacities=True, return_beta=True)
        assert float(fields["density"].detach().max()) < 0.98
        w_d = torch.randn(N_SAMPLES, generator=g); w_s = torch.randn(N_SAMPLES, generator=g)
        w_o = torch.randn(N_SAMPLES, 16, generator=g); w_b = torch.randn(N_SAMPLES, generator=g)
        for k, v in (("w_density", w_d), ("w_sdf", w_s), ("w_opacities", w_o), ("w_beta", w_b)):
            out["field_" + k] = v.numpy()
        functional = ((fields["density"] * w_d).sum() + (fields["sdf"] * w_s).sum()
                      + (fields["closest_gaussian_opacities"] * w_o).sum() + (fields["beta"] * w_b).sum())
        functional.backward()
        for k in ("density", "sdf", "beta", "closest_gaussian_opacities"):
            out["field_out_" + k] = fields[k].detach().numpy().copy()
        out["field_grad_x"] = x.grad.numpy().copy()
        for name in ("_points", "_scales", "_quaternions", "all_densities"):
            out["field_grad" + name] = getattr(model, name).grad.detach().numpy().copy()

        # ---- get_covariance(return_sqrt=True, inverse_scales=True) (:729-736)
        out["inv_scaled_rot"] = model.get_covariance(return_full_matrix=True, return_sqrt=True, inverse_scales=True).detach().numpy().copy()

        # ---- get_points_rgb (:839-883)
        model.zero_grad(set_to_none=True)
        cam_center = cams[2].campos[None]
        rgb = model.get_points_rgb(positions=model.points, camera_centers=cam_center, sh_levels=4)
        w_c = torch.randn(P, 3, generator=g)
        (rgb * w_c).sum().backward()
        out["rgb_camera_center"] = cam_center.numpy(); out["rgb_w"] = w_c.numpy(); out["rgb_out"] = rgb.detach().numpy().copy()
        for name in ("_points", "_sh_coordinates_dc", "_sh_coordinates_rest"):
            out["rgb_grad" + name] = getattr(model, name).grad.detach().numpy().copy()

        # ---- compute_level_surface_points_from_camera_fast(use_gaussian_depth=True) (:1848-2083)
        with torch.no_grad():
            res = model.compute_level_surface_points_from_camera_fast(
                cam_idx=3, rasterizer=None, surface_levels=LEVELS, n_surface_points=-1, primitive_types='diamond',
                triangle_scale=2., n_points_in_range=21, range_size=3., n_points_per_pass=2_000_000, density_factor=1.,
                return_pixel_idx=True, return_gaussian_idx=True, return_normals=True, use_gaussian_depth=True)
        out["ls_cam_idx"] = np.int32(3)
        for lv in LEVELS:
            tag = f"ls_{int(round(lv * 10))}_"
            out[tag + "pixel_idx"] = res[lv]["pixel_idx"].numpy().copy()
            out[tag + "gaussian_idx"] = res[lv]["gaussian_idx"].numpy().copy()
            out[tag + "points"] = res[lv]["intersection_points"].numpy().copy()
            out[tag + "normals"] = res[lv]["normals"].numpy().copy()
        return out
    finally:
        torch.Tensor.cuda = real_cuda


def main():
    out = run()
    path = os.path.join(HERE, "sugar_field.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")
    for k in sorted(out):
        a = np.asarray(out[k])
        print(f"  {k:34s} {str(a.shape):16s} {a.dtype}  mean {float(a.astype(np.float64).mean()):.5g}")


if __name__ == "__main__":
    sys.exit(main())
