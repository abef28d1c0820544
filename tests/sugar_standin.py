"""A stand-in for the reference's SuGaR object on the GPU box (where /root/reference does not exist): the attributes and
helper methods the patched methods of sugar_amd/sugar_patch.py read, rebuilt from the model state stored in
tests/golden/sugar_field.npz (written by the reference model itself, tests/golden/make_sugar_field.py).  Every helper cites
the reference lines it restates; the methods under test are the patched ones, bound here exactly as
`sugar_patch.install()` binds them to the real class."""
import functools
import types

import numpy as np
import torch

from sugar_amd import sugar_patch


class StandInSuGaR:
    def __init__(self, fx, device, cams, p3d_cameras):
        t = lambda k: torch.as_tensor(fx[k]).to(device)
        self._points = t("state_points").requires_grad_(True)
        self._scales = t("state_scales").requires_grad_(True)
        self._quaternions = t("state_quaternions").requires_grad_(True)
        self.all_densities = t("stateall_densities").requires_grad_(True)
        self._sh_coordinates_dc = t("state_sh_coordinates_dc").requires_grad_(True)
        self._sh_coordinates_rest = t("state_sh_coordinates_rest").requires_grad_(True)
        self.knn_idx = t("state_knn_idx")
        self.knn_to_track = 16
        self.image_width, self.image_height = int(fx["W"]), int(fx["H"])
        self.device = torch.device(device)
        self.cams = cams
        self.nerfmodel = types.SimpleNamespace(training_cameras=types.SimpleNamespace(p3d_cameras=p3d_cameras), device=self.device)
        self.beta_mode = "average"
        self.primitive_types, self.triangle_scale = "diamond", 2.0

    # ---- sugar_model.py:384-479 for a model that is not bound to a mesh
    points = property(lambda self: self._points)
    scaling = property(lambda self: torch.exp(self._scales))                                   # :20, :416-418
    quaternions = property(lambda self: torch.nn.functional.normalize(self._quaternions, dim=-1))  # :445-479
    strengths = property(lambda self: torch.sigmoid(self.all_densities.view(-1, 1)))           # :401-405
    sh_coordinates = property(lambda self: torch.cat([self._sh_coordinates_dc, self._sh_coordinates_rest], dim=1))  # :408-409
    n_points = property(lambda self: len(self._points))

    def parameters(self):
        return [self._points, self._scales, self._quaternions, self.all_densities, self._sh_coordinates_dc, self._sh_coordinates_rest]

    def zero_grad(self):
        for p in self.parameters():
            p.grad = None

    def get_beta(self, x, closest_gaussians_idx=None, closest_gaussians_opacities=None, densities=None, opacity_min_clamp=1e-32):
        return self.scaling.min(dim=-1)[0][closest_gaussians_idx].mean(dim=1)                  # :1192-1195 ('average')

    def get_gaussians_closest_to_samples(self, x, n_closest_gaussian=None):                   # :1338-1346
        from sugar_amd.knn import knn_points
        K = self.knn_to_track if n_closest_gaussian is None else n_closest_gaussian
        return knn_points(x[None], self.points[None], K=K).idx[0]

    def render_image_gaussian_rasterizer(self, camera_indices=0, bg_color=None, sh_deg=0, compute_covariance_in_rasterizer=True,
                                         return_2d_radii=False, use_same_scale_in_all_directions=False, point_colors=None, **_):
        """the boundary call of sugar_model.py:2169-2281 for precomputed colours (what the level-set sampler asks for)"""
        from sugar_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
        cam = self.cams[camera_indices]
        dev = self.device
        st = GaussianRasterizationSettings(image_height=self.image_height, image_width=self.image_width, tanfovx=cam.tanfovx,
                                           tanfovy=cam.tanfovy, bg=bg_color, scale_modifier=1., viewmatrix=cam.viewmatrix.to(dev),
                                           projmatrix=cam.projmatrix.to(dev), sh_degree=sh_deg, campos=cam.campos.to(dev),
                                           prefiltered=False, debug=False)
        means2D = torch.zeros(self.n_points, 3, device=dev, requires_grad=True)
        img, _ = GaussianRasterizer(st)(means3D=self.points, means2D=means2D, shs=None, colors_precomp=point_colors,
                                        opacities=self.strengths.view(-1, 1), scales=self.scaling, rotations=self.quaternions,
                                        cov3D_precomp=None)
        return img.transpose(0, 1).transpose(1, 2)

    def _reference_get_covariance(self, return_full_matrix=False, return_sqrt=False, inverse_scales=False):   # :729-736
        from pytorch3d.transforms import quaternion_to_matrix
        scaling = self.scaling
        if inverse_scales:
            scaling = 1. / scaling.clamp(min=1e-8)
        scaled_rotation = quaternion_to_matrix(self.quaternions) * scaling[:, None]
        if return_sqrt:
            return scaled_rotation
        raise NotImplementedError

    # ---- the methods under test, bound like sugar_patch.install() binds them
    get_covariance = functools.partialmethod(sugar_patch.get_covariance, _orig=_reference_get_covariance)
    get_points_rgb = functools.partialmethod(sugar_patch.get_points_rgb, _orig=None)
    get_field_values = functools.partialmethod(sugar_patch.get_field_values, _orig=None)
    compute_level_surface_points_from_camera_fast = functools.partialmethod(
        sugar_patch.compute_level_surface_points_from_camera_fast, _orig=None)
