"""TEST INFRASTRUCTURE / MEASURED BASELINE, NOT PRODUCT: the reference's vanilla-3DGS optimisation loop run with the reference's
OWN code -- `render()` (gaussian_splatting/gaussian_renderer/__init__.py:18-100), `GaussianModel` and its Adam groups
(scene/gaussian_model.py:43-166), `l1_loss` / `ssim` (utils/loss_utils.py) -- on whatever `diff_gaussian_rasterization` /
`simple_knn` resolve to, i.e. this repository's HIP drop-in packages.  `loop_body` is gaussian_splatting/train.py:69-128 with the
network GUI, logging, checkpointing and (never reached in a short run) densify_and_prune left out; every other statement is the
reference's, in its order.

Used by tests/test_gpu_reference_sugar.py (the reference loop against NativeTrainer on the same scene) and by
`bench.py --reference-loop` (images/s and through-API ms of the unmodified path next to the native number).  The reference's
Python is imported from /root/reference or, on the GPU box, from the git-ignored snapshot oracle/_ref/pysrc."""
from __future__ import annotations

import math
import os
import sys
import types
from random import Random

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_root():
    for p in (os.environ.get("SUGAR_REFERENCE"), "/root/reference", os.path.join(ROOT, "oracle", "_ref", "pysrc")):
        if p and os.path.isdir(os.path.join(p, "gaussian_splatting", "gaussian_renderer")):
            return p
    return None


def import_reference():
    ref = reference_root()
    if ref is None:
        raise ImportError("the reference's Python is neither at /root/reference nor staged in oracle/_ref/pysrc "
                          "(oracle/ref_build/build_ref.sh stages it)")
    for p in (os.path.join(ref, "gaussian_splatting"), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    from sugar_amd import shims
    shims.install()  # `plyfile` (gaussian_model.py:18) is absent from the image: stand-in
    import gaussian_renderer
    from scene.gaussian_model import GaussianModel
    from utils.loss_utils import l1_loss, ssim
    assert os.path.abspath(gaussian_renderer.__file__).startswith(os.path.abspath(ref)), gaussian_renderer.__file__
    import diff_gaussian_rasterization as dgr
    assert os.path.abspath(dgr.__file__).startswith(ROOT), dgr.__file__  # the HIP drop-in, not some other install
    return types.SimpleNamespace(render=gaussian_renderer.render, GaussianModel=GaussianModel, l1_loss=l1_loss, ssim=ssim,
                                 module=gaussian_renderer)


def optimization_params(constant_position_lr: bool = False):
    """arguments/__init__.py:74-92 (OptimizationParams defaults)"""
    o = types.SimpleNamespace(iterations=30_000, position_lr_init=0.00016, position_lr_final=0.0000016, position_lr_delay_mult=0.01,
                              position_lr_max_steps=30_000, feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001,
                              percent_dense=0.01, lambda_dssim=0.2, densification_interval=100, opacity_reset_interval=3000,
                              densify_from_iter=500, densify_until_iter=15_000, densify_grad_threshold=0.0002, random_background=False)
    if constant_position_lr:
        o.position_lr_final = o.position_lr_init
    return o


def pipeline_params():
    """arguments/__init__.py:66-71"""
    return types.SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)


def make_gaussians(ref, scene, device, opt, sh_degree: int = 3):
    """a reference GaussianModel holding a synthetic scene (the state create_from_pcd / load_ply would leave, gaussian_model.py:
    119-141,215-256): raw parameters = inverse activations of the scene's values, all SH bands active"""
    g = ref.GaussianModel(sh_degree)
    g.active_sh_degree = sh_degree
    g.spatial_lr_scale = 1.0
    nn = torch.nn
    o = scene.opacities.clamp(1e-6, 1 - 1e-6)
    g._xyz = nn.Parameter(scene.means3D.to(device).clone().requires_grad_(True))
    g._features_dc = nn.Parameter(scene.shs[:, :1].to(device).contiguous().clone().requires_grad_(True))
    g._features_rest = nn.Parameter(scene.shs[:, 1:].to(device).contiguous().clone().requires_grad_(True))
    g._scaling = nn.Parameter(torch.log(scene.scales).to(device).requires_grad_(True))
    g._rotation = nn.Parameter(scene.rotations.to(device).clone().requires_grad_(True))
    g._opacity = nn.Parameter(torch.log(o / (1 - o)).to(device).requires_grad_(True))
    g.max_radii2D = torch.zeros((g.get_xyz.shape[0]), device=device)
    g.training_setup(opt)
    return g


def make_viewpoint(cam, gt_image, device):
    """what render() and the loop read of a scene.cameras.Camera (gaussian_splatting/scene/cameras.py:17-58)"""
    return types.SimpleNamespace(FoVx=2 * math.atan(cam.tanfovx), FoVy=2 * math.atan(cam.tanfovy), image_height=cam.image_height,
                                 image_width=cam.image_width, world_view_transform=cam.viewmatrix.to(device),
                                 full_proj_transform=cam.projmatrix.to(device), camera_center=cam.campos.to(device),
                                 original_image=gt_image.to(device))


class Loop:
    def __init__(self, ref, gaussians, viewpoints, background, opt=None, pipe=None, seed: int = 0, sequential: bool = False):
        self.ref, self.gaussians, self.viewpoints, self.background = ref, gaussians, list(viewpoints), background
        self.opt = opt or optimization_params()
        self.pipe = pipe or pipeline_params()
        self.iteration = 0
        self.viewpoint_stack = None
        self.rng = Random(seed)
        self.sequential = sequential  # tests: cameras in order instead of the reference's random pops

    def loop_body(self):
        """train.py:69-128, one iteration; returns the loss tensor"""
        ref, gaussians, opt, pipe = self.ref, self.gaussians, self.opt, self.pipe
        self.iteration += 1
        iteration = self.iteration
        gaussians.update_learning_rate(iteration)
        if iteration % 1000 == 0:
            gaussians.oneupSHdegree()
        if not self.viewpoint_stack:
            self.viewpoint_stack = self.viewpoints.copy()
        viewpoint_cam = self.viewpoint_stack.pop(0 if self.sequential else self.rng.randint(0, len(self.viewpoint_stack) - 1))
        bg = torch.rand((3), device="cuda") if opt.random_background else self.background
        render_pkg = ref.render(viewpoint_cam, gaussians, pipe, bg)
        image, viewspace_point_tensor, visibility_filter, radii = (render_pkg["render"], render_pkg["viewspace_points"],
                                                                   render_pkg["visibility_filter"], render_pkg["radii"])
        gt_image = viewpoint_cam.original_image.cuda()
        Ll1 = ref.l1_loss(image, gt_image)
        loss = (1.0 - opt.lambda_dssim) * Ll1 + opt.lambda_dssim * (1.0 - ref.ssim(image, gt_image))
        loss.backward()
        with torch.no_grad():
            if iteration < opt.densify_until_iter:
                gaussians.max_radii2D[visibility_filter] = torch.max(gaussians.max_radii2D[visibility_filter], radii[visibility_filter])
                gaussians.add_densification_stats(viewspace_point_tensor, visibility_filter)
                if iteration > opt.densify_from_iter and iteration % opt.densification_interval == 0:
                    raise RuntimeError("densify_and_prune is outside this loop's scope: keep runs below densify_from_iter")
            if iteration < opt.iterations:
                gaussians.optimizer.step()
                gaussians.optimizer.zero_grad(set_to_none=True)
        return loss
