"""The "train step" the headline metric is quoted on, and its view-sharded data-parallel form.

Restates the vanilla 3DGS optimisation step that SuGaR builds on:
    render()                gaussian_splatting/gaussian_renderer/__init__.py:18-100
    loss = 0.8*L1 + 0.2*(1-SSIM), backward, Adam   gaussian_splatting/train.py:86-128
    l1_loss / ssim          sugar_utils/loss_utils.py:17-63
    parameter activations   gaussian_splatting/scene/gaussian_model.py:33-52 (exp / sigmoid / normalize)
    Adam groups and lrs     gaussian_splatting/scene/gaussian_model.py:152-166, arguments/__init__.py:74-83
Only the rasterizer underneath is this repository's HIP code; loss and Adam are stock PyTorch ops, exactly
as in the reference.

Multi-GPU: the path shards by view.  Every rank holds a full replica of the Gaussians,
renders its own camera, and the parameter gradients meet in ONE flat all-reduce (RCCL over xGMI when the
process group backend is "nccl"; gloo on CPU for the rehearsal tests) before identical Adam steps.
"""
from __future__ import annotations

from math import exp

import torch
import torch.distributed as dist
import torch.nn.functional as F


# ---------------------------------------------------------------- loss (sugar_utils/loss_utils.py:17-63)
def l1_loss(network_output, gt):
    return torch.abs((network_output - gt)).mean()


def _gaussian(window_size, sigma):
    gauss = torch.Tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    return gauss / gauss.sum()


_WINDOWS: dict = {}


def create_window(window_size, channel, like: torch.Tensor):
    key = (window_size, channel, like.device, like.dtype)
    w = _WINDOWS.get(key)
    if w is None:
        _1D = _gaussian(window_size, 1.5).unsqueeze(1)
        _2D = _1D.mm(_1D.t()).float().unsqueeze(0).unsqueeze(0)
        w = _2D.expand(channel, 1, window_size, window_size).contiguous().to(device=like.device, dtype=like.dtype)
        _WINDOWS[key] = w
    return w


def ssim(img1, img2, window_size=11, size_average=True):
    channel = img1.size(-3)
    window = create_window(window_size, channel, img1)
    pad = window_size // 2
    mu1 = F.conv2d(img1, window, padding=pad, groups=channel)
    mu2 = F.conv2d(img2, window, padding=pad, groups=channel)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = F.conv2d(img1 * img1, window, padding=pad, groups=channel) - mu1_sq
    sigma2_sq = F.conv2d(img2 * img2, window, padding=pad, groups=channel) - mu2_sq
    sigma12 = F.conv2d(img1 * img2, window, padding=pad, groups=channel) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    return ssim_map.mean() if size_average else ssim_map.mean(1).mean(1).mean(1)


def photometric_loss(image, gt, lambda_dssim=0.2):
    """gaussian_splatting/train.py:88-90 with stock PyTorch ops (the parity reference of the fused HIP loss)"""
    return (1.0 - lambda_dssim) * l1_loss(image, gt) + lambda_dssim * (1.0 - ssim(image, gt))


def train_loss(image, gt, lambda_dssim=0.2, fused=True):
    """The step's loss: the fused HIP kernels (sugar_amd/csrc/loss.hip) on a ROCm device, stock PyTorch otherwise."""
    if fused and image.is_cuda:
        from .fused_loss import l1_ssim_loss
        return l1_ssim_loss(image, gt, lambda_dssim)
    return photometric_loss(image, gt, lambda_dssim)


# ---------------------------------------------------------------- parameters
class GaussianParams:
    """Raw (pre-activation) 3DGS parameters, 59 floats per Gaussian at SH degree 3, stored as views of ONE flat
    buffer so that the data-parallel gradient exchange is a single all-reduce of one contiguous tensor."""

    NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
    LRS = dict(xyz=0.00016, f_dc=0.0025, f_rest=0.0025 / 20.0, opacity=0.05, scaling=0.005, rotation=0.001)

    def __init__(self, scene, device):
        P = scene.means3D.shape[0]
        M = scene.shs.shape[1]
        self.P, self.M = P, M
        shapes = dict(xyz=(P, 3), f_dc=(P, 1, 3), f_rest=(P, M - 1, 3), opacity=(P, 1), scaling=(P, 3), rotation=(P, 4))
        sizes = {k: int(torch.tensor(v).prod()) for k, v in shapes.items()}
        # keep every view 256-byte aligned inside the flat buffer (vector loads in the kernels)
        offs, off = {}, 0
        for k in self.NAMES:
            offs[k] = off
            off += (sizes[k] + 63) // 64 * 64
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        self.flat_grad = torch.zeros(off, dtype=torch.float32, device=device)
        self.params = {}
        for k in self.NAMES:
            v = self.flat[offs[k]: offs[k] + sizes[k]].view(shapes[k])
            v.requires_grad_(True)
            v.grad = self.flat_grad[offs[k]: offs[k] + sizes[k]].view(shapes[k])
            self.params[k] = v
        with torch.no_grad():
            self.params["xyz"].copy_(scene.means3D)
            self.params["f_dc"].copy_(scene.shs[:, :1])
            self.params["f_rest"].copy_(scene.shs[:, 1:])
            o = scene.opacities.clamp(1e-6, 1 - 1e-6)
            self.params["opacity"].copy_(torch.log(o / (1 - o)))  # inverse sigmoid
            self.params["scaling"].copy_(torch.log(scene.scales))
            self.params["rotation"].copy_(scene.rotations)

    def activated(self):
        """gaussian_model.py:92-117: exp / normalize / sigmoid / cat"""
        p = self.params
        return dict(means3D=p["xyz"], scales=torch.exp(p["scaling"]), rotations=F.normalize(p["rotation"]),
                    opacities=torch.sigmoid(p["opacity"]), shs=torch.cat((p["f_dc"], p["f_rest"]), dim=1))

    def make_optimizer(self):
        groups = [{"params": [self.params[k]], "lr": self.LRS[k], "name": k} for k in self.NAMES]
        try:
            return torch.optim.Adam(groups, lr=0.0, eps=1e-15, fused=self.flat.is_cuda)
        except (RuntimeError, TypeError):
            return torch.optim.Adam(groups, lr=0.0, eps=1e-15)


def render(params: GaussianParams, cam, bg, rasterizer_cls, settings_cls, sh_degree=3, debug=False):
    """gaussian_splatting/gaussian_renderer/__init__.py:18-100 with SH and scale/rotation handled in the rasterizer"""
    a = params.activated()
    dev = a["means3D"].device
    screenspace_points = torch.zeros_like(a["means3D"], requires_grad=True)
    settings = settings_cls(image_height=int(cam.image_height), image_width=int(cam.image_width), tanfovx=cam.tanfovx,
                            tanfovy=cam.tanfovy, bg=bg, scale_modifier=1.0, viewmatrix=cam.viewmatrix,
                            projmatrix=cam.projmatrix, sh_degree=sh_degree, campos=cam.campos, prefiltered=False,
                            debug=debug)
    rasterizer = rasterizer_cls(raster_settings=settings)
    image, radii = rasterizer(means3D=a["means3D"], means2D=screenspace_points, shs=a["shs"], colors_precomp=None,
                              opacities=a["opacities"], scales=a["scales"], rotations=a["rotations"], cov3D_precomp=None)
    return dict(render=image, viewspace_points=screenspace_points, visibility_filter=radii > 0, radii=radii)


class ViewShardedTrainer:
    """One optimisation step over a batch of `world_size` views, one view per rank."""

    def __init__(self, params: GaussianParams, rasterizer_cls, settings_cls, bg, sh_degree=3, lambda_dssim=0.2,
                 fused_loss=True):
        self.fused_loss = fused_loss
        self.params = params
        self.opt = params.make_optimizer()
        self.rasterizer_cls, self.settings_cls = rasterizer_cls, settings_cls
        self.bg, self.sh_degree, self.lambda_dssim = bg, sh_degree, lambda_dssim
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    def step(self, cam, gt_image):
        self.params.flat_grad.zero_()
        pkg = render(self.params, cam, self.bg, self.rasterizer_cls, self.settings_cls, self.sh_degree)
        loss = train_loss(pkg["render"], gt_image, self.lambda_dssim, self.fused_loss)
        loss.backward()
        if self.world > 1:
            # the only collective on the path: sum of the per-view parameter gradients (then mean over views)
            dist.all_reduce(self.params.flat_grad, op=dist.ReduceOp.SUM)
            self.params.flat_grad.mul_(1.0 / self.world)
        self.opt.step()
        return loss.detach(), pkg
