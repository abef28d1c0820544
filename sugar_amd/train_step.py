"""The "train step" the headline metric is quoted on, and its view-sharded data-parallel form.

Restates the vanilla 3DGS optimisation step that SuGaR builds on:
    render()                gaussian_splatting/gaussian_renderer/__init__.py:18-100
    loss = 0.8*L1 + 0.2*(1-SSIM), backward, Adam   gaussian_splatting/train.py:86-128
    l1_loss / ssim          sugar_utils/loss_utils.py:17-63
    parameter activations   gaussian_splatting/scene/gaussian_model.py:33-52 (exp / sigmoid / normalize)
    Adam groups and lrs     gaussian_splatting/scene/gaussian_model.py:152-166, arguments/__init__.py:74-83
On a ROCm device every piece runs on this repository's HIP kernels (rasterizer, fused loss, fused activations, SH-Adam and
flat Adam); on CPU tensors the same step runs on stock PyTorch ops over whatever rasterizer class is passed in (the tests
pass the oracle-backed stand-in) -- that path is the parity reference, not a fallback of the product.

Multi-GPU: the path shards by view.  Every rank holds a full replica of the Gaussians and renders its
own camera; the ranks exchange the clamp-masked colour gradients (all-gather, started from inside the rasterizer backward)
and the 11 non-SH gradient floats per Gaussian (all-reduce) -- RCCL over xGMI when the process group backend is "nccl", gloo
for the rehearsal tests -- and take identical Adam steps (see ViewShardedTrainer).
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
class _Activations(torch.autograd.Function):
    """exp / normalize / sigmoid of the raw parameters (sgr_activations_forward / _backward).  `sinks` = three tensors the
    backward writes the raw-parameter gradients into (and returns), or None to allocate."""

    @staticmethod
    def forward(ctx, scaling, rotation, opacity, sinks):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        if not scaling.is_cuda:
            raise RuntimeError("the fused activations need tensors on a ROCm device; there is no CPU fallback")
        P, dev = scaling.shape[0], scaling.device
        scales, rots, opac = torch.empty_like(scaling), torch.empty_like(rotation), torch.empty_like(opacity)
        vp = lambda t: C.c_void_p(t.data_ptr())
        with torch.cuda.device(dev):
            rc = lib.sgr_activations_forward(P, vp(scaling), vp(rotation), vp(opacity), vp(scales), vp(rots), vp(opac),
                                             C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc < 0:
            raise RuntimeError(f"sgr_activations_forward failed ({rc})")
        ctx.save_for_backward(scaling, rotation, opacity)
        ctx.sinks = sinks
        return scales, rots, opac

    @staticmethod
    def backward(ctx, g_scales, g_rots, g_opac):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        scaling, rotation, opacity = ctx.saved_tensors
        P, dev = scaling.shape[0], scaling.device
        gs = torch.zeros_like(scaling) if g_scales is None else g_scales.contiguous()
        gr = torch.zeros_like(rotation) if g_rots is None else g_rots.contiguous()
        go = torch.zeros_like(opacity) if g_opac is None else g_opac.contiguous()
        ds, dr, do = ctx.sinks if ctx.sinks is not None else (torch.empty_like(scaling), torch.empty_like(rotation),
                                                             torch.empty_like(opacity))
        vp = lambda t: C.c_void_p(t.data_ptr())
        with torch.cuda.device(dev):
            rc = lib.sgr_activations_backward(P, vp(scaling), vp(rotation), vp(opacity), vp(gs), vp(gr), vp(go), vp(ds), vp(dr),
                                              vp(do), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc < 0:
            raise RuntimeError(f"sgr_activations_backward failed ({rc})")
        return ds, dr, do, None


class GaussianParams:
    """Raw (pre-activation) 3DGS parameters, 59 floats per Gaussian at SH degree 3, stored as views of ONE flat buffer:
    the gradient exchange works on contiguous slices of it (the 11 non-SH floats per Gaussian are one prefix) and the
    optimiser is two streaming kernels.  The SH coefficients are ONE [P,M,3] tensor (the layout the rasterizer consumes); the reference keeps
    f_dc / f_rest apart and torch.cat()s them every step (gaussian_model.py:108-111: 192 MB of copies per step at 1M)."""

    NAMES = ("xyz", "opacity", "scaling", "rotation", "features")  # the SH tensor last: everything else is one prefix
    # arguments/__init__.py:74-83 (position lr at its initial value; f_rest runs at feature_lr / 20, gaussian_model.py:158)
    LRS = dict(xyz=0.00016, features=0.0025, opacity=0.05, scaling=0.005, rotation=0.001)
    REST_LR = 0.0025 / 20.0

    def __init__(self, scene, device):
        self._layout(scene.means3D.shape[0], scene.shs.shape[1], device)
        with torch.no_grad():
            self.params["xyz"].copy_(scene.means3D)
            self.params["features"].copy_(scene.shs)
            o = scene.opacities.clamp(1e-6, 1 - 1e-6)
            self.params["opacity"].copy_(torch.log(o / (1 - o)))  # inverse sigmoid
            self.params["scaling"].copy_(torch.log(scene.scales))
            self.params["rotation"].copy_(scene.rotations)

    @classmethod
    def from_raw(cls, tensors, device):
        """from the RAW parameter tensors by name (xyz [P,3], opacity [P,1], scaling [P,3], rotation [P,4], features [P,M,3]),
        copied bit for bit: what a topology change (sugar_amd.densify) hands back"""
        self = cls.__new__(cls)
        self._layout(tensors["xyz"].shape[0], tensors["features"].shape[1], device)
        with torch.no_grad():
            for k in self.NAMES:
                self.params[k].copy_(tensors[k].reshape(self.params[k].shape))
        return self

    def raw(self):
        """{name: detached view} of the raw parameters"""
        return {k: v.detach() for k, v in self.params.items()}

    def split_flat(self, flat):
        """{name: view} of a tensor laid out like `flat` (an Adam moment buffer, the gradient buffer)"""
        return {k: flat[self.offsets[k]: self.offsets[k] + self.sizes[k]].view(self.params[k].shape) for k in self.NAMES}

    def _layout(self, P, M, device):
        self.P, self.M = P, M
        shapes = dict(xyz=(P, 3), features=(P, M, 3), opacity=(P, 1), scaling=(P, 3), rotation=(P, 4))
        sizes = {k: int(torch.tensor(v).prod()) for k, v in shapes.items()}
        # keep every view 256-byte aligned inside the flat buffer (vector loads in the kernels)
        offs, off = {}, 0
        for k in self.NAMES:
            offs[k] = off
            off += (sizes[k] + 63) // 64 * 64
        self.offsets, self.sizes = offs, sizes
        self.n_small = offs["features"]  # flat[:n_small] = xyz | opacity | scaling | rotation (11 floats per Gaussian)
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        self.flat_grad = torch.zeros(off, dtype=torch.float32, device=device)
        self.params = {}
        for k in self.NAMES:
            v = self.flat[offs[k]: offs[k] + sizes[k]].view(shapes[k])
            v.requires_grad_(True)
            v.grad = self.flat_grad[offs[k]: offs[k] + sizes[k]].view(shapes[k])
            self.params[k] = v

    def activated(self, fused=None, raw=False):
        """gaussian_model.py:92-117: exp / normalize / sigmoid -- one HIP kernel each way on a ROCm device (the gradients
        land directly in the flat gradient buffer), stock torch ops otherwise.  `raw=True`: the raw tensors themselves, for a
        rasterizer call in raw-parameter mode (the activations then happen inside its preprocess kernels)."""
        p = self.params
        if raw:
            return dict(means3D=p["xyz"], scales=p["scaling"], rotations=p["rotation"], opacities=p["opacity"], shs=p["features"])
        if self.flat.is_cuda if fused is None else fused:
            scales, rotations, opacities = _Activations.apply(p["scaling"], p["rotation"], p["opacity"],
                                                             (p["scaling"].grad, p["rotation"].grad, p["opacity"].grad))
            return dict(means3D=p["xyz"], scales=scales, rotations=rotations, opacities=opacities, shs=p["features"])
        return dict(means3D=p["xyz"], scales=torch.exp(p["scaling"]), rotations=F.normalize(p["rotation"]),
                    opacities=torch.sigmoid(p["opacity"]), shs=p["features"])

    def make_optimizer(self):
        return FlatAdam(self) if self.flat.is_cuda else _torch_adam(self)


def _torch_adam(params: GaussianParams):
    """Stock torch.optim.Adam with the reference's six groups (gaussian_model.py:152-166); f_dc / f_rest are strided views
    of the single SH tensor.  Used on CPU (tests) and as the parity reference of FlatAdam."""
    p = params.params
    feat = p["features"]
    with torch.no_grad():
        f_dc, f_rest = feat[:, :1], feat[:, 1:]
    f_dc.grad, f_rest.grad = feat.grad[:, :1], feat.grad[:, 1:]
    groups = [{"params": [p["xyz"]], "lr": params.LRS["xyz"], "name": "xyz"},
              {"params": [f_dc], "lr": params.LRS["features"], "name": "f_dc"},
              {"params": [f_rest], "lr": params.REST_LR, "name": "f_rest"},
              {"params": [p["opacity"]], "lr": params.LRS["opacity"], "name": "opacity"},
              {"params": [p["scaling"]], "lr": params.LRS["scaling"], "name": "scaling"},
              {"params": [p["rotation"]], "lr": params.LRS["rotation"], "name": "rotation"}]
    return torch.optim.Adam(groups, lr=0.0, eps=1e-15, foreach=False)


class FlatAdam:
    """One-launch Adam over GaussianParams.flat (sugar_amd/csrc/adam.hip) with the same per-group learning rates."""

    def __init__(self, params: GaussianParams, betas=(0.9, 0.999), eps=1e-15):
        import ctypes as C
        from . import _lib
        self._C, self._lib = C, _lib.load()
        self.params = params
        self.betas, self.eps, self.t = betas, eps, 0
        self.exp_avg = torch.zeros_like(params.flat)
        self.exp_avg_sq = torch.zeros_like(params.flat)
        segs = []
        for k in params.NAMES:
            b, e = params.offsets[k], params.offsets[k] + params.sizes[k]
            if k == "features":
                segs.append((b, e, params.LRS[k], params.REST_LR, 3 * params.M, 3))
            else:
                segs.append((b, e, params.LRS[k], params.LRS[k], 1, 1))
        self._seg, self._n = self._pack(segs)
        self._seg_small, self._n_small = self._pack(segs[:-1])  # everything but the SH tensor (the last segment)
        self._dmean_extra = None  # [P,3] view-direction part of dL/dxyz written by the SH kernel (step(sh_dir=True))
        assert params.offsets["xyz"] == 0

    def _pack(self, segs):
        C = self._C
        n = len(segs)
        return ((C.c_longlong * n)(*[s[0] for s in segs]), (C.c_longlong * n)(*[s[1] for s in segs]),
                (C.c_float * n)(*[s[2] for s in segs]), (C.c_float * n)(*[s[3] for s in segs]),
                (C.c_int * n)(*[s[4] for s in segs]), (C.c_int * n)(*[s[5] for s in segs])), n

    def step(self, grad_scale: float = 1.0, sh_views=None, before_small=None, sh_dir=False):
        """One Adam step over the flat buffer.  With `sh_views = (means3D, campos_all[V,3], masked_colors[V,P,3], sh_degree)`
        the SH tensor is updated by sgr_sh_adam_from_views straight from the per-view colour gradients (its 48-float gradient
        is never materialised) and the flat kernel covers the other 11 floats per Gaussian only; `before_small` is called
        between the two launches.  `sh_dir`: the rasterizer backward ran with `sh_dir_elsewhere` (its position gradient
        lacks the term through the view direction): the SH kernel forms that term next to the coefficients it loads anyway
        and the flat kernel adds it to the position gradient."""
        self.begin_step()
        if sh_views is not None:
            extra = None
            if sh_dir:
                if self._dmean_extra is None:
                    self._dmean_extra = torch.empty(self.params.P, 3, dtype=torch.float32, device=self.params.flat.device)
                extra = self._dmean_extra
            self.step_sh(sh_views, grad_scale, dmean_extra=extra)
            if before_small is not None:
                before_small()  # e.g. wait for the all-reduce of the small gradients, which ran next to the SH kernel
            self.step_small(grad_scale, extra=extra)
        else:
            self._flat_step(self.params.flat.numel(), self._seg, self._n, grad_scale)

    def begin_step(self):
        self.t += 1

    def step_sh(self, sh_views, grad_scale: float = 1.0, dmean_extra=None):
        """The SH half of the step on the CURRENT stream (a trainer may run it on a second stream, next to step_small and the
        geometry half of the next forward).  `means3D` must hold the positions the views were rendered with."""
        C, p = self._C, self.params
        dev = p.flat.device
        vp = lambda t: C.c_void_p(t.data_ptr())
        means3D, campos_all, dcolor_all, sh_degree = sh_views
        off = p.offsets["features"]
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            dcol, vstride = _view_rows(dcolor_all)
            campos_all = campos_all.contiguous()  # (kept referenced until the call is enqueued)
            rc = self._lib.sgr_sh_adam_from_views_ex(
                p.P, int(dcolor_all.shape[0]), int(sh_degree), p.M, vp(means3D), vp(campos_all),
                vp(dcol), int(vstride), C.c_void_p(p.flat.data_ptr() + 4 * off),
                C.c_void_p(self.exp_avg.data_ptr() + 4 * off), C.c_void_p(self.exp_avg_sq.data_ptr() + 4 * off),
                p.LRS["features"], p.REST_LR, self.betas[0], self.betas[1], self.eps, self.t, float(grad_scale),
                C.c_void_p(dmean_extra.data_ptr()) if dmean_extra is not None else None, stream)
        if rc < 0:
            raise RuntimeError(f"sgr_sh_adam_from_views failed ({rc})")

    def step_small(self, grad_scale: float = 1.0, extra=None):
        """Everything but the SH tensor: the 11 other floats per Gaussian (positions included)."""
        self._flat_step(self.params.n_small, self._seg_small, self._n_small, grad_scale, extra)

    def _flat_step(self, n_flat, seg, n_seg, grad_scale, extra=None):
        C, p = self._C, self.params
        dev = p.flat.device
        vp = lambda t: C.c_void_p(t.data_ptr())
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            rc = self._lib.sgr_adam_step_ex(n_flat, vp(p.flat), vp(p.flat_grad), vp(self.exp_avg), vp(self.exp_avg_sq), n_seg,
                                            *seg, self.betas[0], self.betas[1], self.eps, self.t, float(grad_scale),
                                            vp(extra) if extra is not None else None, 3 * p.P if extra is not None else 0, stream)
        if rc < 0:
            raise RuntimeError(f"sgr_adam_step failed ({rc})")


def render(params: GaussianParams, cam, bg, rasterizer_cls, settings_cls, sh_degree=3, debug=False, means2D=None,
           visibility=True, raw_params=False):
    """gaussian_splatting/gaussian_renderer/__init__.py:18-100 with SH and scale/rotation handled in the rasterizer.
    `means2D`: a caller-owned [P,3] zero tensor with requires_grad (the reference zero-fills a fresh one per call, :27-31; its
    values are never read, it only carries dL/dmeans2D); `visibility=False` skips the `radii > 0` mask (:97)."""
    a = params.activated(raw=raw_params)  # raw_params: the caller has put the rasterizer into raw-parameter mode (grad_sink)
    dev = a["means3D"].device
    screenspace_points = torch.zeros_like(a["means3D"], requires_grad=True) if means2D is None else means2D
    settings = settings_cls(image_height=int(cam.image_height), image_width=int(cam.image_width), tanfovx=cam.tanfovx,
                            tanfovy=cam.tanfovy, bg=bg, scale_modifier=1.0, viewmatrix=cam.viewmatrix,
                            projmatrix=cam.projmatrix, sh_degree=sh_degree, campos=cam.campos, prefiltered=False,
                            debug=debug)
    rasterizer = rasterizer_cls(raster_settings=settings)
    image, radii = rasterizer(means3D=a["means3D"], means2D=screenspace_points, shs=a["shs"], colors_precomp=None,
                              opacities=a["opacities"], scales=a["scales"], rotations=a["rotations"], cov3D_precomp=None)
    return dict(render=image, viewspace_points=screenspace_points, visibility_filter=(radii > 0) if visibility else None,
                radii=radii)


def _current_sink():
    from . import diff_gaussian_rasterization as dgr
    return dict(dgr._GRAD_SINK)


def _view_rows(dcolor_all):
    """(tensor, rows between views) for a [V,P,3] tensor that is dense or a row-slice of a dense [V,P+k,3] buffer"""
    V, P = dcolor_all.shape[0], dcolor_all.shape[1]
    st = dcolor_all.stride()
    if st[2] == 1 and st[1] == 3 and (V == 1 or (st[0] % 3 == 0 and st[0] // 3 >= P)):
        return dcolor_all, (st[0] // 3 if V > 1 else P)
    return dcolor_all.contiguous(), P


def sh_grad_from_views(means3D, campos_all, dcolor_all, sh_degree, out):
    """out[P,M,3] = sum over views of basis(normalize(means3D - campos_v)) (x) dcolor_all[v]  (HIP: sgr_sh_grad_from_views)"""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    if not means3D.is_cuda:
        raise RuntimeError("sh_grad_from_views needs tensors on a ROCm device; there is no CPU fallback")
    V, P = dcolor_all.shape[0], means3D.shape[0]
    M = out.shape[1]
    dev = means3D.device
    with torch.cuda.device(dev):
        dcol, vstride = _view_rows(dcolor_all)
        means3D, campos_all = means3D.contiguous(), campos_all.contiguous()  # (kept referenced until the call is enqueued)
        rc = lib.sgr_sh_grad_from_views(P, V, int(sh_degree), M, C.c_void_p(means3D.data_ptr()),
                                        C.c_void_p(campos_all.data_ptr()),
                                        C.c_void_p(dcol.data_ptr()), int(vstride), C.c_void_p(out.data_ptr()),
                                        C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc < 0:
        raise RuntimeError(f"sgr_sh_grad_from_views failed ({rc})")
    return out


class ViewShardedTrainer:
    """One optimisation step over a batch of `world_size` views, one view per rank.

    Gradient exchange (world > 1).  The SH gradient of a view is the outer product  basis(view direction) (x) masked dL/dRGB
    (backward.cu:47-97), so instead of all-reducing 59 floats per Gaussian (236 MB at 1M: the ring is xGMI-link bound), the
    ranks all-gather the 3 masked colour gradients per Gaussian and view plus the camera centres, all-reduce the 11
    remaining floats, and every rank rebuilds  sum_v basis_v (x) g_v  locally (`compact_sh`): 3*V + 11 floats per Gaussian
    on the wire instead of 59."""

    def __init__(self, params: GaussianParams, rasterizer_cls, settings_cls, bg, sh_degree=3, lambda_dssim=0.2,
                 fused_loss=True, compact_sh=None, sh_grad_fn=None, grad_sink_cm=None, fused_sh_adam=None, sync_free=None,
                 visibility=False, fuse_activations=None, sh_dir_in_adam=None, force_collectives=False, params_only=True):
        self.fused_loss = fused_loss
        self.params = params
        self.opt = params.make_optimizer()
        self.rasterizer_cls, self.settings_cls = rasterizer_cls, settings_cls
        self.bg, self.sh_degree, self.lambda_dssim = bg, sh_degree, lambda_dssim
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # the gradient exchange runs when there is more than one rank -- or on request with a single-rank process group
        # (`force_collectives`: the all-gather started from inside the backward, the asynchronous all-reduce and their stream
        # semantics execute on the real backend, RCCL on a one-GPU box, with nothing to exchange)
        self.exchange = self.world > 1 or (bool(force_collectives) and dist.is_available() and dist.is_initialized())
        # `params_only=False`: the rasterizer backward also delivers dL/dmeans2D (pkg["viewspace_points"].grad), which the
        # densification statistics of train.py:111-123 / sugar_densifier.py:156-164 need; the lean default skips it
        self.params_only = bool(params_only)
        self.compact_sh = params.flat.is_cuda if compact_sh is None else bool(compact_sh)
        self.sh_grad_fn = sh_grad_fn or sh_grad_from_views
        # compact mode on a ROCm device: the SH gradient is consumed by the optimiser kernel itself
        self.fused_sh_adam = (self.compact_sh and isinstance(self.opt, FlatAdam) and sh_grad_fn is None) if fused_sh_adam is None \
            else bool(fused_sh_adam)
        # Optional (off by default): the other use of the SH coefficients in the backward, dRGB/d(view direction) -> dL/dxyz, can
        # move into the SH-Adam kernel as well: the rasterizer backward then never reads the SH tensor
        # (SGR_MODE_SH_DIR_ELSEWHERE), the SH-Adam kernel forms the term for all views (every rank has their colour gradients
        # and camera centres) and the flat Adam kernel adds it to the position gradient.  Measured: the backward preprocess
        # kernel drops from 0.085 to 0.045 ms and the SH-Adam kernel gains as much (a serial prologue per wave in a kernel
        # that lives on 12 waves per CU): +-1 % per step depending on the box.
        self.sh_dir_in_adam = bool(sh_dir_in_adam) and self.fused_sh_adam and params.flat.is_cuda and params.M == 16
        if grad_sink_cm is None and params.flat.is_cuda:
            from .diff_gaussian_rasterization import grad_sink as grad_sink_cm
        self.grad_sink_cm = grad_sink_cm  # context manager factory honoured by the rasterizer's backward (None: plain autograd)
        self._send = self._recv = None    # all-gather buffers of the compact exchange
        self._work = None                 # the all-gather in flight
        # Sync-free forward (sgr_forward_ex with a binning capacity): the rasterizer does not wait for num_rendered in the
        # middle of the forward; the trainer reads the device header behind the loss and repeats an (extremely rare) forward
        # whose instance list outgrew the capacity, before any backward or collective runs.
        on_hip = params.flat.is_cuda and self.grad_sink_cm is not None
        self.sync_free = on_hip if sync_free is None else bool(sync_free)
        if self.sync_free and not on_hip:
            raise ValueError("sync_free needs the HIP rasterizer")
        # exp / normalize / sigmoid of the raw parameters inside the rasterizer's preprocess kernels (raw-parameter mode) instead
        # of two stand-alone kernels and a round trip of the activated values and their gradients through memory
        self.fuse_activations = on_hip if fuse_activations is None else bool(fuse_activations)
        if self.fuse_activations and not on_hip:
            raise ValueError("fuse_activations needs the HIP rasterizer")
        self._bin_cap = 0                 # instances the binning buffer is sized for (0: next forward runs with the host round trip)
        self._hdr = torch.zeros(16, dtype=torch.int32).pin_memory() if self.sync_free else None
        self._ev_hdr = torch.cuda.Event() if self.sync_free else None
        self.last_num_rendered = 0
        self.redone = 0                   # forwards repeated because the capacity was too small
        self.visibility = visibility      # the step itself has no use for the radii > 0 mask
        self._means2D = (torch.zeros(params.P, 3, dtype=torch.float32, device=params.flat.device, requires_grad=True)
                         if params.flat.is_cuda else None)
        self._one = torch.ones((), dtype=torch.float32, device=params.flat.device)  # d loss / d loss, not re-filled every step

    def _start_gather(self, colors):
        """Called by the rasterizer backward between its two halves: the masked colour gradients (already in the send
        buffer, followed by the camera centre) are final, the backward preprocess has not been launched yet -- the all-gather
        runs on the communication stream next to it."""
        if colors.data_ptr() != self._send.data_ptr():
            self._send[: colors.shape[0]].copy_(colors)
        self._work = dist.all_gather_into_tensor(self._recv, self._send, async_op=True)

    def _forward(self, cam, gt_image, capacity):
        from .diff_gaussian_rasterization import grad_sink
        extra = dict(binning_capacity=capacity, header_out=self._hdr, header_event=self._ev_hdr) if capacity else {}
        import contextlib
        with (grad_sink(**{**_current_sink(), **extra}) if extra else contextlib.nullcontext()):
            pkg = render(self.params, cam, self.bg, self.rasterizer_cls, self.settings_cls, self.sh_degree,
                         means2D=self._means2D, visibility=self.visibility, raw_params=self.fuse_activations)
            loss = train_loss(pkg["render"], gt_image, self.lambda_dssim, self.fused_loss)
        return pkg, loss

    @staticmethod
    def _last_forward_R():
        try:
            from .diff_gaussian_rasterization import _C
            return int(_C.last_forward.get("num_rendered", -1))
        except Exception:
            return -1

    def step(self, cam, gt_image):
        """gaussian_splatting/train.py:86-128 for one view per rank.  Gradients are taken with torch.autograd.grad and
        land in the flat gradient buffer: the large ones are written there by the rasterizer backward itself (grad_sink),
        the small ones are copied; nothing is zero-filled or accumulated."""
        import contextlib
        p = self.params
        names = list(p.NAMES)
        leaves = [p.params[k] for k in names]
        holder = {}
        if self.grad_sink_cm is not None:
            sinks = dict(means3D=p.params["xyz"].grad, params_only=self.params_only)
            if self.fuse_activations:
                sinks.update(raw_params=True, scales=p.params["scaling"].grad, rotations=p.params["rotation"].grad,
                             opacities=p.params["opacity"].grad)
            if self.compact_sh:
                sinks.update(compact_sh=True, out=holder, sh_dir_elsewhere=self.sh_dir_in_adam)
                if self.exchange:
                    # the masked colour gradients land in the send buffer of the all-gather, followed by the camera centre
                    if self._send is None:
                        self._send = torch.empty(p.P + 1, 3, dtype=torch.float32, device=p.flat.device)
                        self._recv = torch.empty(self.world * (p.P + 1), 3, dtype=torch.float32, device=p.flat.device)
                    self._send[p.P:].copy_(cam.campos.reshape(1, 3))
                    self._work = None
                    sinks.update(colors=self._send[: p.P], on_colors=self._start_gather)
            else:
                sinks.update(shs=p.params["features"].grad)
            ctxm = self.grad_sink_cm(**sinks)
        else:
            ctxm = contextlib.nullcontext()
        cap = self._bin_cap if self.sync_free else 0
        with ctxm:
            pkg, loss = self._forward(cam, gt_image, cap)
            if cap:
                # the header travels behind the forward's blend kernel (the loss kernels are already queued behind it)
                self._ev_hdr.synchronize()
                R = int(self._hdr[0]) & 0xFFFFFFFF
                if R > cap or int(self._hdr[6]) != 0 or int(self._hdr[8 + 3]) != 0:
                    self.redone += 1
                    pkg, loss = self._forward(cam, gt_image, 0)  # with the host round trip: any size
                    R = self._last_forward_R()
            else:
                R = self._last_forward_R()
            if self.sync_free and R >= 0 and (cap == 0 or 5 * R > 4 * cap):
                self._bin_cap = R + R // 2 + 65536
            self.last_num_rendered = R
            grads = torch.autograd.grad(loss, leaves, grad_outputs=self._one, allow_unused=True)
        sh_views = wait_small = None
        scale = 1.0 / self.world
        with torch.no_grad():
            for name, leaf, g in zip(names, leaves, grads):
                if g is None:
                    if not (self.compact_sh and name == "features"):
                        leaf.grad.zero_()
                elif g.data_ptr() != leaf.grad.data_ptr():
                    leaf.grad.copy_(g)
            if self.compact_sh and "masked_colors" not in holder:
                raise RuntimeError("compact_sh: the rasterizer backward did not deliver the masked colour gradients")
            if self.compact_sh:
                g_rgb = holder["masked_colors"].contiguous()
                campos = cam.campos.reshape(1, 3).to(g_rgb.dtype).contiguous()
                if self.exchange:
                    P_ = g_rgb.shape[0]
                    if self._send is None or g_rgb.device != self._send.device:  # (a rasterizer that ignored the sink)
                        self._send = torch.empty(P_ + 1, 3, dtype=g_rgb.dtype, device=g_rgb.device)
                        self._recv = torch.empty(self.world * (P_ + 1), 3, dtype=g_rgb.dtype, device=g_rgb.device)
                    if self._work is not None:
                        self._work.wait()  # the all-gather was started from inside the rasterizer backward (_start_gather)
                        self._work = None
                    else:
                        if g_rgb.data_ptr() != self._send.data_ptr():
                            self._send[:P_].copy_(g_rgb)
                        self._send[P_:].copy_(campos)
                        dist.all_gather_into_tensor(self._recv, self._send)  # one collective: colours and camera centres
                    blocks = self._recv.view(self.world, P_ + 1, 3)
                    all_rgb, all_cam = blocks[:, :P_], blocks[:, P_].contiguous()
                    if self.fused_sh_adam:
                        # the SH-Adam kernel needs the gathered colours only: the all-reduce of the 11 small floats runs next
                        # to it and is waited for just before the flat Adam kernel
                        wait_small = dist.all_reduce(p.flat_grad[: p.n_small], op=dist.ReduceOp.SUM, async_op=True).wait
                    else:
                        dist.all_reduce(p.flat_grad[: p.n_small], op=dist.ReduceOp.SUM)
                else:
                    all_rgb, all_cam = g_rgb[None], campos
                if self.fused_sh_adam:
                    sh_views = (p.params["xyz"].detach(), all_cam, all_rgb, self.sh_degree)
                else:
                    self.sh_grad_fn(p.params["xyz"].detach(), all_cam, all_rgb, self.sh_degree, p.params["features"].grad)
            elif self.exchange:
                # plain path: one flat all-reduce of all 59 floats per Gaussian
                dist.all_reduce(p.flat_grad, op=dist.ReduceOp.SUM)
        if isinstance(self.opt, FlatAdam):
            self.opt.step(grad_scale=scale, sh_views=sh_views, before_small=wait_small,  # (the mean over views is folded in)
                          sh_dir=self.sh_dir_in_adam and sh_views is not None)
        else:
            if self.world > 1:
                p.flat_grad.mul_(scale)
            self.opt.step()
        return loss.detach(), pkg


def exchange_pieces(P: int, n_small: int, pieces: int):
    """Cut points of the gradient exchange in `pieces` pieces (csrc/train.hip: sgr_trainer_step_exchange uses the same rule):
    Gaussian ranges [g[k], g[k+1]) on multiples of 256 for the colour all-gather / SH-Adam, float ranges [f[k], f[k+1]) on multiples
    of 1024 for the small all-reduce / flat Adam; empty pieces are allowed (tiny models), the last piece takes the remainder."""
    g = [0] + [(P * k // pieces) & ~255 for k in range(1, pieces)] + [P]
    f = [0] + [(n_small * k // pieces) & ~1023 for k in range(1, pieces)] + [n_small]
    return g, f


class NativeTrainer:
    """ViewShardedTrainer.step with the interpreter taken out: one `sgr_trainer_step` call (include/sugar_raster.h) enqueues the
    whole step -- sync-free rasterizer forward in raw-parameter mode, fused loss and its backward, rasterizer backward into
    the flat gradient buffer, SH-Adam from the colour gradients, flat Adam over the other 11 floats per Gaussian -- on the
    current stream, from buffers allocated once.  Same arithmetic, same kernels, same parameter layout (GaussianParams).

    Validity without a host wait.  The forward runs with a list capacity and, from a camera's second visit on, with a WALK
    HINT (per tile: how many list entries it walked last time, times 1 + hint_margin, + 64): the list-write pass then skips the
    chunks nobody will read.  Since round 5 a tile that outruns its hint is rendered again INSIDE the same forward (its list is
    completed and its blocks re-blended by two gated launches: `repaired_tiles`); only more than 1024 such tiles in one view
    count as a miss.  Three misses within 32 forwards switch the hints off for the next 64 (a scene that changes this
    fast is cheaper to bin in full than to render twice).  A forward that outgrew the capacity or the hint makes every later kernel of the step -- backward and Adam
    included -- a no-op on the device; the host finds out when it next looks at the pinned header (before enqueueing the
    following step, when the copy has long arrived), enlarges the capacity / drops the hint and repeats the step.

    With a gradient exchange (several ranks, or `force_collectives` on a one-rank group) the step is enqueued in its four
    phases with the collectives between them, exactly where ViewShardedTrainer puts them, and the validity check is made
    right behind the forward (the loss and the blend backward are queued by then), before anything is sent."""

    def __init__(self, params: GaussianParams, bg, width, height, sh_degree=3, lambda_dssim=0.2, densify_stats=False,
                 force_collectives=False, walk_hint=True, capacity=None, betas=(0.9, 0.999), eps=1e-15, hint_margin=None,
                 launch_order=True, native_collectives=None):
        import ctypes as C
        from . import _lib
        if not params.flat.is_cuda:
            raise RuntimeError("NativeTrainer needs the parameters on a ROCm device; there is no CPU path")
        self._C, self._L, self._lib = C, _lib, _lib.load()
        lib = self._lib
        self.params, self.bg = params, bg.to(params.flat.device).float().contiguous()
        self.W, self.H, self.sh_degree = int(width), int(height), int(sh_degree)
        dev = params.flat.device
        self.dev = dev
        P, M = params.P, params.M
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.exchange = self.world > 1 or (bool(force_collectives) and dist.is_available() and dist.is_initialized())
        # pieces of the colour all-gather / small all-reduce (`set_exchange_chunks`; any value gives the same bits): every piece costs
        # two launches and an event wait, and on one or two ranks there is next to nothing on the wire to hide behind them
        self.exchange_chunks = 1 if self.world <= 1 else (2 if self.world == 2 else 4)
        self.walk_hint = bool(walk_hint)
        self.launch_order = bool(launch_order)  # sgr_forward_opts.tile_order: the blend kernels start with the deepest tiles
        import os as _os
        # measured on the metric scene while Adam's first steps move it (bench.py, 600 steps): margin 0.25 -> 24 forwards
        # repeated, 0.5 -> 13, 1.0 -> 4; list-write pass 57 -> 37 us at 0.5 (a wider margin skips less)
        self.hint_margin = float(hint_margin if hint_margin is not None else _os.environ.get("SGR_HINT_MARGIN", 0.5))
        self._recent_misses = []   # step numbers of the latest hint misses
        self._hint_pause_until = 0
        self.hint_pauses = 0       # times the hints were switched off for 64 forwards (three misses within 32)
        self.host_work_s = 0.0     # time spent preparing and enqueueing steps (the waits for the header are not in it)
        self._calls = 0            # forwards enqueued so far (the clock of the hint policy; `t` is Adam's and may be reset)
        self.exp_avg = torch.zeros_like(params.flat)
        self.exp_avg_sq = torch.zeros_like(params.flat)
        self.t = 0
        self.capacity = int(capacity) if capacity else 24 * P + (1 << 20)
        self.densify_stats = bool(densify_stats)
        self._betas, self._eps, self._lambda_dssim = betas, eps, lambda_dssim
        self._hdr = torch.zeros(16, dtype=torch.int32).pin_memory()
        self._deep_min = int(lib.sgr_get_deep_min()) or (1 << 31)   # (0 = the long-list kernel is off)
        self._last_max_need = 0
        self._hints = {}       # camera key -> int32[T] walk hint
        self._pending = None   # (cam, gt, key) of the step whose forward has not been validated yet
        self.redone = 0
        self.last_num_rendered = 0
        self.repaired_tiles, self.last_repaired_tiles = 0, 0
        self.T = ((self.W + 15) // 16) * ((self.H + 15) // 16)
        self._h = None
        self._allocate()
        # The gradient exchange inside the library (sgr_trainer_step_exchange: RCCL bound at run time, collectives on the library's own
        # stream, ONE call per step, no interpreter between the phases).  Round 6: the DEFAULT with the RCCL backend on several ranks
        # (measured on one rank with forced collectives: 0.10 ms of exchange overhead per step against 0.28 ms for the
        # torch.distributed path, whose ten phase calls per step the host cannot enqueue inside the 0.27 ms the GPU still has queued
        # when the forward's header arrives); `native_collectives=False` or SGR_NATIVE_COLLECTIVES=0 selects the torch.distributed
        # path (the one the two-rank gloo tests exercise; same plan, same kernels, same bits).
        explicit = native_collectives is True   # (asked for by the caller: a failure is an error, not a fall-back)
        if native_collectives is None:
            env = _os.environ.get("SGR_NATIVE_COLLECTIVES")
            native_collectives = (env == "1") if env is not None else (self.exchange and self.world > 1 and dist.get_backend() == "nccl")
        self.native_collectives = False
        if native_collectives and self.exchange:
            if dist.get_backend() != "nccl":
                raise RuntimeError("native_collectives needs the RCCL (\"nccl\") process group backend")
            # Every rank must end up on the SAME path: a rank that cannot bring its communicator up (RCCL not loadable, a failed
            # init) while the others can would leave them waiting in a collective it never joins.  So the outcome is agreed on
            # through the process group (a MIN all-reduce of "it worked here"), and if it failed anywhere every rank tears its
            # communicator down again and takes the torch.distributed exchange (explicitly requested: an error instead).
            idbuf = C.create_string_buffer(128)
            have_id = 1 if (dist.get_rank() != 0 or lib.sgr_rccl_unique_id(idbuf) >= 0) else 0
            box = [idbuf.raw, have_id]
            dist.broadcast_object_list(box, src=0)
            ok, why = bool(box[1]), "sgr_rccl_unique_id failed on rank 0"
            if ok:
                with torch.cuda.device(dev):
                    rc = lib.sgr_trainer_comm_init(self._h, box[0], self.world, dist.get_rank(), self._recv.data_ptr(), self._recv.numel() * 4)
                ok = rc >= 0
                if not ok:
                    why = "sgr_trainer_comm_init failed: " + lib.sgr_trainer_last_error().decode(errors="replace")
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                self.native_collectives = True
                lib.sgr_trainer_set_exchange_chunks(self._h, self.exchange_chunks)
            else:
                if ok:
                    lib.sgr_trainer_comm_destroy(self._h)
                    why = "another rank could not initialise its communicator"
                if explicit:
                    raise RuntimeError("native_collectives: " + why)
                import warnings
                warnings.warn("in-library gradient exchange unavailable (" + why + "): using torch.distributed collectives")

    def set_exchange_chunks(self, n: int):
        """pieces of the gradient exchange (1 .. 16); the results do not depend on it (on one or two ranks: bit for bit)"""
        n = int(n)
        if not 1 <= n <= 16:
            raise ValueError("exchange chunks: 1 .. 16")
        self.exchange_chunks = n
        if getattr(self, "native_collectives", False) and self._lib.sgr_trainer_set_exchange_chunks(self._h, n) < 0:
            raise RuntimeError("sgr_trainer_set_exchange_chunks failed")

    def _allocate(self):
        """everything sized by the Gaussian count, and the library handle over it (construction, and again after `resize`)"""
        C, lib, _lib, params, dev = self._C, self._lib, self._L, self.params, self.dev
        P, M = params.P, params.M
        u8 = lambda n: torch.empty(int(n), dtype=torch.uint8, device=dev)
        self._geom = u8(lib.sgr_geom_bytes(P))
        self._img = u8(lib.sgr_img_bytes(self.W, self.H) + lib.sgr_bin2_bytes(P, self.W, self.H))
        self._binning = u8(lib.sgr_binning_bytes(self.capacity, self.W, self.H))
        self._loss_scratch = u8(lib.sgr_l1_ssim_scratch_bytes(3, self.W, self.H))
        self.image = torch.empty(3, self.H, self.W, device=dev)
        self._grad_image = torch.empty(3, self.H, self.W, device=dev)
        self.loss_out = torch.zeros(3, device=dev)
        self._send = torch.zeros(P + 1, 3, device=dev)
        self._recv = torch.empty(self.world * (P + 1), 3, device=dev) if self.exchange else None   # chunk k: its own [world][len_k][3] block
        self._cams_all = torch.empty(self.world, 3, device=dev) if self.exchange else None          # every view's camera centre
        self.radii = torch.zeros(P, dtype=torch.int32, device=dev)
        if self.densify_stats:  # gaussian_model.py:125-127 / sugar_densifier.py:152-154
            self.viewspace_grad = torch.zeros(P, 3, device=dev)
            self.max_radii2D = torch.zeros(P, device=dev)
            self.xyz_gradient_accum = torch.zeros(P, device=dev)
            self.denom = torch.zeros(P, device=dev)
        o, vp = params.offsets, (lambda t: t.data_ptr() if t is not None else None)
        st = (lambda name: vp(getattr(self, name))) if self.densify_stats else (lambda name: None)
        betas = self._betas
        self._cfg = _lib.TrainConfig(
            P, self.sh_degree, M, self.W, self.H, vp(params.flat), vp(params.flat_grad), vp(self.exp_avg), vp(self.exp_avg_sq),
            o["xyz"], o["opacity"], o["scaling"], o["rotation"], o["features"], params.n_small,
            params.LRS["xyz"], params.LRS["opacity"], params.LRS["scaling"], params.LRS["rotation"], params.LRS["features"],
            params.REST_LR, betas[0], betas[1], self._eps, self._lambda_dssim, vp(self.bg), vp(self._geom), self._geom.numel(),
            vp(self._img), self._img.numel(), vp(self._binning), self._binning.numel(), self.capacity, vp(self._loss_scratch),
            vp(self.image), vp(self._grad_image), vp(self.loss_out), vp(self._send), vp(self.radii), vp(self._hdr),
            st("viewspace_grad"), st("max_radii2D"), st("xyz_gradient_accum"), st("denom"))
        self._h = lib.sgr_trainer_create(C.byref(self._cfg))
        if not self._h:
            raise RuntimeError("sgr_trainer_create failed: " + lib.sgr_trainer_last_error().decode(errors="replace"))

    # ---- topology changes: statistics over all ranks, identical densification, new buffers
    def all_reduce_densification_stats(self, group=None):
        """before a densification event: `xyz_gradient_accum` / `denom` summed, `max_radii2D` maximised over the ranks
        (sugar_densifier.py:156-164); the step in flight is validated first"""
        if not self.densify_stats:
            raise RuntimeError("all_reduce_densification_stats: the trainer was built without densify_stats=True")
        self.synchronize()
        from .view_parallel import all_reduce_densification_stats
        all_reduce_densification_stats(self, group=group)

    def densify_and_prune(self, max_grad=0.0002, min_opacity=0.005, extent=1.0, max_screen_size=None, seed=None, group=None):
        """gaussian_model.py:390-403 on the trainer's own buffers: statistics exchanged over the ranks, clone / split / prune with a
        generator every rank seeds identically, then `resize`.  Returns (cloned, split, pruned).
        The split offsets come from ONE generator per trainer that advances from event to event (the reference draws from the
        advancing global RNG, gaussian_model.py:352-356): seeded once (0) and identically on every rank; an explicit `seed` reseeds it."""
        from . import densify
        if self.world > 1 or (dist.is_available() and dist.is_initialized()):
            self.all_reduce_densification_stats(group)
        else:
            self.synchronize()
        if seed is not None or getattr(self, "_densify_gen", None) is None:
            self._densify_gen = torch.Generator(device=self.dev).manual_seed(int(seed or 0))
        g = self._densify_gen
        p = self.params
        stats = dict(xyz_gradient_accum=self.xyz_gradient_accum, denom=self.denom, max_radii2D=self.max_radii2D)
        t, m1, m2, nc, ns, npr = densify.densify_and_prune(p.raw(), p.split_flat(self.exp_avg), p.split_flat(self.exp_avg_sq), stats,
                                                           max_grad=max_grad, min_opacity=min_opacity, extent=extent,
                                                           max_screen_size=max_screen_size, generator=g)
        self.resize(t, m1, m2)
        return nc, ns, npr

    def resize(self, tensors, exp_avg=None, exp_avg_sq=None):
        """new topology: the raw parameter tensors by name (and optionally their Adam moments; zeros otherwise).  Parameters move
        into a fresh flat buffer (`self.params` is a NEW GaussianParams), every per-Gaussian buffer is reallocated, the
        densification statistics start from zero (gaussian_model.py:343-345), the step counter, the per-camera walk hints and launch
        orders (per tile, not per Gaussian) are kept; the list capacity scales with the Gaussian count."""
        self.synchronize()
        old_P = self.params.P
        if getattr(self, "native_collectives", False):
            raise RuntimeError("resize: not with the in-library exchange (its receive buffer is registered with the communicator)")
        if self._h:
            self._lib.sgr_trainer_destroy(self._h)
            self._h = None
        params = GaussianParams.from_raw(tensors, self.dev)
        self.params = params
        self.exp_avg = torch.zeros_like(params.flat)
        self.exp_avg_sq = torch.zeros_like(params.flat)
        with torch.no_grad():
            for buf, src in ((self.exp_avg, exp_avg), (self.exp_avg_sq, exp_avg_sq)):
                if src is not None:
                    for k, v in params.split_flat(buf).items():
                        v.copy_(src[k].reshape(v.shape))
        self.capacity = max(int(self.capacity * (params.P / max(old_P, 1)) * 1.1), 1 << 20)
        self._pending = None
        self._allocate()
        return params


    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.sgr_trainer_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- one phase mask, one call
    def _call(self, cam, gt, key, phases, ex, use_hint=True, exchange_step=None):
        import time as _time
        _t0 = _time.perf_counter()
        C, L = self._C, self._L
        need = need_out = order = order_out = None
        if phases & 1:
            ent = self._hints.get(key)
            if ent is None:  # per camera: [walk hint, hint usable, level-2 chunks of the last validated visit, launch order, order usable]
                ent = self._hints[key] = [torch.zeros(self.T, dtype=torch.int32, device=self.dev), False, 0,
                                          torch.zeros(self.T, dtype=torch.int32, device=self.dev), False, 0]   # [5]: largest hint written
            if self.walk_hint:
                # (usable once a forward that wrote it was validated, and not while the hints are paused)
                need = ent[0].data_ptr() if (ent[1] and use_hint and self._calls >= self._hint_pause_until) else None
                need_out = ent[0].data_ptr()
            if self.launch_order:
                # deepest tiles first, as the camera's previous validated visit sorted them (an invalid forward leaves the buffer
                # as it was); the same sort serves this step's backward
                order = ent[3].data_ptr() if ent[4] else None
                order_out = ent[3].data_ptr()
            self._last_hinted = need is not None
            self._calls += 1
        # long lists: the eight-wave blend kernel (and the side stream it runs on) only for a camera whose hint has a tile above the
        # library's threshold -- the largest hint of its last validated visit came back in the host header
        ent5 = self._hints.get(key)
        deep = 1 if (need is not None and ent5 is not None and ent5[5] > self._deep_min) else 0
        view = L.TrainView(cam.viewmatrix.data_ptr(), cam.projmatrix.data_ptr(), cam.campos.data_ptr(), cam.tanfovx, cam.tanfovy,
                           gt.data_ptr(), need, need_out, self.hint_margin, self._chunk_grid(key), order, order_out, deep)
        with torch.cuda.device(self.dev):
            stream = C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
            if exchange_step is not None:
                rc = self._lib.sgr_trainer_step_exchange(self._h, C.byref(view), int(exchange_step), stream)
            else:
                rc = self._lib.sgr_trainer_step(self._h, C.byref(view), phases, C.byref(ex) if ex is not None else None, stream)
        self.host_work_s += _time.perf_counter() - _t0
        if exchange_step is not None:  # (the wait for the forward's header inside the call is not the host's own work)
            self.host_work_s -= 1e-3 * float(self._lib.sgr_trainer_last_exchange_wait_ms(self._h))
        if rc < 0:
            raise RuntimeError(f"sgr_trainer_step failed ({rc}): " + self._lib.sgr_trainer_last_error().decode(errors="replace"))
        return rc

    def _chunk_grid(self, key):
        """workgroups worth launching for the level-2 tile passes: what this camera's last visit had, plus a quarter"""
        ent = self._hints.get(key)
        n = ent[2] if ent is not None else 0
        return (n + n // 4 + 64) if n else 0

    def _grow(self, R):
        self.capacity = int(R) + int(R) // 2 + 65536
        self._binning = torch.empty(int(self._lib.sgr_binning_bytes(self.capacity, self.W, self.H)), dtype=torch.uint8, device=self.dev)
        rc = self._lib.sgr_trainer_set_binning(self._h, self._binning.data_ptr(), self._binning.numel(), self.capacity)
        if rc < 0:
            raise RuntimeError("sgr_trainer_set_binning failed")

    def _valid(self):
        """waits for the header of the most recent forward; returns (ok, overflow R or 0, hint missed)"""
        hdr = (self._C.c_uint32 * 16)()
        ok = self._lib.sgr_trainer_forward_valid(self._h, hdr)
        if ok < 0:
            raise RuntimeError("sgr_trainer_forward_valid failed")
        self.last_num_rendered = int(hdr[0])
        self._last_chunks = int(hdr[5])
        self.last_repaired_tiles = int(hdr[8 + 7])   # tiles that outran their walk hint and were rendered again in place
        self._last_max_need = int(hdr[8 + 1])        # the largest walk hint this forward wrote (tile_order.h)
        if ok:
            self.repaired_tiles += self.last_repaired_tiles
        return bool(ok), (int(hdr[0]) if int(hdr[0]) > self.capacity else 0), bool(hdr[8 + 3])

    def _hint_feedback(self, missed):
        if missed:
            self._recent_misses = [t for t in self._recent_misses if t > self._calls - 32] + [self._calls]
            if len(self._recent_misses) >= 3:
                self._hint_pause_until = self._calls + 64
                self._recent_misses = []
                self.hint_pauses += 1

    def _repair(self, key, R, missed):
        self.redone += 1
        self._hint_feedback(missed)
        if R:
            torch.cuda.synchronize(self.dev)  # (the old list buffer may still be in use by queued kernels)
            self._grow(R)
        if missed and key in self._hints:
            self._hints[key][1] = False  # the next forward of this camera runs unhinted and leaves a fresh hint

    def _resolve(self):
        """validate the step that is still in flight; repeat it until its forward is valid"""
        while self._pending is not None:
            cam, gt, key = self._pending
            ok, R, missed = self._valid()
            if ok:
                if key in self._hints:
                    self._hints[key][1] = True
                    self._hints[key][2] = self._last_chunks
                    self._hints[key][4] = True
                    self._hints[key][5] = self._last_max_need
                self._hint_feedback(False)
                self._pending = None
                return
            if not (R or missed):
                raise RuntimeError("level-1 binning overflow: this view needs the single-level binning (use ViewShardedTrainer)")
            self._repair(key, R, missed)
            ex = self._L.TrainExchange(1, None, 0, None, 1.0, self.t)
            self._call(cam, gt, key, 15, ex)

    def step(self, cam, gt_image, cam_key=None):
        """One optimisation step on `cam` (device tensors) against `gt_image` [3,H,W]; returns the loss as a device scalar
        (a view of `self.loss_out`: read it before the next step, or clone it).

        Without a gradient exchange the step is validated ONE STEP LATE (see the class docstring): until the next `step()` or
        `synchronize()` has looked at the forward's header, `loss_out` / `image` may belong to a forward that did not happen
        (list capacity or walk hint exceeded: the blend wrote nothing and the loss kernel saw the previous image).  Call
        `synchronize()` before reading them -- it repeats such a step -- as bench.py and the tests do."""
        key = id(cam) if cam_key is None else cam_key
        if not self.exchange:
            self._resolve()
            self.t += 1
            ex = self._L.TrainExchange(1, None, 0, None, 1.0, self.t)
            self._call(cam, gt_image, key, 15, ex)
            self._pending = (cam, gt_image, key)
            return self.loss_out[0]
        # ---- with the gradient exchange
        P, world = self.params.P, self.world
        self.t += 1
        if self.native_collectives:  # ONE call: the phases, the header check and the RCCL collectives are enqueued by the library
            while True:
                rc = self._call(cam, gt_image, key, 1, None, exchange_step=self.t)
                ok, R, missed = self._valid()  # (the call has already waited for this header: no second wait)
                if rc == 0:
                    if key in self._hints:
                        self._hints[key][1] = True
                        self._hints[key][2] = self._last_chunks
                        self._hints[key][4] = True
                        self._hints[key][5] = self._last_max_need
                    self._hint_feedback(False)
                    return self.loss_out[0]
                if not (R or missed):
                    # (the other ranks are already waiting in the all-gather this rank will never join: abort the communicator so
                    # that they fail instead of hanging)
                    self._lib.sgr_trainer_comm_abort(self._h)
                    raise RuntimeError("level-1 binning overflow: this view needs the single-level binning (use ViewShardedTrainer)")
                self._repair(key, R, missed)
        # ... or as four phase calls with torch.distributed collectives between them
        while True:
            self._call(cam, gt_image, key, 1, None)
            ok, R, missed = self._valid()
            if ok:
                if key in self._hints:
                    self._hints[key][1] = True
                    self._hints[key][2] = self._last_chunks
                    self._hints[key][4] = True
                    self._hints[key][5] = self._last_max_need
                self._hint_feedback(False)
                break
            if not (R or missed):
                raise RuntimeError("level-1 binning overflow: this view needs the single-level binning (use ViewShardedTrainer)")
            self._repair(key, R, missed)
        # The exchange in `exchange_chunks` pieces (round 6; the same plan as sgr_trainer_step_exchange, csrc/train.hip): the camera
        # centres, then the colour gradients of Gaussian chunk 0 .. C-1 -- each into the chunk's own [world][len][3] block of the
        # receive buffer -- beside the preprocess half; then the 11 small floats per Gaussian as C slices of the flat gradient
        # buffer beside the SH half of Adam.  SH-Adam of chunk k is enqueued behind ITS gather, flat Adam of slice j behind ITS
        # reduction: only the first piece of each collective (and whatever the wire cannot hide) is exposed.
        C = self.exchange_chunks
        n_small = self.params.n_small
        g_cut, f_cut = exchange_pieces(P, n_small, C)
        g_at, f_at = (lambda k: g_cut[k]), (lambda k: f_cut[k])
        send, recv = self._send.view(-1), self._recv.view(-1)
        w_cam = dist.all_gather_into_tensor(self._cams_all.view(-1), send[3 * P: 3 * P + 3], async_op=True)
        gathers = []
        for k in range(C):
            g0, g1 = g_at(k), g_at(k + 1)
            gathers.append(dist.all_gather_into_tensor(recv[3 * world * g0: 3 * world * g1], send[3 * g0: 3 * g1], async_op=True)
                           if g1 > g0 else None)
        self._call(cam, gt_image, key, 2, None)
        fg = self.params.flat_grad
        reduces = []
        for k in range(C):
            f0, f1 = f_at(k), f_at(k + 1)
            reduces.append(dist.all_reduce(fg[f0:f1], op=dist.ReduceOp.SUM, async_op=True) if f1 > f0 else None)
        w_cam.wait()
        for k in range(C):
            g0, g1 = g_at(k), g_at(k + 1)
            if gathers[k] is None:
                continue
            gathers[k].wait()
            ex = self._L.TrainExchange(world, self._recv.data_ptr() + 12 * world * g0, g1 - g0, self._cams_all.data_ptr(), 1.0 / world, self.t,
                                       g0, g1, 0, 0)
            self._call(cam, gt_image, key, 4, ex)
        for k in range(C):
            f0, f1 = f_at(k), f_at(k + 1)
            if reduces[k] is None:
                continue
            reduces[k].wait()
            ex = self._L.TrainExchange(world, None, 0, self._cams_all.data_ptr(), 1.0 / world, self.t, 0, 0, f0, f1)
            self._call(cam, gt_image, key, 8, ex)
        return self.loss_out[0]

    def synchronize(self):
        self._resolve()
        torch.cuda.synchronize(self.dev)
