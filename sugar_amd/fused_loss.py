"""Fused L1 + D-SSIM photometric loss (HIP, gfx950) with the call shape of the reference's loss helpers.

`l1_ssim_loss(image, gt, lambda_dssim)` equals `(1 - lambda) * l1_loss(image, gt) + lambda * (1 - ssim(image, gt))` of
sugar_utils/loss_utils.py:17-63 (combined at gaussian_splatting/train.py:88-90), forward and backward, in two kernels
instead of ~60.  GPU tensors only; the HIP library must be built (no CPU path here -- sugar_amd.train_step keeps the
stock-PyTorch restatement for CPU use and as the parity reference).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class _L1SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, gt, lambda_dssim):
        if not image.is_cuda:
            raise RuntimeError("fused l1_ssim_loss needs tensors on a ROCm device; there is no CPU fallback")
        lib = _lib.load()
        image = image.contiguous(); gt = gt.contiguous()
        assert image.dtype == torch.float32 and gt.dtype == torch.float32 and image.shape == gt.shape and image.dim() == 3
        Cn, H, W = image.shape
        dev = image.device
        scratch = torch.empty(lib.sgr_l1_ssim_scratch_bytes(Cn, W, H), dtype=torch.uint8, device=dev)
        out = torch.empty(3, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.sgr_l1_ssim_forward(Cn, W, H, C.c_void_p(image.data_ptr()), C.c_void_p(gt.data_ptr()), float(lambda_dssim),
                                         C.c_void_p(scratch.data_ptr()), C.c_void_p(out.data_ptr()),
                                         C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc < 0:
            raise RuntimeError(f"sgr_l1_ssim_forward failed ({rc})")
        ctx.save_for_backward(image, gt, scratch)
        ctx.lambda_dssim = float(lambda_dssim)
        return out[0]

    @staticmethod
    def backward(ctx, grad_loss):
        image, gt, scratch = ctx.saved_tensors
        lib = _lib.load()
        Cn, H, W = image.shape
        dev = image.device
        g = grad_loss.to(dtype=torch.float32, device=dev).reshape(1).contiguous()
        grad_img = torch.empty_like(image)
        with torch.cuda.device(dev):
            rc = lib.sgr_l1_ssim_backward(Cn, W, H, C.c_void_p(image.data_ptr()), C.c_void_p(gt.data_ptr()), ctx.lambda_dssim,
                                          C.c_void_p(scratch.data_ptr()), C.c_void_p(g.data_ptr()),
                                          C.c_void_p(grad_img.data_ptr()),
                                          C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc < 0:
            raise RuntimeError(f"sgr_l1_ssim_backward failed ({rc})")
        return grad_img, None, None


def l1_ssim_loss(image: torch.Tensor, gt: torch.Tensor, lambda_dssim: float = 0.2) -> torch.Tensor:
    return _L1SSIM.apply(image, gt, lambda_dssim)


# ------------------------------------------------------------------------------------------------------------------
# `ssim` with the reference's call shape (sugar_utils/loss_utils.py:39-48 = gaussian_splatting/utils/loss_utils.py): what an
# UNMODIFIED training loop calls once per iteration (train.py:89, coarse_sdf.py:459).  Stock PyTorch runs it as five grouped 11x11
# convolutions and ~25 elementwise kernels plus their autograd twins; here it is the same two kernels as the fused loss with
# lambda = 1 (loss = 1 - mean SSIM).  `sugar_amd.shims.install(patch_losses=True)` puts it in place of the reference's function.
class _SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, gt):
        lib = _lib.load()
        image = image.contiguous(); gt = gt.contiguous()
        Cn, H, W = image.shape
        dev = image.device
        scratch = torch.empty(lib.sgr_l1_ssim_scratch_bytes(Cn, W, H), dtype=torch.uint8, device=dev)
        out = torch.empty(3, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.sgr_l1_ssim_forward(Cn, W, H, C.c_void_p(image.data_ptr()), C.c_void_p(gt.data_ptr()), 1.0,
                                         C.c_void_p(scratch.data_ptr()), C.c_void_p(out.data_ptr()),
                                         C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc < 0:
            raise RuntimeError(f"sgr_l1_ssim_forward failed ({rc})")
        ctx.save_for_backward(image, gt, scratch)
        return out[2]

    @staticmethod
    def backward(ctx, grad_ssim):
        image, gt, scratch = ctx.saved_tensors
        lib = _lib.load()
        Cn, H, W = image.shape
        dev = image.device
        g = (-grad_ssim).to(dtype=torch.float32, device=dev).reshape(1).contiguous()   # loss(lambda = 1) = 1 - ssim
        grad_img = torch.empty_like(image)
        with torch.cuda.device(dev):
            rc = lib.sgr_l1_ssim_backward(Cn, W, H, C.c_void_p(image.data_ptr()), C.c_void_p(gt.data_ptr()), 1.0,
                                          C.c_void_p(scratch.data_ptr()), C.c_void_p(g.data_ptr()),
                                          C.c_void_p(grad_img.data_ptr()),
                                          C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc < 0:
            raise RuntimeError(f"sgr_l1_ssim_backward failed ({rc})")
        return grad_img, None


def ssim_supported(img1, img2, window_size=11, size_average=True) -> bool:
    """the call shape the HIP kernels cover: one float32 image [C,H,W] or [1,C,H,W] on the GPU against a target that needs no
    gradient, the default 11-tap window, the mean over everything"""
    return (torch.is_tensor(img1) and torch.is_tensor(img2) and img1.is_cuda and img2.is_cuda and window_size == 11
            and size_average is True and img1.dtype == torch.float32 and img2.dtype == torch.float32
            and img1.shape == img2.shape and (img1.dim() == 3 or (img1.dim() == 4 and img1.shape[0] == 1))
            and not (img2.requires_grad and torch.is_grad_enabled()))


def make_ssim(original):
    """`ssim(img1, img2, window_size=11, size_average=True)` of loss_utils.py:39-48 on the HIP kernels; any other call shape
    (batches, other windows, per-image means, a target that needs a gradient, CPU tensors) goes to `original`, the caller's own
    function, unchanged."""
    def ssim(img1, img2, window_size=11, size_average=True):
        if not ssim_supported(img1, img2, window_size, size_average):
            return original(img1, img2, window_size, size_average)
        a = img1[0] if img1.dim() == 4 else img1
        b = img2[0] if img2.dim() == 4 else img2
        return _SSIM.apply(a, b.detach())
    ssim._sugar_amd_original = original
    return ssim
