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
