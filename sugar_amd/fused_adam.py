"""`torch.optim.Adam` with its `step()` on the one-launch HIP kernel (csrc/adam.hip: k_adam) -- for the optimisers the reference
builds itself: `GaussianModel.training_setup` (gaussian_splatting/scene/gaussian_model.py:152-166) and `SuGaROptimizer`
(sugar_scene/sugar_optimizer.py:60-85), both `torch.optim.Adam(l, lr=0.0, eps=1e-15)` over six parameter tensors.  Stock PyTorch
runs that step as ~50 multi-tensor kernels (1.5 ms at 1M Gaussians on an MI355X); here it is one launch per parameter tensor.

`FusedAdam` IS a `torch.optim.Adam`: same constructor, same `param_groups`, same per-parameter state (`step`, `exp_avg`,
`exp_avg_sq` -- what the reference's densifier cuts, concatenates and resets: gaussian_model.py:258-316, sugar_densifier.py), same
`state_dict()`.  Only `step()` differs, and only when every parameter is what the kernel covers (float32, contiguous, on a ROCm
device, dense gradient, no weight decay / amsgrad / maximize / capturable / differentiable); anything else is the parent's step.
`adopt(optimizer)` turns an existing `torch.optim.Adam` instance into one in place."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _supported(group, p) -> bool:
    return (not group.get("amsgrad", False) and group.get("weight_decay", 0) == 0 and not group.get("maximize", False)
            and not group.get("capturable", False) and not group.get("differentiable", False)
            and p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad is not None and not p.grad.is_sparse
            and p.grad.dtype == torch.float32 and p.data_ptr() % 16 == 0)


class FusedAdam(torch.optim.Adam):
    @torch.no_grad()
    def step(self, closure=None):
        todo = [(g, p) for g in self.param_groups for p in g["params"] if p.grad is not None]
        if not todo or not all(_supported(g, p) for g, p in todo):
            return super().step(closure)
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        one = (C.c_longlong * 1)
        for group, p in todo:
            state = self.state[p]
            if len(state) == 0:       # (torch.optim.Adam._init_group)
                state["step"] = torch.tensor(0.0, dtype=torch.float32)
                state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            state["step"] += 1
            m, v = state["exp_avg"], state["exp_avg_sq"]
            grad = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            if not (m.is_contiguous() and v.is_contiguous() and m.data_ptr() % 16 == 0 and v.data_ptr() % 16 == 0
                    and grad.data_ptr() % 16 == 0):
                raise RuntimeError("FusedAdam: optimiser state is not contiguous / 16-byte aligned")
            lr = float(group["lr"])
            b1, b2 = group["betas"]
            n = p.numel()
            with torch.cuda.device(p.device):
                rc = lib.sgr_adam_step(n, C.c_void_p(p.data_ptr()), C.c_void_p(grad.data_ptr()), C.c_void_p(m.data_ptr()),
                                       C.c_void_p(v.data_ptr()), 1, one(0), one(n), (C.c_float * 1)(lr), (C.c_float * 1)(lr),
                                       (C.c_int * 1)(1), (C.c_int * 1)(1), float(b1), float(b2), float(group["eps"]),
                                       int(state["step"]), 1.0, C.c_void_p(torch.cuda.current_stream(p.device).cuda_stream))
            if rc < 0:
                raise RuntimeError(f"sgr_adam_step failed ({rc})")
        return loss


def adopt(optimizer):
    """an existing `torch.optim.Adam` (exactly that class) becomes a FusedAdam in place: state and param_groups are kept"""
    if type(optimizer) is torch.optim.Adam:
        optimizer.__class__ = FusedAdam
        if hasattr(optimizer, "_patch_step_function"):
            optimizer._patch_step_function()      # step hooks / profiler wrapper of torch.optim.Optimizer for the new class
    return optimizer
