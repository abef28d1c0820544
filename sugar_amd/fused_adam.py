"""`torch.optim.Adam` with its `step()` on the one-launch HIP kernel (csrc/adam.hip: k_adam) -- for the optimisers the reference
builds itself: `GaussianModel.training_setup` (gaussian_splatting/scene/gaussian_model.py:152-166) and `SuGaROptimizer`
(sugar_scene/sugar_optimizer.py:60-85), both `torch.optim.Adam(l, lr=0.0, eps=1e-15)` over six parameter tensors.  Stock PyTorch
runs that step as ~50 multi-tensor kernels (1.5 ms at 1M Gaussians on an MI355X); here it is ONE launch for up to eight parameter
tensors (sgr_adam_step_multi).

`FusedAdam` IS a `torch.optim.Adam`: same constructor, same `param_groups`, same per-parameter state (`step`, `exp_avg`,
`exp_avg_sq` -- what the reference's densifier cuts, concatenates and resets: gaussian_model.py:258-316, sugar_densifier.py), same
`state_dict()`.  Only `step()` differs, and only when every parameter is what the kernel covers (float32, on a ROCm
device, dense gradient, no weight decay / amsgrad / maximize / capturable / differentiable; strided parameters are updated in a
dense copy); anything else is the parent's step.
`adopt(optimizer)` turns an existing `torch.optim.Adam` instance into one in place."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


STATS = {"fused_steps": 0, "fallback_steps": 0, "last_fallback_reason": None}   # (diagnostics: which path the steps took)


def _unsupported_reason(group, p):
    """None when the kernel covers this parameter, else a short reason (the step then is the parent's)"""
    for flag in ("amsgrad", "maximize", "capturable", "differentiable"):
        if group.get(flag, False):
            return flag
    if group.get("weight_decay", 0) != 0:
        return "weight_decay"
    if not p.is_cuda:
        return "parameter not on a ROCm device"
    if p.dtype != torch.float32 or p.grad.dtype != torch.float32:
        return "dtype"
    if p.grad.is_sparse:
        return "sparse gradient"
    return None


def _parent_step(opt, closure):
    """torch.optim.Adam.step WITHOUT its hook wrapper: `FusedAdam.step` is itself the hooked entry point (Optimizer.__init__ /
    `adopt` wrap the class's `step` with `profile_hook_step`), so the fallback must not run the step pre/post hooks a second time --
    a gradient-exchange pre-hook (sugar_amd.view_parallel.attach) would all-reduce twice, and with average=False scale the
    gradients by the world size."""
    f = torch.optim.Adam.step
    if getattr(f, "hooked", False) and hasattr(f, "__wrapped__"):
        f = f.__wrapped__
    return f(opt, closure)


class FusedAdam(torch.optim.Adam):
    @torch.no_grad()
    def step(self, closure=None):
        todo = [(g, p) for g in self.param_groups for p in g["params"] if p.grad is not None]
        reason = "no gradients" if not todo else next((r for r in (_unsupported_reason(g, p) for g, p in todo) if r), None)
        if reason is not None:
            STATS["fallback_steps"] += 1
            STATS["last_fallback_reason"] = reason
            return _parent_step(self, closure)
        STATS["fused_steps"] += 1
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        # Everything a launch can reject is prepared BEFORE any state changes (a step that failed half way would leave some
        # parameters updated and some step counters advanced): dense, 16-byte aligned storage for parameter, gradient and moments.
        # `.contiguous()` returns the SAME storage for a contiguous tensor at an odd offset (a `[1:]` slice): those get a real copy.
        def dense16(t):
            return t if (t.is_contiguous() and t.data_ptr() % 16 == 0) else t.detach().clone(memory_format=torch.contiguous_format)
        plan = []
        for group, p in todo:
            state = self.state[p]
            if len(state) == 0:       # (torch.optim.Adam._init_group)
                state["step"] = torch.tensor(0.0, dtype=torch.float32)
                state["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            # A strided parameter -- SuGaR's `_scales` / `_quaternions` are column slices of one [1, P, 7] tensor until the first
            # pruning (sugar_model.py:313-318) -- is updated in a dense copy and written back.
            dense, grad = dense16(p), dense16(p.grad)
            m, v = dense16(state["exp_avg"]), dense16(state["exp_avg_sq"])
            if any(t.data_ptr() % 16 != 0 for t in (dense, grad, m, v)):
                STATS["fallback_steps"] += 1
                STATS["last_fallback_reason"] = "allocator returned storage that is not 16-byte aligned"
                STATS["fused_steps"] -= 1
                _parent_step(self, None)
                return loss
            plan.append((group, p, state, dense, grad, m, v))
        for group, p, state, dense, grad, m, v in plan:
            state["step"] += 1
            state["exp_avg"], state["exp_avg_sq"] = m, v
        # one launch per 8 tensors of a device (sgr_adam_step_multi: the six tensors of the reference's optimisers are ONE launch)
        by_dev = {}
        for item in plan:
            by_dev.setdefault(item[1].device, []).append(item)
        for dev, items in by_dev.items():
            for k0 in range(0, len(items), 8):
                chunk = items[k0:k0 + 8]
                T = len(chunk)
                ptrs = lambda sel: (C.c_void_p * T)(*[sel(it).data_ptr() for it in chunk])
                with torch.cuda.device(dev):
                    rc = lib.sgr_adam_step_multi(
                        T, (C.c_longlong * T)(*[it[1].numel() for it in chunk]), ptrs(lambda it: it[3]), ptrs(lambda it: it[4]),
                        ptrs(lambda it: it[5]), ptrs(lambda it: it[6]), (C.c_float * T)(*[float(it[0]["lr"]) for it in chunk]),
                        (C.c_float * T)(*[float(it[0]["betas"][0]) for it in chunk]), (C.c_float * T)(*[float(it[0]["betas"][1]) for it in chunk]),
                        (C.c_float * T)(*[float(it[0]["eps"]) for it in chunk]), (C.c_int * T)(*[int(it[2]["step"]) for it in chunk]),
                        C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
                if rc < 0:
                    raise RuntimeError(f"sgr_adam_step_multi failed ({rc})")
        for group, p, state, dense, grad, m, v in plan:
            if dense is not p:
                p.copy_(dense)
        return loss


def adopt(optimizer):
    """an existing `torch.optim.Adam` (exactly that class) becomes a FusedAdam in place: state and param_groups are kept"""
    if type(optimizer) is torch.optim.Adam:
        optimizer.__class__ = FusedAdam
        if hasattr(optimizer, "_patch_step_function"):
            optimizer._patch_step_function()      # step hooks / profiler wrapper of torch.optim.Optimizer for the new class
    return optimizer
