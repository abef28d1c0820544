"""View-sharded data parallelism for ANY model the reference trains, without touching the trainer.

The rasterizer path shards by view: every rank holds a full replica of the model, renders its own camera, and the ranks only meet
in the parameter gradient.  `sugar_amd.train_step` does that for the vanilla-3DGS parameterisation with a compact SH exchange; this
module does it for arbitrary parameter lists -- in particular SuGaR's refine-mode model, whose parameters are the mesh vertices
`_points[n_verts, 3]`, the in-plane scales `_scales[P, 2]`, the in-plane rotation `_quaternions[P, 2]`, `all_densities` and the SH
tensors (sugar_scene/sugar_model.py:222, 326-350; optimised by sugar_trainers/refine.py:786-808 through SuGaROptimizer):

    attach(optimizer)            a step pre-hook: before every `optimizer.step()` the `.grad` of all its parameters are summed over
                                 the ranks in flat buckets (one all-reduce per bucket: RCCL over xGMI with the "nccl" backend) and
                                 divided by the world size -- the mean over the views of the batch, i.e. single-GPU sequential
                                 accumulation of the same views followed by ONE Adam step, the parity target of SURVEY.md 8(e)
    all_reduce_densification_stats(obj)
                                 the densifier's statistics before any topology change (sugar_densifier.py:156-164,
                                 gaussian_model.py:405-407, train.py:111-123): gradient accumulators and denominators are SUMMED,
                                 `max_radii2D` takes the MAXIMUM; every rank then applies the identical clone / split / prune (seeded
                                 identically, sugar_densifier.py:206)
    broadcast_parameters(module_or_params)
                                 replicas start identical

A parameter that received no gradient on this rank's view (a Gaussian outside the frustum of THIS camera has a zero, not a missing,
gradient in the reference too; a whole tensor can be unused, e.g. `_points` with frozen positions) takes part as zeros, so that all
ranks issue the same collectives in the same order.  Works on CPU tensors over gloo (the tests) and on ROCm tensors over RCCL."""
from __future__ import annotations

import torch
import torch.distributed as dist

BUCKET_BYTES = 256 << 20  # one flat fp32 bucket per 256 MB of gradients: few, large collectives (xGMI rings are per-link bound)


def _world(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def _params_of(optimizer):
    opt = getattr(optimizer, "optimizer", optimizer)  # SuGaROptimizer wraps its torch.optim.Adam (sugar_optimizer.py:84)
    return [p for g in opt.param_groups for p in g["params"]]


@torch.no_grad()
def all_reduce_gradients(params, group=None, average=True, bucket_bytes=BUCKET_BYTES):
    """sum (mean) of `.grad` over the ranks, in place, in flat buckets.  A gradient missing on this rank but present on another counts as zeros; one missing on every rank
    stays None."""
    world = _world(group)
    if world == 1:
        return 0
    # (a tensor that neither asks for a gradient nor carries one is frozen on every rank alike; one that carries a hand-assigned
    # `.grad` -- views of a flat gradient buffer, sugar_amd.train_step._torch_adam -- takes part)
    params = [p for p in params if p.requires_grad or p.grad is not None]
    if not params:
        return 0
    # A parameter without a gradient on EVERY rank stays without one (torch.optim.Adam skips it: no step count, no momentum update --
    # what the single-GPU loop does for a tensor that is not in this iteration's graph); one that has a gradient on SOME rank gets
    # zeros on the others.  One small MAX all-reduce of the per-parameter flags decides.
    flags = torch.tensor([0.0 if p.grad is None else 1.0 for p in params], dtype=torch.float32, device=params[0].device)
    dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=group)
    has = flags.cpu().tolist()
    params = [p for p, h in zip(params, has) if h > 0.0]
    n_coll = 1
    i = 0
    while i < len(params):
        bucket, nbytes = [], 0
        dev, dt = params[i].device, params[i].dtype
        while i < len(params) and params[i].device == dev and params[i].dtype == dt and (not bucket or nbytes < bucket_bytes):
            bucket.append(params[i]); nbytes += params[i].numel() * params[i].element_size(); i += 1
        for p in bucket:
            if p.grad is None:
                p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
        if len(bucket) == 1 and bucket[0].grad.is_contiguous():
            flat = bucket[0].grad.view(-1)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            if average:
                flat.mul_(1.0 / world)
        else:
            flat = torch.cat([p.grad.reshape(-1) for p in bucket])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            if average:
                flat.mul_(1.0 / world)
            off = 0
            for p in bucket:
                n = p.numel()
                p.grad.copy_(flat[off: off + n].view_as(p.grad))
                off += n
        n_coll += 1
    return n_coll


def attach(optimizer, group=None, average=True):
    """Registers the gradient exchange as a step pre-hook of `optimizer` (a torch.optim.Optimizer, e.g. the torch.optim.Adam the
    reference constructs, sugar_amd.fused_adam.FusedAdam, or a SuGaROptimizer wrapping one).  Returns the hook handle
    (`handle.remove()` detaches).  With one rank the hook does nothing."""
    opt = getattr(optimizer, "optimizer", optimizer)
    if not isinstance(opt, torch.optim.Optimizer):
        raise TypeError("attach: expected a torch.optim.Optimizer (or an object with an `.optimizer` attribute holding one)")

    def _pre_step(o, args, kwargs):
        all_reduce_gradients([p for g in o.param_groups for p in g["params"]], group=group, average=average)

    return opt.register_step_pre_hook(_pre_step)


_SUM_NAMES = ("xyz_gradient_accum", "points_gradient_accum", "denom")
_MAX_NAMES = ("max_radii2D",)


@torch.no_grad()
def all_reduce_densification_stats(obj, group=None):
    """`obj`: a GaussianModel (gaussian_model.py:125-127), a SuGaRDensifier (sugar_densifier.py:152-154), a NativeTrainer with
    `densify_stats=True`, or a dict -- whatever of `xyz_gradient_accum` / `points_gradient_accum` / `denom` (summed) and
    `max_radii2D` (maximum) it holds.  Call on every rank right before `densify_and_prune`."""
    if _world(group) == 1:
        return
    get = (lambda n: obj.get(n)) if isinstance(obj, dict) else (lambda n: getattr(obj, n, None))
    for names, op in ((_SUM_NAMES, dist.ReduceOp.SUM), (_MAX_NAMES, dist.ReduceOp.MAX)):
        for n in names:
            t = get(n)
            if torch.is_tensor(t):
                if t.is_contiguous():
                    dist.all_reduce(t, op=op, group=group)
                else:
                    c = t.contiguous()
                    dist.all_reduce(c, op=op, group=group)
                    t.copy_(c)


@torch.no_grad()
def broadcast_parameters(module_or_params, src=0, group=None):
    if _world(group) == 1:
        return
    params = module_or_params.parameters() if hasattr(module_or_params, "parameters") else module_or_params
    for p in params:
        dist.broadcast(p.data, src=src, group=group)
