"""Densification for the native trainer: the statistics and the clone / split / prune step of 3DGS as plain tensor functions,
so that a view-sharded `NativeTrainer` can change its topology identically on every rank.

Restates gaussian_splatting/scene/gaussian_model.py:
    add_densification_stats   :405-407   (the accumulation itself is fused into the rasterizer backward: sgr_backward_opts)
    densify_and_clone         :376-388
    densify_and_split         :350-374   (N = 2 samples per split Gaussian, new scales / (0.8 N), `torch.normal` draws)
    densify_and_prune         :390-403   (opacity below the threshold, screen radius above `max_screen_size`, world size > 0.1 extent)
    the optimiser bookkeeping :258-316   (moments of appended Gaussians are zero, pruned rows are dropped)
and sugar_scene/sugar_densifier.py:156-244, which is the same procedure on SuGaR's tensors.

The random draws of the split come from a `torch.Generator` the caller seeds identically on every rank (sugar_densifier.py:206 uses
the global generator: same effect under a common seed).  Everything is deterministic given the statistics, so ranks that all-reduced
their statistics (sugar_amd.view_parallel.all_reduce_densification_stats) stay bit-identical replicas.  Device-agnostic torch code:
runs on CPU tensors in the gloo tests and on ROCm tensors in the trainer."""
from __future__ import annotations

import torch

NAMES = ("xyz", "opacity", "scaling", "rotation", "features")


def _build_rotation(r: torch.Tensor) -> torch.Tensor:
    """gaussian_splatting/utils/general_utils.py:78-101 (normalises the quaternion, real part first)"""
    norm = torch.sqrt(r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1] + r[:, 2] * r[:, 2] + r[:, 3] * r[:, 3])  # (the reference's own sum order)
    q = r / norm[:, None]
    w, x, y, z = q.unbind(-1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.reshape(-1, 3, 3)


def _append(t, m1, m2, new):
    """cat_tensors_to_optimizer (:302-326): parameters grow by `new`, both moments by zeros"""
    return ({k: torch.cat((t[k], new[k]), dim=0) for k in NAMES},
            {k: torch.cat((m1[k], torch.zeros_like(new[k])), dim=0) for k in NAMES},
            {k: torch.cat((m2[k], torch.zeros_like(new[k])), dim=0) for k in NAMES})


def _keep(t, m1, m2, keep):
    """_prune_optimizer (:272-288)"""
    return ({k: t[k][keep] for k in NAMES}, {k: m1[k][keep] for k in NAMES}, {k: m2[k][keep] for k in NAMES})


@torch.no_grad()
def densify_and_prune(tensors, exp_avg, exp_avg_sq, stats, *, max_grad, min_opacity, extent, max_screen_size, percent_dense=0.01,
                      n_split=2, generator=None):
    """`tensors` / `exp_avg` / `exp_avg_sq`: dicts of the RAW parameters and Adam moments by name (xyz [P,3], opacity [P,1],
    scaling [P,3], rotation [P,4], features [P,M,3]); `stats`: dict(xyz_gradient_accum[P], denom[P], max_radii2D[P]).
    Returns (tensors, exp_avg, exp_avg_sq, n_cloned, n_split, n_pruned) for the new topology; the statistics start from zero
    afterwards (densification_postfix, :343-345)."""
    t, m1, m2 = dict(tensors), dict(exp_avg), dict(exp_avg_sq)
    grads = stats["xyz_gradient_accum"].reshape(-1) / stats["denom"].reshape(-1)
    grads[grads.isnan()] = 0.0
    scal = lambda tt: torch.exp(tt["scaling"])
    # ---- clone (:376-388): small Gaussians with a large positional gradient are duplicated in place
    sel = (grads >= max_grad) & (scal(t).max(dim=1).values <= percent_dense * extent)
    n_cloned = int(sel.sum())
    t, m1, m2 = _append(t, m1, m2, {k: t[k][sel] for k in NAMES})
    # ---- split (:350-374): large ones are replaced by N samples of themselves (the gradients of the clones just appended count as zero)
    P_now = t["xyz"].shape[0]
    padded = torch.zeros(P_now, dtype=grads.dtype, device=grads.device)
    padded[: grads.shape[0]] = grads
    sel = (padded >= max_grad) & (scal(t).max(dim=1).values > percent_dense * extent)
    n_sel = int(sel.sum())
    stds = scal(t)[sel].repeat(n_split, 1)
    samples = torch.normal(mean=torch.zeros_like(stds), std=stds, generator=generator) if n_sel else stds
    rots = _build_rotation(t["rotation"][sel]).repeat(n_split, 1, 1)
    new = {"xyz": torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + t["xyz"][sel].repeat(n_split, 1),
           "scaling": torch.log(scal(t)[sel].repeat(n_split, 1) / (0.8 * n_split)),
           "rotation": t["rotation"][sel].repeat(n_split, 1), "features": t["features"][sel].repeat(n_split, 1, 1),
           "opacity": t["opacity"][sel].repeat(n_split, 1)}
    t, m1, m2 = _append(t, m1, m2, new)
    drop = torch.cat((sel, torch.zeros(n_split * n_sel, dtype=torch.bool, device=sel.device)))
    t, m1, m2 = _keep(t, m1, m2, ~drop)
    # ---- prune (:395-401): transparent, and (with a screen-size limit) too large on screen or in the world
    prune = (torch.sigmoid(t["opacity"]) < min_opacity).reshape(-1)
    if max_screen_size:
        radii = torch.zeros(t["xyz"].shape[0], dtype=stats["max_radii2D"].dtype, device=prune.device)  # (postfix zeroed the statistics)
        big_vs = radii > max_screen_size
        big_ws = scal(t).max(dim=1).values > 0.1 * extent
        prune = prune | big_vs | big_ws
    n_pruned = int(prune.sum())
    t, m1, m2 = _keep(t, m1, m2, ~prune)
    return ({k: v.contiguous() for k, v in t.items()}, {k: v.contiguous() for k, v in m1.items()},
            {k: v.contiguous() for k, v in m2.items()}, n_cloned, n_sel, n_pruned)
