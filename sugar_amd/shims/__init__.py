"""Stand-ins for the third-party modules the reference imports next to the rasterizer.

`install()` makes `from pytorch3d.ops import knn_points` (sugar_scene/sugar_model.py:7) resolve to the HIP k-NN of this
package:
  * pytorch3d is installed  -> its `ops.knn_points` is replaced by `sugar_amd.knn.knn_points` for the call shape SuGaR uses
                               (everything else stays pytorch3d's own);
  * pytorch3d is absent     -> the minimal `pytorch3d` package under this directory is put on `sys.path`: `ops.knn_points`,
                               `ops.estimate_pointcloud_normals`, the quaternion helpers of `transforms` that SuGaR uses, the
                               camera algebra of `renderer.cameras`, the `structures.Meshes` container and the two
                               `loss` mesh regularisers the surface-bound (refine) model needs, texture containers, and
                               a mesh rasterizer that raises on use -- mesh rasterization / extraction is outside this
                               package's scope.
"""
from __future__ import annotations

import importlib
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


def install(patch_sugar=False, patch_losses=False, patch_optimizer=False, patch_gathers=False, patch_densifier=False) -> str:
    """Returns "patched" (real pytorch3d found, knn_points redirected) or "shim" (stand-in package activated).

    `patch_sugar`: also route SuGaR's own Gaussian-buffer-sharing tensor code -- `get_points_rgb`, `get_covariance(return_sqrt)`,
    `get_field_values`, `compute_level_surface_points_from_camera_fast(use_gaussian_depth=True)` -- to the HIP kernels
    (sugar_amd.sugar_patch), without touching the reference's files.  Pass the imported `sugar_scene.sugar_model` module, or
    True to import it (the reference must then be on sys.path).

    `patch_losses`: also replace the reference's `ssim` by the fused HIP loss kernels (see install_losses).
    `patch_optimizer`: the `torch.optim.Adam` instances the reference builds step on the one-launch HIP Adam (install_optimizer).
    `patch_gathers`: `SuGaR.points / scaling / quaternions / get_normals()` return tensors whose row gathers `x[idx]` have a HIP
    backward (sugar_amd.sugar_patch.install_row_gathers); pass the module like `patch_sugar`, or True.
    `patch_densifier`: the per-iteration densification statistics (`SuGaRDensifier.update_densification_stats`,
    `GaussianModel.add_densification_stats`) without boolean-mask indexing (install_densifier)."""
    real_plyfile = _real_package("plyfile")  # (probed BEFORE the stand-in directory can shadow it on sys.path)
    mode = _install_pytorch3d()
    _install_plyfile(real_plyfile)
    if patch_losses:
        install_losses()
    if patch_optimizer:
        install_optimizer()
    if patch_densifier:
        install_densifier()
    if patch_sugar:
        from .. import sugar_patch
        module = importlib.import_module("sugar_scene.sugar_model") if patch_sugar is True else patch_sugar
        sugar_patch.install(module)
    if patch_gathers:
        from .. import sugar_patch
        module = importlib.import_module("sugar_scene.sugar_model") if patch_gathers is True else patch_gathers
        sugar_patch.install_row_gathers(module)
    return mode


_LOSS_MODULES = ("sugar_utils.loss_utils", "utils.loss_utils")


def install_losses() -> int:
    """`patch_losses`: the reference's `ssim` (sugar_utils/loss_utils.py:39-48, gaussian_splatting/utils/loss_utils.py:39-48) --
    five grouped 11x11 convolutions and ~25 elementwise kernels per call, plus autograd -- is replaced by the HIP pair of
    `sugar_amd.fused_loss` for the call shape the training loops use, in the two modules that define it and in every loaded
    module that had already imported the name (`from sugar_utils.loss_utils import ssim`, coarse_sdf.py:11, train.py:16).
    Returns the number of names rebound.  `uninstall_losses()` undoes it."""
    from ..fused_loss import make_ssim
    originals = {}
    for name in _LOSS_MODULES:
        mod = sys.modules.get(name)
        if mod is None and name.startswith("sugar_utils."):   # (a bare `utils` package could be anybody's: only if already loaded)
            try:
                mod = importlib.import_module(name)
            except ImportError:
                mod = None
        if mod is None:
            continue
        f = getattr(mod, "ssim", None)
        if f is not None and not hasattr(f, "_sugar_amd_original"):
            originals[id(f)] = (f, make_ssim(f))
    count = 0
    for mod in list(sys.modules.values()):
        d = getattr(mod, "__dict__", None)
        if not isinstance(d, dict):
            continue
        f = d.get("ssim")
        if f is not None and id(f) in originals and originals[id(f)][0] is f:
            d["ssim"] = originals[id(f)][1]
            count += 1
    return count


def uninstall_losses() -> int:
    count = 0
    for mod in list(sys.modules.values()):
        d = getattr(mod, "__dict__", None)
        if isinstance(d, dict) and hasattr(d.get("ssim"), "_sugar_amd_original"):
            d["ssim"] = d["ssim"]._sugar_amd_original
            count += 1
    return count


_OPTIMIZER_SITES = (("scene.gaussian_model", "GaussianModel", "training_setup"),        # gaussian_model.py:152-166
                    ("sugar_scene.sugar_optimizer", "SuGaROptimizer", "__init__"))       # sugar_optimizer.py:60-85


def install_optimizer() -> int:
    """`patch_optimizer`: the two places where the reference builds its `torch.optim.Adam(l, lr=0.0, eps=1e-15)` are wrapped so
    that the instance they leave in `self.optimizer` is adopted by `sugar_amd.fused_adam.FusedAdam` -- the same object, state and
    param_groups, a `torch.optim.Adam` subclass whose `step()` is one HIP launch per parameter tensor instead of ~50 multi-tensor
    kernels.  Only modules that are already imported (or `sugar_scene.*`, importable by name) are touched.  Returns the number of
    sites wrapped; `uninstall_optimizer()` undoes it."""
    import functools
    from ..fused_adam import adopt
    count = 0
    for modname, clsname, method in _OPTIMIZER_SITES:
        mod = sys.modules.get(modname)
        if mod is None and modname.startswith("sugar_scene."):
            try:
                mod = importlib.import_module(modname)
            except ImportError:
                mod = None
        cls = getattr(mod, clsname, None) if mod is not None else None
        if cls is None or hasattr(cls.__dict__.get(method), "_sugar_amd_original"):
            continue
        original = getattr(cls, method)

        def make(original):
            @functools.wraps(original)
            def wrapped(self, *a, **k):
                out = original(self, *a, **k)
                if getattr(self, "optimizer", None) is not None:
                    adopt(self.optimizer)
                return out
            wrapped._sugar_amd_original = original
            return wrapped
        setattr(cls, method, make(original))
        count += 1
    return count


def uninstall_optimizer() -> int:
    count = 0
    for modname, clsname, method in _OPTIMIZER_SITES:
        cls = getattr(sys.modules.get(modname), clsname, None)
        f = cls.__dict__.get(method) if cls is not None else None
        if hasattr(f, "_sugar_amd_original"):
            setattr(cls, method, f._sugar_amd_original)
            count += 1
    return count


# ---- densification statistics (sugar_densifier.py:156-164, gaussian_model.py:405-407)
def _is_mask(t, n):
    import torch
    return torch.is_tensor(t) and t.dtype == torch.bool and t.dim() == 1 and t.shape[0] == n


def _update_densification_stats(self, viewspace_point_tensor, radii, visibility_filter, _orig=None):
    """sugar_densifier.py:156-164 for a boolean `visibility_filter`, as full-length masked updates: `x[mask] op= y[mask]` costs a
    `nonzero` (a device-to-host round trip), a gather and an index_put per statement -- ~0.7 ms per iteration at 1M Gaussians on an
    MI355X, a fifth of a coarse iteration before the SDF phase.  Same values: rows outside the mask are left as they are."""
    import torch
    g = viewspace_point_tensor.grad
    if not _is_mask(visibility_filter, self.max_radii2D.shape[0]) or g is None:
        return _orig(self, viewspace_point_tensor, radii, visibility_filter)
    vis = visibility_filter
    r = radii.to(self.max_radii2D.dtype)
    self.max_radii2D.copy_(torch.where(vis, torch.max(self.max_radii2D, r), self.max_radii2D))
    gn = torch.norm(g[:, :2], dim=-1, keepdim=True)
    self.points_gradient_accum.add_(torch.where(vis[:, None], gn, torch.zeros_like(gn)))
    self.denom.add_(vis[:, None].to(self.denom.dtype))


def _add_densification_stats(self, viewspace_point_tensor, update_filter, _orig=None):
    """gaussian_model.py:405-407, same idea"""
    import torch
    g = viewspace_point_tensor.grad
    if not _is_mask(update_filter, self.denom.shape[0]) or g is None:
        return _orig(self, viewspace_point_tensor, update_filter)
    gn = torch.norm(g[:, :2], dim=-1, keepdim=True)
    self.xyz_gradient_accum.add_(torch.where(update_filter[:, None], gn, torch.zeros_like(gn)))
    self.denom.add_(update_filter[:, None].to(self.denom.dtype))


_DENSIFIER_SITES = (("sugar_scene.sugar_densifier", "SuGaRDensifier", "update_densification_stats", _update_densification_stats),
                    ("scene.gaussian_model", "GaussianModel", "add_densification_stats", _add_densification_stats))


def install_densifier() -> int:
    """`patch_densifier`: the two statistics methods the trainers call every iteration are rebound to mask-free equivalents (see
    `_update_densification_stats`); an index-tensor filter, or anything else the replacement does not cover, goes to the original.
    (The `max_radii2D[visibility_filter] = ...` line of the vanilla loop, train.py:114, is inline trainer code and stays as it is.)
    Returns the number of methods rebound; `uninstall_densifier()` undoes it."""
    import functools
    count = 0
    for modname, clsname, method, impl in _DENSIFIER_SITES:
        mod = sys.modules.get(modname)
        if mod is None and modname.startswith("sugar_scene."):
            try:
                mod = importlib.import_module(modname)
            except ImportError:
                mod = None
        cls = getattr(mod, clsname, None) if mod is not None else None
        if cls is None or hasattr(cls.__dict__.get(method), "_sugar_amd_original"):
            continue
        original = getattr(cls, method)
        wrapped = functools.wraps(original)(functools.partial(impl, _orig=original))

        def make(f, original):
            def bound(self, *a, **k):
                return f(self, *a, **k)
            bound.__name__, bound.__doc__ = original.__name__, f.func.__doc__
            bound._sugar_amd_original = original
            return bound
        setattr(cls, method, make(wrapped, original))
        count += 1
    return count


def uninstall_densifier() -> int:
    count = 0
    for modname, clsname, method, _impl in _DENSIFIER_SITES:
        cls = getattr(sys.modules.get(modname), clsname, None)
        f = cls.__dict__.get(method) if cls is not None else None
        if hasattr(f, "_sugar_amd_original"):
            setattr(cls, method, f._sugar_amd_original)
            count += 1
    return count


def _real_package(name: str):
    """the module spec of an installed `name` that is NOT the stand-in under this directory, or None"""
    try:
        spec = importlib.util.find_spec(name)
    except (ImportError, ValueError):
        return None
    if spec is None or spec.origin is None or os.path.abspath(spec.origin).startswith(_HERE):
        return None
    return spec


def _install_plyfile(real_spec=None) -> None:
    """`plyfile` (gaussian_model.py:18, dataset_readers.py:22) is not in the ROCm image: when no real package exists, the stand-in
    under this directory (vertex elements of binary little-endian PLY files only) takes the name.  A real `plyfile` always wins:
    this directory also holds the `pytorch3d` stand-in and may sit at the front of sys.path on a box that has plyfile but no
    pytorch3d, so the real module is imported from its own location and pinned in sys.modules."""
    if real_spec is not None:
        mod = sys.modules.get("plyfile")
        if mod is None or os.path.abspath(getattr(mod, "__file__", "") or "").startswith(_HERE):
            mod = importlib.util.module_from_spec(real_spec)
            sys.modules["plyfile"] = mod
            real_spec.loader.exec_module(mod)
        return
    if "plyfile" in sys.modules:
        return
    if _HERE not in sys.path:
        sys.path.insert(0, _HERE)
    importlib.invalidate_caches()


def _install_pytorch3d() -> str:
    spec = None
    try:
        spec = importlib.util.find_spec("pytorch3d")
    except (ImportError, ValueError):
        spec = None
    real = spec is not None and spec.origin is not None and not os.path.abspath(spec.origin).startswith(_HERE)
    if real:
        import pytorch3d.ops as p3d_ops
        from ..knn import knn_points_pytorch3d
        if not hasattr(p3d_ops, "_sugar_amd_original_knn_points"):
            p3d_ops._sugar_amd_original_knn_points = p3d_ops.knn_points
        p3d_ops.knn_points = knn_points_pytorch3d(p3d_ops._sugar_amd_original_knn_points)
        return "patched"
    if _HERE not in sys.path:
        sys.path.insert(0, _HERE)
    importlib.invalidate_caches()
    importlib.import_module("pytorch3d")
    return "shim"
