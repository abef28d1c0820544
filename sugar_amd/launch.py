"""Run one of the reference's scripts UNMODIFIED on this stack:

    python -m sugar_amd.launch [options] /path/to/SuGaR/train.py -s <scene> ...
    python -m sugar_amd.launch [options] /path/to/SuGaR/gaussian_splatting/train.py -s <scene> ...

What it does before handing over to the script (`runpy.run_path(..., run_name="__main__")`, `sys.argv` = the script's own):
  * puts this repository first on `sys.path`, so that `diff_gaussian_rasterization` and `simple_knn` resolve to the HIP drop-ins;
  * `sugar_amd.shims.install()`: the `pytorch3d` / `plyfile` stand-ins where the real packages are absent (an installed pytorch3d
    only gets its `knn_points` redirected);
  * the opt-in bindings, all on by default here: SuGaR's field / sampler methods (`--no-patch-sugar`), the reference's `ssim`
    (`--no-patch-losses`), the optimisers it constructs (`--no-patch-optimizer`), the row gathers of its per-Gaussian tensors
    (`--no-patch-gathers`), the densification statistics without boolean-mask indexing (`--no-patch-densifier`).  They need the reference's modules importable: the script's directory (and its `gaussian_splatting/`
    sub-directory, which the reference itself appends to `sys.path`) are added the way `python script.py` would.
Nothing under the reference's tree is written or edited.  `open3d` is NOT provided: the mesh-extraction scripts need the real one."""
from __future__ import annotations

import argparse
import importlib
import importlib.util
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def prepare(script: str, patch_sugar=True, patch_losses=True, patch_optimizer=True, patch_gathers=True, patch_densifier=True) -> dict:
    """everything `main` does except running the script; returns what was bound (for logging and tests)"""
    script_dir = os.path.dirname(os.path.abspath(script))
    for p in (os.path.join(script_dir, "gaussian_splatting"), script_dir, ROOT):
        if os.path.isdir(p):
            while p in sys.path:
                sys.path.remove(p)
            sys.path.insert(0, p)
    from sugar_amd import shims
    done = {"pytorch3d": shims.install(), "patch_sugar": False, "patch_gathers": False, "patch_losses": 0, "patch_optimizer": 0,
            "patch_densifier": 0}
    sm = None
    if patch_sugar or patch_gathers:
        try:
            sm = importlib.import_module("sugar_scene.sugar_model")
        except ImportError:
            sm = None    # (e.g. gaussian_splatting/train.py: vanilla 3DGS has no SuGaR model)
    if sm is not None:
        shims.install(patch_sugar=sm if patch_sugar else False, patch_gathers=sm if patch_gathers else False)
        done["patch_sugar"], done["patch_gathers"] = bool(patch_sugar), bool(patch_gathers)
    for name in ("utils.loss_utils", "scene.gaussian_model", "sugar_scene.sugar_densifier"):      # vanilla 3DGS modules: bound only once they are loaded
        try:
            if importlib.util.find_spec(name) is not None and os.path.abspath(importlib.util.find_spec(name).origin).startswith(script_dir):
                importlib.import_module(name)
        except (ImportError, ValueError, AttributeError):
            pass
    if patch_losses:
        done["patch_losses"] = shims.install_losses()
    if patch_optimizer:
        done["patch_optimizer"] = shims.install_optimizer()
    if patch_densifier:
        done["patch_densifier"] = shims.install_densifier()
    return done


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m sugar_amd.launch", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    for flag in ("sugar", "losses", "optimizer", "gathers", "densifier"):
        ap.add_argument(f"--no-patch-{flag}", action="store_true")
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("script")
    ap.add_argument("script_args", nargs=argparse.REMAINDER)
    a = ap.parse_args(argv)
    if not os.path.isfile(a.script):
        ap.error(f"no such script: {a.script}")
    done = prepare(a.script, not a.no_patch_sugar, not a.no_patch_losses, not a.no_patch_optimizer, not a.no_patch_gathers,
                   not a.no_patch_densifier)
    if not a.quiet:
        print(f"[sugar_amd.launch] {done}", file=sys.stderr)
    sys.argv = [a.script] + list(a.script_args)
    runpy.run_path(a.script, run_name="__main__")


if __name__ == "__main__":
    main()
