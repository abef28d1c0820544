"""TEST INFRASTRUCTURE: where the reference's own Python lives.

In the build container the reference tree is /root/reference.  On the GPU box it does not exist; oracle/ref_build/build_ref.sh
stages an UNMODIFIED, git-ignored snapshot of the path's Python (sugar_scene/, sugar_utils/, gaussian_splatting/{scene,utils,
gaussian_renderer,arguments}) next to the compiled reference kernels in oracle/_ref/pysrc, which travels with the snapshot like
oracle/_ref/*.so.  `import_sugar_model()` / `import_gaussian_splatting()` import from whichever exists, with this repository's
drop-in packages (`diff_gaussian_rasterization`, `simple_knn`), the `pytorch3d` / `plyfile` stand-ins on the path and an empty
`open3d` module for an import the path never calls."""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_root():
    for p in (os.environ.get("SUGAR_REFERENCE"), "/root/reference", os.path.join(ROOT, "oracle", "_ref", "pysrc")):
        if p and os.path.isdir(os.path.join(p, "sugar_scene")):
            return p
    return None


def _stubs():
    from sugar_amd import shims
    shims.install()  # pytorch3d and plyfile stand-ins where the real packages are absent
    if "open3d" not in sys.modules:
        try:
            import open3d  # noqa: F401
        except ImportError:
            sys.modules["open3d"] = types.ModuleType("open3d")  # imported by the trainers/extractors, never called on this path


def import_sugar_model(patch_sugar=False):
    """the reference's `sugar_scene.sugar_model`, untouched; `patch_sugar=True` routes its four Gaussian-buffer-sharing methods
    to the HIP kernels (sugar_amd.sugar_patch) the way a user would: shims.install(patch_sugar=...)"""
    ref = reference_root()
    if ref is None:
        raise ImportError("no reference tree: neither /root/reference nor oracle/_ref/pysrc (run oracle/ref_build/build_ref.sh)")
    for p in (os.path.join(ref, "gaussian_splatting"), ref, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    from sugar_amd import shims
    shims.install()
    _stubs()
    import sugar_scene.sugar_model as sm
    assert os.path.abspath(sm.__file__).startswith(os.path.abspath(ref)), sm.__file__
    if patch_sugar:
        shims.install(patch_sugar=sm)
    return sm


def import_gaussian_splatting():
    """(render, GaussianModel, l1_loss, ssim) of the reference's vanilla 3DGS code: gaussian_renderer/__init__.py:18-100,
    scene/gaussian_model.py, utils/loss_utils.py"""
    ref = reference_root()
    if ref is None:
        raise ImportError("no reference tree")
    for p in (os.path.join(ref, "gaussian_splatting"), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    _stubs()
    from gaussian_renderer import render
    from scene.gaussian_model import GaussianModel
    from utils.loss_utils import l1_loss, ssim
    import gaussian_renderer
    assert os.path.abspath(gaussian_renderer.__file__).startswith(os.path.abspath(ref)), gaussian_renderer.__file__
    return render, GaussianModel, l1_loss, ssim
