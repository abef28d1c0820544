"""CPU tests of the third-party stand-ins (sugar_amd/shims) and of the drop-in at the reference's real call site:
the unmodified sugar_scene/sugar_model.py imports against this repository's packages and renders through the boundary."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

REF = "/root/reference"
GOLD = os.path.join(os.path.dirname(__file__), "golden", "sugar_callsite.npz")


@pytest.fixture(scope="module")
def p3d():
    from sugar_amd import shims
    mode = shims.install()
    assert mode in ("shim", "patched")
    assert shims.install() == mode  # idempotent
    import pytorch3d
    return pytorch3d


def test_quaternion_helpers_match_scipy(p3d):
    from scipy.spatial.transform import Rotation
    T = importlib.import_module("pytorch3d.transforms")
    g = torch.Generator().manual_seed(0)
    q = torch.randn(2000, 4, generator=g, dtype=torch.float64)
    q = q / q.norm(dim=-1, keepdim=True)
    rot = Rotation.from_quat(q[:, [1, 2, 3, 0]].numpy())  # scipy is scalar-last
    M = T.quaternion_to_matrix(q)
    assert np.abs(M.numpy() - rot.as_matrix()).max() < 1e-13
    # un-normalised quaternions describe the same rotation (two_s = 2 / |q|^2)
    assert np.abs(T.quaternion_to_matrix(3.0 * q).numpy() - rot.as_matrix()).max() < 1e-13
    q2 = T.matrix_to_quaternion(M)
    err = torch.minimum((q2 - q).abs().amax(-1), (q2 + q).abs().amax(-1))
    assert float(err.max()) < 1e-12
    v = torch.randn(2000, 3, generator=g, dtype=torch.float64)
    assert np.abs(T.quaternion_apply(q, v).numpy() - rot.apply(v.numpy())).max() < 1e-12
    back = T.quaternion_apply(T.quaternion_invert(q), T.quaternion_apply(q, v))
    assert float((back - v).abs().max()) < 1e-12
    # broadcasting as SuGaR uses it (sugar_model.py:509, 1095-1097): [P,1,4] applied to [P,n,3]
    out = T.quaternion_apply(q[:, None], v[:, None].expand(-1, 3, -1))
    assert out.shape == (2000, 3, 3)
    ident = T.matrix_to_quaternion(torch.eye(3)[None, None].repeat(1, 5, 1, 1))  # sugar_model.py:77-79
    assert torch.equal(ident, torch.tensor([1.0, 0, 0, 0]).expand(1, 5, 4))


def test_out_of_scope_classes_import_and_raise(p3d):
    if getattr(p3d, "__version__", "").endswith("sugar_amd.shim"):
        from pytorch3d.renderer import MeshRasterizer, RasterizationSettings, TexturesUV, TexturesVertex  # noqa: F401
        from pytorch3d.structures import Meshes
        from pytorch3d.renderer.cameras import _get_sfm_calibration_matrix
        with pytest.raises(NotImplementedError):
            Meshes(verts=[], faces=[])
        with pytest.raises(NotImplementedError):
            _get_sfm_calibration_matrix(1, "cpu", None, None)
        from pytorch3d.ops import knn_points
        with pytest.raises(RuntimeError):  # HIP only, no CPU fallback
            knn_points(torch.zeros(1, 10, 3), torch.zeros(1, 10, 3), K=4)


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")
def test_unmodified_sugar_model_renders_through_the_boundary(p3d):
    """SuGaR (imported from the reference, untouched) -> render_image_gaussian_rasterizer -> GaussianRasterizer boundary
    (CPU oracle behind it here) reproduces the committed fixture; the GPU test replays the same boundary tensors."""
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_sugar_callsite as mk
    out = mk.run()
    gold = np.load(GOLD)
    assert set(out) == set(gold.files)
    for k in gold.files:
        a, b = np.asarray(out[k]), gold[k]
        assert a.shape == b.shape, k
        if a.dtype.kind in "iu" or "_in_" in k or k in ("W", "H"):
            assert np.array_equal(a, b), k
        else:
            np.testing.assert_allclose(a, b, rtol=2e-4, atol=1e-6 * max(1.0, float(np.abs(b).max())), err_msg=k)
    sm = sys.modules["sugar_scene.sugar_model"]
    assert os.path.abspath(sm.__file__).startswith(REF)
    from sugar_amd.diff_gaussian_rasterization import GaussianRasterizationSettings
    assert sm.GaussianRasterizationSettings is GaussianRasterizationSettings
    import simple_knn._C as knn_c
    assert sm.distCUDA2 is knn_c.distCUDA2
