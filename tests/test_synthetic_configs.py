"""CPU tests of the BASELINE config 3 / config 4 scene generators (sugar_amd/synthetic.py) against the reference's own code:

  * config 4: `make_bound_scene` restates what SuGaR's surface-bound model hands to the rasterizer (sugar_scene/sugar_model.py:
    149-228, 384-397, 399-442, 444-475).  The reference's unmodified `SuGaR` class is built on the same mesh with the same
    in-plane parameters; positions, scales and rotations must agree.
  * config 3: `sh_to_rgb` against the reference's `eval_sh` + 0.5 clamped (sugar_model.py:839-883), `depth_as_colour` against
    the view transform the trainer applies (coarse_sdf.py:575-590).
The first test needs the reference tree (/root/reference or the staged oracle/_ref/pysrc); the others run anywhere."""
import math
import os
import sys
import types

import numpy as np
import pytest
import torch

from sugar_amd import synthetic as syn
from tests import ref_env


def _rotmat(q):
    r, x, y, z = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(-1, 3, 3)


@pytest.mark.skipif(ref_env.reference_root() is None, reason="needs the reference tree")
@pytest.mark.parametrize("n_per_triangle", [1, 6])
def test_bound_scene_is_what_the_reference_model_hands_to_the_rasterizer(n_per_triangle):
    from tests.golden import make_sugar_callsite as mk
    sm = mk._import_reference_model()
    b = syn.make_bound_scene(600 * n_per_triangle, seed=4, n_per_triangle=n_per_triangle)
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        cams = syn.orbit_cameras(64, 48)
        cw = mk._Cameras(cams)
        cw.get_spatial_extent = lambda: 3.3
        nerf = types.SimpleNamespace(device=torch.device("cpu"), training_cameras=cw)
        mesh = mk._TriangleMesh(b.verts.double().numpy(), b.faces.numpy(), np.zeros((b.verts.shape[0], 3)))
        model = sm.SuGaR(nerfmodel=nerf, points=None, colors=None, initialize=False, sh_levels=4, keep_track_of_knn=False,
                         surface_mesh_to_bind=mesh, n_gaussians_per_surface_triangle=n_per_triangle,
                         learn_surface_mesh_positions=True, learn_surface_mesh_opacity=True, learn_surface_mesh_scales=True)
        assert model.binded_to_surface_mesh and model._n_points == b.scene.means3D.shape[0]
        assert abs(float(model.surface_mesh_thickness) - b.thickness) < 1e-12
        with torch.no_grad():
            model._scales.copy_(model.scale_inverse_activation(b.plane_scales))
            model._quaternions.copy_(b.complex_rot)
            pts, sc, q = model.points, model.scaling, model.quaternions
    finally:
        torch.Tensor.cuda = real_cuda
    s = b.scene
    assert torch.allclose(pts, s.means3D, rtol=0, atol=1e-6)
    assert torch.equal(sc[:, 0], s.scales[:, 0])                      # the thin axis is the FIRST one
    assert float((sc[:, 1:] / s.scales[:, 1:] - 1).abs().max()) < 1e-5  # exp(log(x)) of the reference's parameterisation
    # q and -q are one rotation: compare matrices
    assert float((_rotmat(q) - _rotmat(s.rotations)).abs().max()) < 2e-5
    assert float((s.rotations.norm(dim=-1) - 1).abs().max()) < 1e-6


def test_config4_geometry_is_flat_and_on_the_mesh():
    b = syn.make_bound_scene(20_000, seed=4)
    s = b.scene
    P = s.means3D.shape[0]
    assert P == b.faces.shape[0] * b.n_per_triangle and abs(P - 20_000) < 400
    assert float(s.scales[:, 0].max()) == pytest.approx(3.3e-6) and float(s.scales[:, 1:].min()) > 10 * 3.3e-6
    # the thin axis (first column of R) is the face normal
    R = _rotmat(s.rotations)
    fv = b.verts[b.faces]
    n = torch.linalg.cross(fv[:, 1] - fv[:, 0], fv[:, 2] - fv[:, 0], dim=-1)
    n = n / n.norm(dim=-1, keepdim=True)
    assert float((R[:, :, 0] - n).abs().max()) < 1e-4
    # centres lie in their triangle's plane
    assert float(((s.means3D - fv[:, 0]) * n).sum(-1).abs().max()) < 1e-6
    scene, cams, bg = syn.make_config("config4", P=5000)
    assert scene.means3D.shape[0] == 2 * 50 * 50 or scene.means3D.shape[0] > 4000
    assert cams[0].image_width == 1920 and cams[0].image_height == 1080 and float(bg.sum()) == 0.0


def test_config3_colours_match_the_reference_eval_sh():
    """golden vectors written by the reference's own eval_sh / get_points_rgb arithmetic (tests/golden/make_golden.py)"""
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_helpers.npz"))
    got = syn.sh_to_rgb(torch.from_numpy(gold["shs"]), torch.from_numpy(gold["means"]), torch.from_numpy(gold["campos"]))
    assert torch.allclose(got, torch.from_numpy(gold["rgb_deg3"]), rtol=0, atol=2e-6)
    got = syn.sh_to_rgb(torch.from_numpy(gold["prgb_sh_coordinates"]), torch.from_numpy(gold["prgb_positions"]),
                        torch.from_numpy(gold["prgb_camera_center"]).reshape(3))
    assert torch.allclose(got, torch.from_numpy(gold["prgb_colors_l4"]), rtol=0, atol=2e-6)


def test_config3_depth_as_colour():
    scene, cams, _ = syn.make_config("config3", P=4096)
    cam = cams[3]
    pd, bg = syn.depth_as_colour(scene.means3D, cam.viewmatrix)
    w2c = cam.viewmatrix.t().double()
    z = (scene.means3D.double() @ w2c[:3, :3].t() + w2c[:3, 3])[:, 2]
    assert pd.shape == (4096, 3) and torch.equal(pd[:, 0], pd[:, 2])
    assert float((pd[:, 0].double() - z).abs().max()) < 1e-5
    assert torch.equal(bg, pd.max().expand(3)) and float(bg[0]) > 3.0  # far beyond [0,1]: the background is a depth
