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
        from pytorch3d.structures import Pointclouds
        from pytorch3d.renderer.cameras import _get_sfm_calibration_matrix
        with pytest.raises(NotImplementedError):
            Pointclouds(points=[])
        with pytest.raises(NotImplementedError):  # a texture atlas is a container here; sampling it needs mesh fragments
            TexturesUV(maps=torch.zeros(1, 4, 4, 3), faces_uvs=[torch.zeros(1, 3, dtype=torch.long)],
                       verts_uvs=[torch.zeros(3, 2)]).sample_textures(None)
        # the camera algebra is functional (sugar_scene/cameras.py:311-324 builds its cameras from it) ...
        K = _get_sfm_calibration_matrix(1, "cpu", torch.tensor([[1.5, 2.0]]), torch.tensor([[0.1, -0.2]]))
        assert K.shape == (1, 4, 4) and float(K[0, 0, 0]) == 1.5 and abs(float(K[0, 1, 2]) + 0.2) < 1e-6 and float(K[0, 3, 2]) == 1.0
        # ... and the mesh rasterizer (SuGaR constructs it unconditionally, sugar_model.py:1880-1893) runs on the HIP z-buffer only:
        # CPU tensors raise (tests/test_mesh_oracle.py drives its host code on the oracle backend, tests/test_gpu_mesh_raster.py the kernel)
        from pytorch3d.renderer import FoVPerspectiveCameras
        from pytorch3d.structures import Meshes
        r = MeshRasterizer(cameras=FoVPerspectiveCameras(), raster_settings=RasterizationSettings(image_size=(8, 8), faces_per_pixel=10))
        tri = Meshes(verts=[torch.tensor([[0., 0., 2.], [1., 0., 2.], [0., 1., 2.]])], faces=[torch.tensor([[0, 1, 2]])])
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            r(tri)
        with pytest.raises(ValueError):
            MeshRasterizer(cameras=None)(tri)
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


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")
def test_unmodified_sugar_model_bound_to_a_surface_mesh_renders_through_the_boundary(p3d):
    """The refine-mode model (BASELINE.json config 4): SuGaR(surface_mesh_to_bind=...) from the reference, untouched, builds on
    the stand-in `Meshes` / `TexturesVertex`, derives its flat Gaussians from the mesh and renders through the boundary; the
    committed fixture (replayed on the GPU by tests/test_gpu_sugar_callsite.py) is reproduced, gradients on the mesh vertices
    and the in-plane scale / rotation parameters included."""
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_sugar_callsite as mk
    out = mk.run_bound()
    gold = np.load(GOLD.replace(".npz", "_bound.npz"))
    assert set(out) == set(gold.files)
    assert out["c0_param_grad_points"].shape[0] < out["c0_in_means3D"].shape[0] == 6 * int(out["n_faces"])
    for k in gold.files:
        a, b = np.asarray(out[k]), gold[k]
        assert a.shape == b.shape, k
        if a.dtype.kind in "iu" or "_in_" in k or k in ("W", "H"):
            assert np.array_equal(a, b), k
        else:
            np.testing.assert_allclose(a, b, rtol=2e-4, atol=1e-6 * max(1.0, float(np.abs(b).max())), err_msg=k)


def test_mesh_container_and_mesh_regularisers_closed_form(p3d):
    """The stand-in `Meshes` and `pytorch3d.loss` (parity-unpinned against pytorch3d, which is absent) on meshes whose values
    are known in closed form: a cube split into 12 triangles has 18 edges, 12 of them between perpendicular faces
    (normal consistency 12/18); a flat regular grid has zero normal-consistency loss and a uniform Laplacian that vanishes at
    interior vertices."""
    if not getattr(p3d, "__version__", "").endswith("sugar_amd.shim"):
        pytest.skip("real pytorch3d installed")
    from pytorch3d.loss import mesh_laplacian_smoothing, mesh_normal_consistency
    from pytorch3d.renderer import TexturesVertex
    from pytorch3d.structures import Meshes
    v = torch.tensor([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]],
                     dtype=torch.float32, requires_grad=True)
    f = torch.tensor([[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4], [1, 2, 6], [1, 6, 5], [2, 3, 7], [2, 7, 6],
                      [3, 0, 4], [3, 4, 7]])
    cube = Meshes(verts=[v], faces=[f], textures=TexturesVertex(verts_features=torch.rand(1, 8, 3)))
    assert cube.edges_packed().shape == (18, 2) and bool((cube.edges_packed()[:, 0] < cube.edges_packed()[:, 1]).all())
    n = cube.faces_normals_list()[0]
    centre = v.detach()[f].mean(1) - 0.5
    assert torch.allclose(n.norm(dim=1), torch.ones(12)) and bool(((n * centre).sum(1) > 0).all())  # unit, outward
    assert cube.verts_list()[0] is v and torch.equal(cube.faces_list()[0], f)
    nc = mesh_normal_consistency(cube)
    assert abs(float(nc) - 12 / 18) < 1e-6
    # uniform Laplacian by hand: mean of the neighbours minus the vertex
    e = cube.edges_packed()
    nb = [[] for _ in range(8)]
    for a, b in e.tolist():
        nb[a].append(b); nb[b].append(a)
    want = np.mean([np.linalg.norm(v.detach().numpy()[nb[i]].mean(0) - v.detach().numpy()[i]) for i in range(8)])
    lap = mesh_laplacian_smoothing(cube, method="uniform")
    assert abs(float(lap) - want) < 1e-6
    (nc + lap).backward()
    assert torch.isfinite(v.grad).all() and float(v.grad.abs().sum()) > 0
    with pytest.raises(NotImplementedError):
        mesh_laplacian_smoothing(cube, method="cot")
    # a flat 5x5 grid, two meshes in one batch
    ys, xs = torch.meshgrid(torch.arange(5.0), torch.arange(5.0), indexing="ij")
    gv = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.zeros(25)], dim=1)
    gf = []
    for y in range(4):
        for x in range(4):
            a = y * 5 + x
            gf += [[a, a + 1, a + 6], [a, a + 6, a + 5]]
    gf = torch.tensor(gf)
    flat = Meshes(verts=[gv, gv * 2.0], faces=[gf, gf])
    assert float(mesh_normal_consistency(flat)) < 1e-6
    assert flat.faces_packed().max() == 49 and flat.edges_packed().shape[0] == 2 * (2 * 4 * 5 + 16)
    # bending the grid makes both losses positive
    bent = gv.clone(); bent[:, 2] = 0.3 * (bent[:, 0] - 2.0) ** 2
    assert float(mesh_normal_consistency(Meshes([bent], [gf]))) > 1e-3
    assert float(mesh_laplacian_smoothing(Meshes([bent], [gf]))) > float(mesh_laplacian_smoothing(Meshes([gv], [gf])))


def test_camera_algebra_round_trips_and_matches_the_gaussian_splatting_camera(p3d):
    """The stand-in FoVPerspectiveCameras (parity-unpinned against pytorch3d itself, which is absent): unproject(project(x))
    = x, the camera centre is where the view transform maps to the origin, and a camera built the way
    sugar_scene/cameras.py:convert_camera_from_gs_to_pytorch3d builds it sees a world point at the pixel the rasterizer's
    viewmatrix / projmatrix (synthetic.look_at_camera, a restatement of sugar_scene/cameras.py:203-212) puts it."""
    if not getattr(p3d, "__version__", "").endswith("sugar_amd.shim"):
        pytest.skip("real pytorch3d installed")
    from pytorch3d.renderer import FoVPerspectiveCameras
    from pytorch3d.renderer.cameras import _get_sfm_calibration_matrix
    from sugar_amd import synthetic as syn
    W, H = 200, 152
    cam = syn.orbit_cameras(W, H)[3]
    w2c = cam.viewmatrix.t().double()                       # COLMAP-convention world-to-camera (x right, y down, z forward)
    fx, fy = W / (2 * cam.tanfovx), H / (2 * cam.tanfovy)
    scale = min(W, H) / 2.0
    K = _get_sfm_calibration_matrix(1, "cpu", torch.tensor([[fx / scale, fy / scale]]), torch.zeros(1, 2))
    # cameras.py:319-322: pytorch3d looks along +z with x LEFT and y UP: flip x and y of the COLMAP camera frame
    flip = torch.tensor([-1.0, -1.0, 1.0], dtype=torch.float64)
    R = (w2c[:3, :3].t() * flip)[None].float()              # row-vector convention: X_view = X_world @ R + T
    T = (w2c[:3, 3] * flip)[None].float()
    p3 = FoVPerspectiveCameras(R=R, T=T, K=K, znear=0.0001)
    assert torch.allclose(p3.get_camera_center()[0], cam.campos, atol=1e-5)
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(50, 3, generator=g) - 0.5)
    view = p3.get_world_to_view_transform().transform_points(x)
    ndc = p3.transform_points(x)
    back = p3.unproject_points(torch.cat([ndc[:, :2], view[:, 2:3]], dim=1)[None], scaled_depth_input=False)[0]
    assert torch.allclose(back, x, atol=1e-5)
    # the Gaussian-splatting projection of the same points: pixel = ((ndc + 1) * S - 1) / 2 (auxiliary.h:41-44)
    hom = torch.cat([x, torch.ones(50, 1)], dim=1) @ cam.projmatrix
    gs = hom[:, :2] / hom[:, 3:4]
    px_gs = ((gs[:, 0] + 1) * W - 1) / 2
    py_gs = ((gs[:, 1] + 1) * H - 1) / 2
    # pytorch3d NDC: +x left, +y up, the SHORTER side spans [-1, 1] (sugar_model.py:1937-1944 maps pixel j to
    # W / min - 2 j / (min - 1))
    px_p3 = (W / min(W, H) - ndc[:, 0]) * (min(W, H) - 1) / 2
    py_p3 = (H / min(W, H) - ndc[:, 1]) * (min(W, H) - 1) / 2
    # the two pixel conventions differ by the (min - 1) / min factor SuGaR's table uses: within a pixel over the image
    assert float((px_p3 - px_gs).abs().max()) < 1.0 and float((py_p3 - py_gs).abs().max()) < 1.0


def test_patch_losses_rebinds_ssim_everywhere_and_cpu_calls_reach_the_original():
    """shims.install(patch_losses=True): the name `ssim` is rebound in the defining module and in modules that imported it
    earlier; a call the HIP kernels do not cover (here: CPU tensors) is answered by the reference's own function"""
    import types as _types
    from tests import ref_env
    if ref_env.reference_root() is None:
        pytest.skip("no reference tree")
    ref_env.import_sugar_model()
    import sugar_utils.loss_utils as lu
    from sugar_amd import shims
    shims.uninstall_losses()
    original = lu.ssim
    early = _types.ModuleType("early_importer"); early.ssim = lu.ssim
    sys.modules["early_importer"] = early
    try:
        n = shims.install_losses()
        assert n >= 2 and lu.ssim is not original and early.ssim is lu.ssim and lu.ssim._sugar_amd_original is original
        assert shims.install_losses() == 0   # idempotent
        g = torch.Generator().manual_seed(0)
        a = torch.rand(3, 40, 52, generator=g, requires_grad=True); b = torch.rand(3, 40, 52, generator=g)
        v = lu.ssim(a, b)
        assert torch.equal(v, original(a, b))
        v.backward()
        assert a.grad is not None
        per_image = lu.ssim(a[None].detach(), b[None], size_average=False)      # other call shapes: the original's answer
        assert per_image.shape == (1,)
    finally:
        assert shims.uninstall_losses() >= 2
        del sys.modules["early_importer"]
    assert lu.ssim is original


def test_patch_optimizer_adopts_the_reference_optimizers_and_keeps_adam_semantics_on_cpu():
    """shims.install_optimizer(): `GaussianModel.training_setup` / `SuGaROptimizer.__init__` leave a FusedAdam -- still a
    torch.optim.Adam with the same groups and state layout; with CPU parameters its step is the parent's step, bit for bit"""
    from tests import ref_env
    if ref_env.reference_root() is None:
        pytest.skip("no reference tree")
    ref_env.import_sugar_model()
    import sugar_scene.sugar_optimizer as so
    from sugar_amd import shims
    from sugar_amd.fused_adam import FusedAdam, adopt
    shims.uninstall_optimizer()
    try:
        assert shims.install_optimizer() >= 1 and shims.install_optimizer() == 0
        assert hasattr(so.SuGaROptimizer.__dict__["__init__"], "_sugar_amd_original")
        g = torch.Generator().manual_seed(3)
        ps = [torch.nn.Parameter(torch.randn(7, 3, generator=g)), torch.nn.Parameter(torch.randn(5, generator=g))]
        qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
        a = adopt(torch.optim.Adam([{"params": [ps[0]], "lr": 1e-2, "name": "a"}, {"params": [ps[1]], "lr": 3e-3, "name": "b"}], lr=0.0, eps=1e-15))
        b = torch.optim.Adam([{"params": [qs[0]], "lr": 1e-2, "name": "a"}, {"params": [qs[1]], "lr": 3e-3, "name": "b"}], lr=0.0, eps=1e-15)
        assert isinstance(a, FusedAdam) and isinstance(a, torch.optim.Adam)
        for it in range(3):
            for p, q in zip(ps, qs):
                gr = torch.randn(p.shape, generator=g)
                p.grad = gr.clone(); q.grad = gr.clone()
            a.step(); b.step()
        for p, q in zip(ps, qs):
            assert torch.equal(p, q)
            assert set(a.state[p]) == set(b.state[q]) == {"step", "exp_avg", "exp_avg_sq"}
            assert torch.equal(a.state[p]["exp_avg_sq"], b.state[q]["exp_avg_sq"])
        assert a.state_dict()["param_groups"][0]["name"] == "a"
    finally:
        assert shims.uninstall_optimizer() >= 1
    assert not hasattr(so.SuGaROptimizer.__dict__["__init__"], "_sugar_amd_original")


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference tree")
def test_patch_densifier_updates_the_statistics_exactly_as_the_reference_methods_do():
    """shims.install(patch_densifier=True): `SuGaRDensifier.update_densification_stats` (sugar_densifier.py:156-164) and
    `GaussianModel.add_densification_stats` (gaussian_model.py:405-407) as full-length masked updates -- bit-identical statistics,
    no boolean-mask indexing; an index-tensor filter still goes to the reference's own method."""
    import types
    from sugar_amd import shims
    from tests import ref_env
    ref_env.import_sugar_model()
    ref_env.import_gaussian_splatting()
    import sugar_scene.sugar_densifier as sd
    import scene.gaussian_model as gmod
    g = torch.Generator().manual_seed(0)
    n = 500
    grad = torch.randn(n, 3, generator=g)
    grad[7] = float("nan")                     # a Gaussian outside the mask may carry anything
    vis = torch.rand(n, generator=g) > 0.4
    vis[7] = False
    radii = torch.randint(0, 40, (n,), generator=g).float()
    vsp = types.SimpleNamespace(grad=grad)

    def fresh():
        return (types.SimpleNamespace(max_radii2D=torch.rand(n, generator=torch.Generator().manual_seed(1)) * 30,
                                      points_gradient_accum=torch.rand(n, 1, generator=torch.Generator().manual_seed(2)),
                                      denom=torch.ones(n, 1)),
                types.SimpleNamespace(xyz_gradient_accum=torch.rand(n, 1, generator=torch.Generator().manual_seed(3)), denom=torch.ones(n, 1)))
    a_s, a_g = fresh()
    sd.SuGaRDensifier.update_densification_stats(a_s, vsp, radii, vis)
    gmod.GaussianModel.add_densification_stats(a_g, vsp, vis)
    assert shims.install_densifier() == 2 and shims.install_densifier() == 0
    try:
        b_s, b_g = fresh()
        sd.SuGaRDensifier.update_densification_stats(b_s, vsp, radii, vis)
        gmod.GaussianModel.add_densification_stats(b_g, vsp, vis)
        for name in ("max_radii2D", "points_gradient_accum", "denom"):
            assert torch.equal(getattr(a_s, name), getattr(b_s, name)), name
        assert torch.equal(a_g.xyz_gradient_accum, b_g.xyz_gradient_accum) and torch.equal(a_g.denom, b_g.denom)
        assert not torch.isnan(b_s.points_gradient_accum).any()
        # an index tensor instead of a mask: the reference's own statements
        c_s, _ = fresh()
        idx = vis.nonzero(as_tuple=True)[0]
        sd.SuGaRDensifier.update_densification_stats(c_s, vsp, radii, idx)
        assert torch.equal(c_s.points_gradient_accum, a_s.points_gradient_accum)
    finally:
        assert shims.uninstall_densifier() == 2
    assert not hasattr(sd.SuGaRDensifier.update_densification_stats, "_sugar_amd_original")
