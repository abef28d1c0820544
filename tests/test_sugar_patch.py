"""CPU tests of sugar_amd.sugar_patch against the UNMODIFIED reference (sugar_scene/sugar_model.py imported from
/root/reference; skipped where the reference tree is absent, i.e. on the GPU box):

  * `shims.install(patch_sugar=...)` replaces exactly the four methods, keeps their signatures, and `uninstall` restores them;
  * on CPU tensors the patched methods hand over to the reference's own code (no CPU compute path of ours);
  * the HOST logic of the replacements -- everything around the kernels: the density normalisation, beta and sdf arithmetic of
    get_field_values, the depth render / unprojection / pixel bookkeeping of the level-set sampler -- reproduces the
    reference's original methods when the kernels are stood in for by the float64-capable torch restatements of oracle/
    (test infrastructure; the GPU tests run the same replacements on the real HIP kernels against the same fixture);
  * the committed fixture tests/golden/sugar_field.npz is what the reference's methods return today.
"""
import inspect
import os
import sys
import types

import numpy as np
import pytest
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")


@pytest.fixture(scope="module")
def ref():
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_sugar_callsite as mk
    import make_sugar_field as mf
    sm = mk._import_reference_model()
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    old = (sm.knn_points, sm.GaussianRasterizer)
    sm.knn_points = mk._scipy_knn_points
    from tests.oracle_rasterizer import GaussianRasterizer as OracleRasterizer
    sm.GaussianRasterizer = OracleRasterizer
    model, cams = mf.build_model(sm)
    yield sm, model, cams, mf
    torch.Tensor.cuda = real_cuda
    sm.knn_points, sm.GaussianRasterizer = old
    from sugar_amd import sugar_patch
    sugar_patch.uninstall(sm)


def test_install_replaces_four_methods_with_the_same_signatures(ref):
    sm, model, _, _ = ref
    from sugar_amd import shims, sugar_patch
    originals = {n: getattr(sm.SuGaR, n) for n in sugar_patch.PATCHED}
    shims.install(patch_sugar=sm)
    shims.install(patch_sugar=sm)  # idempotent
    try:
        for n in sugar_patch.PATCHED:
            assert getattr(sm.SuGaR, n) is not originals[n]
            assert sm.SuGaR._sugar_amd_original[n] is originals[n]
            ref_sig = inspect.signature(originals[n])
            new = inspect.signature(sugar_patch._IMPL[n])
            mine = [p for p in new.parameters.values() if not p.name.startswith("_")]
            theirs = list(ref_sig.parameters.values())
            assert [p.name for p in mine] == [p.name for p in theirs], n
            for a, b in zip(mine, theirs):
                assert a.default == b.default or (a.default is inspect.Parameter.empty and b.default is inspect.Parameter.empty), (n, a.name)
        # CPU tensors: the reference's own code answers (bit-identical results)
        x = model.points[:50].detach() + 0.01
        gi = torch.arange(50)
        a = model.get_field_values(x, gi, return_beta=True, return_closest_gaussian_opacities=True)
        b = originals["get_field_values"](model, x, gi, return_beta=True, return_closest_gaussian_opacities=True)
        assert all(torch.equal(a[k], b[k]) for k in b)
        assert torch.equal(model.get_covariance(return_full_matrix=True, return_sqrt=True, inverse_scales=True),
                           originals["get_covariance"](model, return_full_matrix=True, return_sqrt=True, inverse_scales=True))
        c = torch.tensor([[1.0, 2.0, 0.5]])
        assert torch.equal(model.get_points_rgb(positions=model.points, camera_centers=c, sh_levels=3),
                           originals["get_points_rgb"](model, positions=model.points, camera_centers=c, sh_levels=3))
    finally:
        sugar_patch.uninstall(sm)
    for n in sugar_patch.PATCHED:
        assert getattr(sm.SuGaR, n) is originals[n]


def test_field_values_host_logic_reproduces_the_reference_method(ref):
    sm, model, _, _ = ref
    from oracle import sugar_field_torch as restated
    from sugar_amd import sugar_patch
    g = torch.Generator().manual_seed(3)
    gi = torch.randint(0, model.n_points, (3000,), generator=g)
    x0 = (model.points[gi] + 0.5 * model.scaling[gi] * torch.randn(3000, 3, generator=g)).detach()
    for factor in (1.3, 0.2):  # densities above and below 1
        outs = []
        for patched in (False, True):
            model.zero_grad(set_to_none=True)
            x = x0.clone().requires_grad_(True)
            kw = dict(return_sdf=True, density_threshold=1., density_factor=factor, return_closest_gaussian_opacities=True,
                      return_beta=True)
            if patched:
                f = sugar_patch.get_field_values(model, x, gi, _orig=None, _density_field=restated.density_field, **kw)
            else:
                f = model.get_field_values(x, gi, **kw)
            grads = None
            if factor < 1:
                (f["density"].sum() + 0.3 * f["sdf"].sum() + f["beta"].sum() + f["closest_gaussian_opacities"].sum()).backward()
                grads = [x.grad.clone()] + [getattr(model, n).grad.clone() for n in ("_points", "_scales", "_quaternions", "all_densities")]
            outs.append(({k: v.detach().clone() for k, v in f.items()}, grads))
        (fa, ga), (fb, gb) = outs
        assert set(fa) == set(fb) == {"density", "sdf", "beta", "closest_gaussian_opacities"}
        for k in fa:
            ok = torch.isfinite(fa[k])
            assert torch.equal(ok, torch.isfinite(fb[k])) and torch.allclose(fa[k][ok], fb[k][ok], rtol=1e-5, atol=1e-7), k
        if ga is not None:
            for a, b in zip(ga, gb):
                assert float((a - b).norm() / a.norm()) < 1e-5


def test_level_set_host_logic_reproduces_the_reference_method(ref):
    sm, model, _, mf = ref
    from oracle import sugar_field_torch as restated
    from sugar_amd import sugar_patch

    def stand_in(world, nbr, cam_center, centers, B, strengths, gstd, surface_levels, n_points_in_range, range_size, density_factor,
                 return_normals):
        r = restated.level_set_points(world, nbr, cam_center.reshape(1, 3), centers, B, strengths, gstd, surface_levels,
                                      n_points_in_range, range_size, density_factor)
        return {lv: dict(valid=r[lv]["valid"], intersection_points=r[lv]["intersection_points"], normals=r[lv]["normals"]) for lv in r}

    kw = dict(cam_idx=5, rasterizer=None, surface_levels=mf.LEVELS, n_surface_points=-1, primitive_types='diamond', triangle_scale=2.,
              n_points_in_range=21, range_size=3., density_factor=1., return_pixel_idx=True, return_gaussian_idx=True,
              return_normals=True, use_gaussian_depth=True)
    with torch.no_grad():
        a = model.compute_level_surface_points_from_camera_fast(**kw)  # the reference's own method
        # the replacement hands CPU tensors to `_orig`; with _orig=None it runs its own host logic on whatever it is given
        b = sugar_patch.compute_level_surface_points_from_camera_fast(model, _orig=None, _level_set_points=stand_in, **kw)
    for lv in mf.LEVELS:
        assert set(a[lv]) == set(b[lv]) == {"intersection_points", "pixel_idx", "gaussian_idx", "normals"}
        assert torch.equal(a[lv]["pixel_idx"], b[lv]["pixel_idx"]) and len(a[lv]["pixel_idx"]) > 300
        assert torch.equal(a[lv]["gaussian_idx"], b[lv]["gaussian_idx"])
        assert torch.allclose(a[lv]["intersection_points"], b[lv]["intersection_points"], rtol=1e-5, atol=1e-6)
        assert torch.allclose(a[lv]["normals"], b[lv]["normals"], rtol=1e-4, atol=1e-5)


def test_committed_field_fixture_is_what_the_reference_returns(ref):
    _, _, _, mf = ref
    out = mf.run()
    gold = np.load(os.path.join(HERE, "golden", "sugar_field.npz"))
    assert set(out) == set(gold.files)
    for k in gold.files:
        a, b = np.asarray(out[k]), gold[k]
        assert a.shape == b.shape, k
        if a.dtype.kind in "iub" or k.startswith("state") or k in ("field_x", "W", "H"):
            assert np.array_equal(a, b), k
        else:
            ok = np.isfinite(b)
            np.testing.assert_allclose(a[ok], b[ok], rtol=2e-4, atol=1e-6 * max(1.0, float(np.abs(b[ok]).max())), err_msg=k)


def test_level_set_host_logic_on_the_mesh_depth_path_reproduces_the_reference_method(ref):
    """`use_gaussian_depth=False` (coarse_mesh.py:26): depth and front Gaussian from the splat mesh's z-buffer.  Both sides run the
    stand-in MeshRasterizer on the CPU oracle backend; the replacement's own host logic (no texture, fragments -> depth ->
    unprojection -> knn_idx rows) must give the reference method's pixels, Gaussians and points, also for a seeded random subset."""
    sm, model, _, mf = ref
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_sugar_meshdepth as mm
    from oracle import sugar_field_torch as restated
    from sugar_amd import sugar_patch
    from tests.mesh_backend import oracle_mesh_rasterizer

    def stand_in(world, nbr, cam_center, centers, B, strengths, gstd, surface_levels, n_points_in_range, range_size, density_factor,
                 return_normals):
        r = restated.level_set_points(world, nbr, cam_center.reshape(1, 3), centers, B, strengths, gstd, surface_levels,
                                      n_points_in_range, range_size, density_factor)
        return {lv: dict(valid=r[lv]["valid"], intersection_points=r[lv]["intersection_points"], normals=r[lv]["normals"]) for lv in r}

    with torch.no_grad(), oracle_mesh_rasterizer():
        model.primitive_types = 'diamond'
        model.triangle_scale = 2.
        model.update_texture_features()
        rasterizer = mm.make_rasterizer(model)
        model._sugar_amd_cpu_randperm = True  # (same draw as the reference, sugar_model.py:1955)
        for n in (-1, 500):
            kw = mm.sampler_kwargs(n)
            torch.manual_seed(9)
            a = model.compute_level_surface_points_from_camera_fast(rasterizer=rasterizer, **kw)
            torch.manual_seed(9)
            b = sugar_patch.compute_level_surface_points_from_camera_fast(model, rasterizer=rasterizer, _orig=None,
                                                                          _level_set_points=stand_in, **kw)
            for lv in mm.LEVELS:
                assert set(a[lv]) == set(b[lv]) == {"intersection_points", "pixel_idx", "gaussian_idx", "normals"}
                assert torch.equal(a[lv]["pixel_idx"], b[lv]["pixel_idx"]) and len(a[lv]["pixel_idx"]) > 150
                assert torch.equal(a[lv]["gaussian_idx"], b[lv]["gaussian_idx"])
                assert torch.allclose(a[lv]["intersection_points"], b[lv]["intersection_points"], rtol=1e-5, atol=1e-6)
                assert torch.allclose(a[lv]["normals"], b[lv]["normals"], rtol=1e-4, atol=1e-5)
        # without a rasterizer argument the replacement builds the one the reference builds (:1880-1893)
        c = sugar_patch.compute_level_surface_points_from_camera_fast(model, rasterizer=None, _orig=None, _level_set_points=stand_in,
                                                                      **mm.sampler_kwargs(-1))
        full = model.compute_level_surface_points_from_camera_fast(rasterizer=None, **mm.sampler_kwargs(-1))
        assert all(torch.equal(c[lv]["pixel_idx"], full[lv]["pixel_idx"]) for lv in mm.LEVELS)


def test_committed_meshdepth_fixture_is_what_the_reference_returns(ref):
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_sugar_meshdepth as mm
    out = mm.run()
    gold = np.load(os.path.join(HERE, "golden", "sugar_meshdepth.npz"))
    assert set(out) == set(gold.files)
    for k in gold.files:
        a, b = np.asarray(out[k]), gold[k]
        assert a.shape == b.shape, k
        if a.dtype.kind in "iub":
            assert np.array_equal(a, b), k
        else:
            np.testing.assert_allclose(a, b, rtol=2e-4, atol=1e-6 * max(1.0, float(np.abs(b).max())), err_msg=k)


def test_row_gather_wrappers_leave_cpu_models_alone_and_come_off_again():
    """sugar_patch.install_row_gathers on the reference class: properties and get_normals keep returning what they returned for a CPU
    model (plain tensors, same values, gradients flow), and uninstall restores the original descriptors"""
    from tests import ref_env
    if ref_env.reference_root() is None:
        pytest.skip("no reference tree")
    sm = ref_env.import_sugar_model()
    from sugar_amd import sugar_patch
    cls = sm.SuGaR
    before = {n: cls.__dict__[n] for n in sugar_patch.ROW_GATHER_PROPERTIES + sugar_patch.ROW_GATHER_METHODS}
    names = sugar_patch.install_row_gathers(sm)
    try:
        assert set(names) == set(before) and sugar_patch.install_row_gathers(sm) == names      # idempotent
        assert all(cls.__dict__[n] is not before[n] for n in before)
        fake = types.SimpleNamespace(binded_to_surface_mesh=False, learnable_positions=True, learnable_shifts=False,
                                     _points=torch.nn.Parameter(torch.randn(6, 3)),
                                     _quaternions=torch.nn.Parameter(torch.randn(6, 4)))
        pts = cls.points.fget(fake)
        q = cls.quaternions.fget(fake)
        assert type(q) is torch.Tensor and pts is fake._points
        assert torch.allclose(q, torch.nn.functional.normalize(fake._quaternions, dim=-1))
        q[torch.tensor([0, 0, 5])].sum().backward()
        assert fake._quaternions.grad is not None
    finally:
        sugar_patch.uninstall_row_gathers(sm)
    assert all(cls.__dict__[n] is before[n] for n in before)


def test_random_prefix_of_permutation_is_a_sample_without_replacement():
    """distinct, in range, right length; every index equally likely in every slot (chi-square over many draws, fixed seed); the
    large-k branch is torch.randperm itself"""
    from sugar_amd.sugar_patch import random_prefix_of_permutation as rp
    torch.manual_seed(1234)
    n, k = 400, 40            # k * 8 <= n: the rejection branch
    counts = torch.zeros(n)
    first = torch.zeros(n)
    trials = 4000
    for _ in range(trials):
        s = rp(n, k, "cpu")
        assert s.shape == (k,) and s.dtype == torch.int64 and int(s.min()) >= 0 and int(s.max()) < n and len(set(s.tolist())) == k
        counts[s] += 1
        first[s[0]] += 1
    exp = trials * k / n
    chi2 = float(((counts - exp) ** 2 / exp).sum())
    assert chi2 < 400 + 5 * (2 * 400) ** 0.5, chi2            # chi-square with 399 degrees of freedom: mean 399, sd 28
    chi2_first = float(((first - trials / n) ** 2 / (trials / n)).sum())
    assert chi2_first < 400 + 5 * (2 * 400) ** 0.5, chi2_first
    s = rp(100, 60, "cpu")
    assert len(set(s.tolist())) == 60


def test_the_standalone_sampler_has_no_cpu_path():
    """sugar_amd.sampler is device code behind a thin wrapper: CPU tensors are refused loudly, nothing falls back"""
    import pytest
    from sugar_amd import sampler, synthetic as syn
    cam = syn.orbit_cameras(64, 48, 2)[0]
    m = torch.zeros(10, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sampler.sample_level_sets(m, torch.ones(10, 3), torch.tensor([[1.0, 0, 0, 0]]).repeat(10, 1), torch.ones(10, 1), cam)

