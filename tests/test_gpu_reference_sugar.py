"""THE REFERENCE'S OWN CLASSES on the MI355X with the HIP kernels underneath (verdict of round 3, items "missing 1, 2").

The reference's Python is imported UNMODIFIED -- from /root/reference where it exists, on the GPU box from the git-ignored
snapshot oracle/_ref/pysrc that oracle/ref_build/build_ref.sh stages next to the compiled reference kernels -- with this
repository's drop-in packages on the path: `diff_gaussian_rasterization` and `simple_knn` (HIP), the stand-in `pytorch3d`
(camera algebra, HIP k-NN, HIP mesh z-buffer).  Nothing is stood in for on the device side: `.cuda()` is real, the cameras are the
reference's own `CamerasWrapper` / `GSCamera` / `convert_camera_from_gs_to_pytorch3d`.

  (a) `SuGaR.render_image_gaussian_rasterizer` (sugar_model.py:2085-2294), free and bound to a surface mesh, forward and backward
      onto the model's own parameters, against the committed fixtures the same method wrote on the CPU oracle rasterizer;
  (b) `SuGaR.compute_level_surface_points_from_camera_fast(use_gaussian_depth=False, ...)` with the arguments of
      sugar_extractors/coarse_mesh.py:271-287 -- the splat mesh through `MeshRasterizer` (HIP z-buffer) -- both the untouched
      method (its level sets are the reference's tensor code) and the one `shims.install(patch_sugar=True)` routes to
      `k_level_set`, against the fixture the untouched method wrote on the CPU (tests/golden/make_sugar_meshdepth.py);
  (c) 20 iterations of the reference's optimisation loop (train.py:69-128: its render(), l1_loss / ssim, backward(),
      GaussianModel.optimizer.step()) against NativeTrainer on the same scene and views."""
import math
import os
import sys
import types

import numpy as np
import pytest
import torch

from sugar_amd import synthetic as syn
from tests import ref_env

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(ref_env.reference_root() is None, reason="no reference tree and no staged snapshot "
                                                  "(oracle/ref_build/build_ref.sh stages oracle/_ref/pysrc in the build container)")]
DEV = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
STATE = ("_points", "_scales", "_quaternions", "all_densities", "_sh_coordinates_dc", "_sh_coordinates_rest")


def _rel(a, b):
    a = torch.as_tensor(a).detach().cpu().double(); b = torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def sm():
    mod = ref_env.import_sugar_model()
    yield mod
    from sugar_amd import sugar_patch
    sugar_patch.uninstall(mod)


def _training_cameras(cams):
    """the reference's own camera classes around the synthetic orbit (sugar_scene/cameras.py:140-216, :418-465)"""
    from sugar_scene.cameras import CamerasWrapper, GSCamera
    gs = []
    for i, c in enumerate(cams):
        w2c = c.viewmatrix.t().double().numpy()
        gs.append(GSCamera(colmap_id=i, R=w2c[:3, :3].T.copy(), T=w2c[:3, 3].copy(), FoVx=2 * math.atan(c.tanfovx),
                           FoVy=2 * math.atan(c.tanfovy), image=None, gt_alpha_mask=None, image_name=f"view_{i}", uid=i,
                           image_height=c.image_height, image_width=c.image_width))
    return CamerasWrapper(gs)


def _load_state(model, fx):
    with torch.no_grad():
        for n in STATE:
            p = getattr(model, n)
            v = torch.as_tensor(fx["state" + n]).to(p.device)
            assert p.shape == v.shape, (n, p.shape, v.shape)
            p.copy_(v)
        if "state_knn_idx" in fx.files:
            model.knn_idx = torch.as_tensor(fx["state_knn_idx"]).to(model.device)


def _free_model(sm, fx, W, H, seed=77):
    cams = syn.orbit_cameras(W, H)
    nerf = types.SimpleNamespace(device=torch.device(DEV), training_cameras=_training_cameras(cams))
    P = fx["state_points"].shape[0]
    g = torch.Generator().manual_seed(seed)
    pts = torch.as_tensor(fx["state_points"]).to(DEV)
    cols = torch.rand(P, 3, generator=g).to(DEV)
    model = sm.SuGaR(nerfmodel=nerf, points=pts, colors=cols, initialize=True, sh_levels=4, keep_track_of_knn=True, knn_to_track=16)
    _load_state(model, fx)
    return model, cams


def _check_render(model, fx, calls, bgs):
    wimg = torch.as_tensor(fx["dL_dimage_hw3"]).to(DEV)
    for ci, (cam_idx, in_rast) in enumerate(calls):
        pre = f"c{ci}_"
        model.zero_grad(set_to_none=True)
        res = model.render_image_gaussian_rasterizer(camera_indices=cam_idx, bg_color=bgs[ci], sh_deg=3,
                                                     compute_color_in_rasterizer=in_rast, return_2d_radii=True)
        (res["image"] * wimg).sum().backward()
        assert res["image"].is_cuda
        err = _rel(res["image"], fx[pre + "image_hw3"])
        assert err < 2e-5, (pre, err)
        radii = res["radii"].cpu().numpy()
        assert (radii == fx[pre + "radii"]).mean() > 0.999, pre  # (the cameras' float path differs from the fixture's: see module doc)
        for name in STATE:
            g = getattr(model, name).grad
            e = _rel(g, fx[pre + "param_grad" + name])
            assert e < 2e-4, (pre, name, e)


def test_the_reference_sugar_class_renders_through_the_hip_rasterizer(sm):
    fx = np.load(os.path.join(GOLD, "sugar_callsite.npz"))
    model, _ = _free_model(sm, fx, int(fx["W"]), int(fx["H"]))
    _check_render(model, fx, ((1, False), (5, True)), [None, torch.tensor([1.0, 1.0, 1.0], device=DEV)])


def test_the_reference_sugar_class_bound_to_a_surface_mesh_renders_through_the_hip_rasterizer(sm):
    sys.path.insert(0, GOLD)
    import make_sugar_callsite as mk
    fx = np.load(os.path.join(GOLD, "sugar_callsite_bound.npz"))
    W, H = int(fx["W"]), int(fx["H"])
    cams = syn.orbit_cameras(W, H)
    nerf = types.SimpleNamespace(device=torch.device(DEV), training_cameras=_training_cameras(cams))
    mesh = mk._bumpy_sphere()
    model = sm.SuGaR(nerfmodel=nerf, points=None, colors=None, initialize=False, sh_levels=4, keep_track_of_knn=False,
                     surface_mesh_to_bind=mesh, n_gaussians_per_surface_triangle=6, learn_surface_mesh_positions=True,
                     learn_surface_mesh_opacity=True, learn_surface_mesh_scales=True)
    assert model.binded_to_surface_mesh and model._n_points == 6 * int(fx["n_faces"])
    _load_state(model, fx)
    _check_render(model, fx, ((2, False), (6, True)), [None, None])


def _compare_levels(res, fx, tag, scale, min_common=0.995, min_pix=300):
    for lv in (0.1, 0.3, 0.5):
        t = f"{tag}_{int(round(lv * 10))}_"
        ref_pix = fx[t + "pixel_idx"]
        out = res[lv]
        pix = out["pixel_idx"].cpu().numpy()
        common, ia, ib = np.intersect1d(pix, ref_pix, return_indices=True)
        assert len(ref_pix) > min_pix and len(common) >= min_common * max(len(pix), len(ref_pix)), (t, len(pix), len(ref_pix), len(common))
        assert (out["gaussian_idx"].cpu().numpy()[ia] == fx[t + "gaussian_idx"][ib]).mean() > 0.998, t
        d = np.linalg.norm(out["intersection_points"].cpu().numpy()[ia] - fx[t + "points"][ib], axis=1)
        assert np.quantile(d, 0.995) < 2e-3 * scale, (t, np.quantile(d, 0.995), scale)
        dots = (out["normals"].cpu().numpy()[ia] * fx[t + "normals"][ib]).sum(axis=1)
        assert np.quantile(dots, 0.005) > 0.9999, (t, np.quantile(dots, 0.005))


def test_the_reference_level_set_sampler_on_the_mesh_depth_path(sm):
    sys.path.insert(0, GOLD)
    import make_sugar_meshdepth as mm
    from sugar_amd import shims, sugar_patch
    state = np.load(os.path.join(GOLD, "sugar_field.npz"))
    fx = np.load(os.path.join(GOLD, "sugar_meshdepth.npz"))
    model, _ = _free_model(sm, state, int(state["W"]), int(state["H"]))
    scale = float(np.exp(state["state_scales"]).mean())
    loaded_before = _loaded_hip_library()
    with torch.no_grad():
        model.primitive_types = 'diamond'   # coarse_mesh.py:207-210
        model.triangle_scale = 2.
        model.update_texture_features()
        rasterizer = mm.make_rasterizer(model)
        cam = model.nerfmodel.training_cameras.p3d_cameras[int(fx["cam_idx"])]
        # -- the fragments of the splat mesh: nearest face and depth per pixel (HIP z-buffer) against the CPU oracle's
        mesh = model.splat_mesh(cam)
        assert _rel(mesh.verts_list()[0], fx["splat_verts"]) < 1e-5
        # -- the same faces out of ONE kernel reading the Gaussian buffers (sgr_splat_mesh_face_verts) against the reference's
        #    triangle_vertices / splat_mesh followed by the rasterizer's vertex transform
        from sugar_amd.mesh_raster import splat_face_verts
        proj_mesh = rasterizer.transform(mesh, cameras=cam)
        ref_fv = proj_mesh.verts_packed()[proj_mesh.faces_packed()]
        fused = splat_face_verts(model.points, model.scaling, model.quaternions, model._diamond_verts, model.triangle_scale,
                                 cam.get_world_to_view_transform().get_matrix(), cam.get_projection_transform().get_matrix())
        assert fused.shape == ref_fv.shape == (2 * model.n_points, 3, 3)
        finite = torch.isfinite(ref_fv).all(-1).all(-1) & torch.isfinite(fused).all(-1).all(-1)
        assert float(finite.float().mean()) > 0.999
        err = (fused[finite] - ref_fv[finite]).abs().amax(dim=(1, 2)) / ref_fv[finite].abs().amax(dim=(1, 2)).clamp_min(1e-3)
        assert float(err.max()) < 2e-4 and float(err.median()) < 2e-6, (float(err.max()), float(err.median()))
        fr = rasterizer(mesh, cameras=cam)
        assert fr.zbuf.is_cuda and fr.pix_to_face.dtype == torch.int64 and fr.pix_to_face.shape == (1,) + fx["frag_pix_to_face"].shape
        same = (fr.pix_to_face[0].cpu().numpy() == fx["frag_pix_to_face"])
        assert same.mean() > 0.999, same.mean()
        z, zr = fr.zbuf[0].cpu().numpy(), fx["frag_zbuf"]
        assert np.allclose(z[same], zr[same], rtol=2e-5, atol=0)
        n_cover, n_cover_ref = int((z[..., 0] >= 0).sum()), int((zr[..., 0] >= 0).sum())
        # -- the UNTOUCHED method (sugar_model.py:1848-2083): its level sets are the reference's tensor code on the device
        assert sm.SuGaR.compute_level_surface_points_from_camera_fast.__module__ == "sugar_scene.sugar_model"
        res = model.compute_level_surface_points_from_camera_fast(rasterizer=rasterizer, **mm.sampler_kwargs(-1))
        _compare_levels(res, fx, "all", scale)
        # -- the same call routed to the fused level-set kernel, as shims.install(patch_sugar=True) does for a user
        shims.install(patch_sugar=sm)
        try:
            assert sm.SuGaR.compute_level_surface_points_from_camera_fast is not sm.SuGaR._sugar_amd_original["compute_level_surface_points_from_camera_fast"]
            res2 = model.compute_level_surface_points_from_camera_fast(rasterizer=rasterizer, **mm.sampler_kwargs(-1))
            _compare_levels(res2, fx, "all", scale)
            for lv in (0.1, 0.3, 0.5):  # untouched vs routed on the same device inputs
                common, ia, ib = np.intersect1d(res[lv]["pixel_idx"].cpu().numpy(), res2[lv]["pixel_idx"].cpu().numpy(), return_indices=True)
                assert len(common) >= 0.998 * len(res[lv]["pixel_idx"])
                assert torch.equal(res[lv]["gaussian_idx"][ia], res2[lv]["gaussian_idx"][ib])
            # the extractor's call: a seeded random subset of the pixels (coarse_mesh.py:274: n_surface_points = 2 * n_pts_per_frame)
            n_sub = int(fx["n_subset"])
            model._sugar_amd_cpu_randperm = True  # draw the subset where the reference draws it (sugar_model.py:1955)
            torch.manual_seed(int(fx["seed"]))
            sub = model.compute_level_surface_points_from_camera_fast(rasterizer=rasterizer, **mm.sampler_kwargs(n_sub))
            full_pix = {lv: set(res2[lv]["pixel_idx"].cpu().tolist()) for lv in (0.1, 0.3, 0.5)}
            for lv in (0.1, 0.3, 0.5):
                sp = sub[lv]["pixel_idx"].cpu().tolist()
                assert 0 < len(sp) <= n_sub and len(set(sp)) == len(sp) and set(sp) <= full_pix[lv]
            if n_cover == n_cover_ref:  # same number of covered pixels -> the CPU-drawn permutation is the fixture's
                _compare_levels(sub, fx, "sub", scale, min_common=0.99, min_pix=100)
        finally:
            sugar_patch.uninstall(sm)
    assert loaded_before


def _loaded_hip_library():
    with open("/proc/self/maps") as f:
        return any("libsugar_raster.so" in line for line in f)


def test_twenty_iterations_of_the_reference_loop_against_the_native_trainer():
    """gaussian_splatting/train.py:69-128 with the reference's render / losses / GaussianModel Adam on the HIP drop-in rasterizer,
    next to sgr_trainer_step (NativeTrainer) from the same parameters on the same 8 views: losses and parameters after 20 Adam
    steps.  (Adam with eps 1e-15 normalises every gradient to a +-lr step at first, so parameters whose gradient is pure
    rounding noise may step in opposite directions: the bar is on the bulk, as in tests/test_gpu_distributed.py.)"""
    from oracle import reference_loop as rl
    from sugar_amd.train_step import GaussianParams, NativeTrainer
    ref = rl.import_reference()
    dev = torch.device(DEV)
    W, H, P = 400, 304, 60_000
    scene = syn.make_scene(P, 3, 0.004, 0.03)
    cams = syn.orbit_cameras(W, H)
    g = torch.Generator().manual_seed(5)
    gts = [torch.rand(3, H, W, generator=g).to(dev) for _ in cams]
    bg = torch.zeros(3, device=dev)
    opt = rl.optimization_params(constant_position_lr=True)  # NativeTrainer takes the learning rates from its caller: no schedule inside
    gaussians = rl.make_gaussians(ref, scene, dev, opt)
    loop = rl.Loop(ref, gaussians, [rl.make_viewpoint(c, gt, dev) for c, gt in zip(cams, gts)], bg, opt=opt, sequential=True)
    params = GaussianParams(scene, dev)
    trainer = NativeTrainer(params, bg, W, H)
    dcams = [syn.Camera(c.image_height, c.image_width, c.tanfovx, c.tanfovy, c.viewmatrix.to(dev), c.projmatrix.to(dev), c.campos.to(dev)) for c in cams]
    ref_losses, nat_losses = [], []
    for it in range(20):
        ref_losses.append(float(loop.loop_body()))
        trainer.step(dcams[it % len(cams)], gts[it % len(cams)], cam_key=it % len(cams))
        trainer.synchronize()
        nat_losses.append(float(trainer.loss_out[0]))
    assert np.allclose(ref_losses, nat_losses, rtol=2e-4), (ref_losses, nat_losses)
    assert np.mean(ref_losses[-8:]) < np.mean(ref_losses[:8])  # (the same eight views, two passes later)
    pairs = (("xyz", gaussians._xyz, 0.00016), ("opacity", gaussians._opacity, 0.05), ("scaling", gaussians._scaling, 0.005),
             ("rotation", gaussians._rotation, 0.001))
    for name, theirs, lr in pairs:
        mine = params.params[name].detach()
        d = (mine - theirs.detach()).abs()
        # a step is at most lr; after 20 steps two runs can differ by 40 lr only where gradients are noise -- the bulk agrees
        assert float((d > 2 * lr).float().mean()) < 0.02, (name, float((d > 2 * lr).float().mean()))
        assert float(d.median()) < 0.05 * lr, (name, float(d.median()), lr)
    feats = torch.cat([gaussians._features_dc, gaussians._features_rest], dim=1).detach()
    d = (params.params["features"].detach() - feats).abs()
    assert float((d[:, :1] > 2 * 0.0025).float().mean()) < 0.02 and float(d[:, :1].median()) < 0.05 * 0.0025
    # densification statistics of the loop (train.py:111-114) exist and saw the views
    assert float(gaussians.denom.sum()) > 0 and float(gaussians.max_radii2D.max()) > 0


def test_the_standalone_sampler_equals_the_reference_class_on_the_gaussian_depth_path(sm):
    """sugar_amd.sampler.sample_level_sets (raw Gaussian buffers + a plain camera tuple: what bench.py's config-4 line runs) against
    `SuGaR.compute_level_surface_points_from_camera_fast(use_gaussian_depth=True)` of the reference's own class with its own camera
    objects (routed to the same kernels by shims.install(patch_sugar=True), which the round-3 fixtures pin to the reference's
    tensor code): same pixels picked (CPU permutation, same seed), same front Gaussians, same crossing points and normals."""
    from sugar_amd import sampler, shims, sugar_patch
    state = np.load(os.path.join(GOLD, "sugar_field.npz"))
    W, H = int(state["W"]), int(state["H"])
    model, cams = _free_model(sm, state, W, H)
    scale = float(np.exp(state["state_scales"]).mean())
    cam_idx = 3
    c = cams[cam_idx]
    cam = syn.Camera(c.image_height, c.image_width, c.tanfovx, c.tanfovy, c.viewmatrix.to(DEV), c.projmatrix.to(DEV), c.campos.to(DEV))
    n_sub = 1500
    shims.install(patch_sugar=sm)
    try:
        model._sugar_amd_cpu_randperm = True
        with torch.no_grad():
            torch.manual_seed(11)
            ref = model.compute_level_surface_points_from_camera_fast(
                cam_idx=cam_idx, surface_levels=[0.1, 0.3, 0.5], n_surface_points=n_sub, n_points_in_range=21, range_size=3.,
                density_factor=1., return_pixel_idx=True, return_gaussian_idx=True, return_normals=True, use_gaussian_depth=True)
            torch.manual_seed(11)
            got = sampler.sample_level_sets(model.points, model.scaling, model.quaternions, model.strengths, cam,
                                            n_surface_points=n_sub, cpu_randperm=True)
    finally:
        sugar_patch.uninstall(sm)
    for lv in (0.1, 0.3, 0.5):
        a, b = got[lv], ref[lv]
        pa, pb = a["pixel_idx"].cpu().numpy(), b["pixel_idx"].cpu().numpy()
        common, ia, ib = np.intersect1d(pa, pb, return_indices=True)
        assert len(pb) > 100 and len(common) >= 0.99 * max(len(pa), len(pb)), (lv, len(pa), len(pb), len(common))
        assert (a["gaussian_idx"].cpu().numpy()[ia] == b["gaussian_idx"].cpu().numpy()[ib]).mean() > 0.995
        d = np.linalg.norm(a["intersection_points"].cpu().numpy()[ia] - b["intersection_points"].cpu().numpy()[ib], axis=1)
        assert np.quantile(d, 0.99) < 2e-3 * scale, (lv, np.quantile(d, 0.99), scale)
        dots = (a["normals"].cpu().numpy()[ia] * b["normals"].cpu().numpy()[ib]).sum(axis=1)
        assert np.quantile(dots, 0.01) > 0.9999
