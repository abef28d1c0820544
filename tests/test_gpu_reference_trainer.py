"""The reference's coarse trainer itself (sugar_trainers/coarse_sdf.py:17-825, untouched) on the HIP drop-ins: it loads a 3DGS
checkpoint and ground-truth images from disk (through the `plyfile` stand-in and the reference's own camera loader), builds the
reference's SuGaR model, optimiser and densifier, and optimises through entropy regularisation, the opacity pruning at 9 000 and
the first SDF / normal regularisation iterations (1M density samples per iteration, depth renders, neighbour resets).  The run
is cut at iteration 9 060 by a counter around the module-level `ssim` -- see oracle/reference_trainer.py.  What is asserted is
what a user switching rasterizers would look at: every logged loss finite, the photometric phase converging, the regularisers
announced when the schedule says, Gaussians pruned, no exception anywhere on the way."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("trainer,patches", [("coarse_sdf", False), ("coarse_density", True)])
def test_the_unmodified_coarse_trainer_runs_through_its_schedule_on_the_drop_ins(tmp_path, trainer, patches):
    from tests import ref_env
    if ref_env.reference_root() is None:
        pytest.skip("the reference's Python is not staged (oracle/ref_build/build_ref.sh)")
    from oracle import reference_trainer as rt
    data = rt.write_dataset(str(tmp_path / "data"), P=30_000, n_cams=24, W=320, H=208)
    # (`patches`: also the reference's `ssim`, the optimiser it builds and the row gathers of SuGaR's per-Gaussian tensors on the HIP
    # kernels -- shims.install(patch_losses=True, patch_optimizer=True, patch_gathers=True); the second trainer of the reference, coarse_density.py, has the same loop with a density regulariser)
    res = rt.run(data, str(tmp_path / "out"), stop_at=9060, patch_sugar=True, patch_losses=patches, patch_optimizer=patches,
                 patch_gathers=patches, trainer=trainer)
    assert not res["finished"] and res["last_iteration"] == 9060
    if patches:   # every optimiser step of the run went through the HIP Adam (SuGaR's strided `_scales` / `_quaternions` included)
        assert res["adam_stats"]["fused_steps"] >= 2061 and res["adam_stats"]["fallback_steps"] == 0, res["adam_stats"]
    its = [i for i, _ in res["losses"]]
    assert its[0] == 7000 and its[-1] == 9050 and len(its) == 42          # one line per 50 iterations (coarse_sdf.py:221,761)
    assert all(math.isfinite(v) for _, v in res["losses"])
    by_it = dict(res["losses"])
    # photometric + entropy phase: the perturbed checkpoint is pulled back onto the ground-truth views
    early = sum(by_it[i] for i in (7000, 7050, 7100)) / 3
    late = sum(by_it[i] for i in (8850, 8900, 8950)) / 3
    assert late < 0.8 * early, (early, late)
    for e in ("Starting entropy regularization", "Stopping entropy regularization", "Pruning gaussians with low-opacity",
              "Starting SDF regularization", "Starting SDF estimation loss", "Starting SDF better normal loss", "Resetting neighbors"):
        assert e in res["events"], e
    assert 0 < res["gaussians_after_pruning"] <= 30_000
    assert torch.cuda.is_available()


def test_the_unmodified_coarse_mesh_extractor_samples_every_camera_up_to_its_poisson_step(tmp_path):
    """sugar_extractors/coarse_mesh.py::extract_mesh_from_coarse_sugar, untouched, in its `--use_vanilla_3dgs` mode: checkpoint
    and cameras from disk, pruning, the splat mesh, pytorch3d's MeshRasterizer (here: the HIP z-buffer behind the stand-in), then
    per training camera an RGB render and compute_level_surface_points_from_camera_fast(use_gaussian_depth=False) for the levels
    0.1 / 0.3 / 0.5 (coarse_mesh.py:243-327).  The first open3d call (Poisson) ends the run; the accumulated point clouds are read
    from the function's frame.  The scene is a known surface, so the samples can be judged: they lie on it, normals along it."""
    from tests import ref_env
    if ref_env.reference_root() is None:
        pytest.skip("the reference's Python is not staged (oracle/ref_build/build_ref.sh)")
    from oracle import reference_trainer as rt
    data = rt.write_dataset(str(tmp_path / "data"), P=30_000, n_cams=24, W=320, H=208)
    res = rt.run_extractor(data, str(tmp_path / "extract"), coarse_model_path=None, patch_sugar=True, patch_gathers=True)
    assert res["reached_poisson"] and res["cameras"] == 21          # 24 views, every 8th held out (coarse_mesh.py:20-21)
    assert sorted(res["outputs"]) == [0.1, 0.3, 0.5]
    for level, o in res["outputs"].items():
        p, n = o["points"], o["normals"]
        assert p.is_cuda and p.shape[0] > 100_000 and p.shape == n.shape and o["pix_to_gaussians"].shape[0] == p.shape[0]
        assert bool(torch.isfinite(p).all()) and bool(torch.isfinite(n).all())
        d = p / p.norm(dim=1, keepdim=True)
        err = (p.norm(dim=1) - rt.surface_radius(d)).abs()
        assert float(err.median()) < 0.03 and float(err.quantile(0.9)) < 0.06, (level, float(err.median()))
        assert float((torch.nn.functional.normalize(n, dim=1) * d).sum(1).abs().median()) > 0.9
        assert int(o["pix_to_gaussians"].min()) >= 0


def test_the_unmodified_refinement_trainer_optimises_gaussians_bound_to_a_mesh(tmp_path):
    """sugar_trainers/refine.py::refined_training, untouched (BASELINE config 4's model): six flat Gaussians per triangle of a
    surface mesh, vertices + in-plane scales / rotations + SH + opacities optimised through the HIP rasterizer with pytorch3d's
    mesh_normal_consistency (stand-in) on the mesh; the mesh comes from the harness's `open3d.io.read_triangle_mesh`.  The trainer
    ends by itself, saves its checkpoint and exports the Gaussians through GaussianModel.save_ply (plyfile stand-in)."""
    from tests import ref_env
    if ref_env.reference_root() is None:
        pytest.skip("the reference's Python is not staged (oracle/ref_build/build_ref.sh)")
    from oracle import reference_trainer as rt
    from sugar_amd import io as sio
    data = rt.write_dataset(str(tmp_path / "data"), P=30_000, n_cams=24, W=320, H=208)
    res = rt.run_refine(data, str(tmp_path / "refine"), iterations=200, mesh_level=4, patch_sugar=True, patch_losses=True,
                        patch_optimizer=True, patch_gathers=True)
    assert res["finished"] and res["iterations_run"] == 200 and res["mesh_faces"] == 5120
    its = [i for i, _ in res["losses"]]
    assert its == [1, 50, 100, 150, 200]
    vals = [v for _, v in res["losses"]]
    assert all(math.isfinite(v) for v in vals) and vals[-1] < 0.5 * vals[0], vals
    ply = sio.load_gaussian_ply(res["exported_ply"])
    assert ply["xyz"].shape == (5120 * 6, 3) and ply["features"].shape == (5120 * 6, 16, 3)
    assert bool(torch.isfinite(ply["xyz"]).all()) and bool(torch.isfinite(ply["scaling"]).all())
    # the bound Gaussians still sit on the surface they were tied to
    p = ply["xyz"]
    d = p / p.norm(dim=1, keepdim=True)
    assert float((p.norm(dim=1) - rt.surface_radius(d)).abs().median()) < 0.02


def test_the_vanilla_3dgs_trainer_command_line_runs_through_the_launcher(tmp_path):
    """`python -m sugar_amd.launch <reference>/gaussian_splatting/train.py -s ... --iterations 800 --eval ...`: the reference's script
    and everything under it, untouched -- Scene / COLMAP text reader / points3D -> PLY / create_from_pcd from 10k SfM-like points / the loop of
    train.py:69-128 with densification from iteration 500 / evaluation / save -- with all opt-in bindings of the launcher on.  From
    a sparse point cloud to renders of the training views above 22 dB in 800 iterations, and a point cloud on disk that grew."""
    from tests import ref_env
    if ref_env.reference_root() is None:
        pytest.skip("the reference's Python is not staged (oracle/ref_build/build_ref.sh)")
    from oracle import reference_trainer as rt
    from sugar_amd import io as sio
    data = rt.write_colmap_dataset(str(tmp_path / "colmap"), P=30_000, n_cams=32, W=320, H=208, n_sfm_points=10_000)
    res = rt.run_vanilla_cli(data, str(tmp_path / "out"), iterations=800)
    assert res["returncode"] == 0 and res["complete"], res["text"][-3000:]
    assert "'patch_losses': 1" in res["launch_line"] and "'patch_optimizer': 1" in res["launch_line"], res["launch_line"]
    by = {(it, name): psnr for it, name, _, psnr in res["evals"]}
    assert by[(800, "train")] > 22.0 and by[(800, "test")] > 20.0 and by[(800, "train")] > by[(400, "train")], res["evals"]
    ply = sio.load_gaussian_ply(res["ply"])
    assert ply["xyz"].shape[0] > 10_000 and ply["features"].shape[1:] == (16, 3)      # densified beyond the 10k initial points
    assert bool(torch.isfinite(ply["xyz"]).all())
