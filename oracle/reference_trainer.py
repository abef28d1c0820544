"""TEST INFRASTRUCTURE / DEMONSTRATION, NOT PRODUCT: the reference's own SuGaR trainer --
`sugar_trainers/coarse_sdf.py::coarse_training_with_sdf_regularization(args)`, every statement of it, its files untouched -- run
on this repository's drop-ins: `diff_gaussian_rasterization` and `simple_knn` (HIP), the `pytorch3d` / `plyfile` stand-ins
(sugar_amd.shims) and, optionally, the HIP routes of SuGaR's field methods (shims.install(patch_sugar=...)).

What it needs and the reference does not ship: a scene on disk.  `write_dataset` makes one in the layout the trainer reads
(sugar_scene/gs_model.py:69-160, sugar_scene/cameras.py:15-139):

    <root>/scene/images/<name>.png                                  ground-truth views (rendered here from a "true" surface)
    <root>/gs/cameras.json                                          camera_to_JSON records (camera_utils.py:62-82)
    <root>/gs/point_cloud/iteration_7000/point_cloud.ply            the "trained 3DGS" the trainer starts from: the true scene
                                                                    with perturbed colours, opacities and scales

The trainer's schedule is hard-coded in its body (15 000 iterations, starting at 6 999 when initialised from a trained 3DGS:
coarse_sdf.py:18-226,472-473: entropy regularisation from 7 000, pruning + SDF / normal regularisation with 1M samples from
9 000).  To stop a run early WITHOUT touching the file, the module-level name `ssim` the loop calls once per iteration
(coarse_sdf.py:11,459) is wrapped by a counter that raises after `stop_at`; the module-level name `Console` the function builds its rich console from
(coarse_sdf.py:13,18) is replaced by one that writes to a file so the loss lines it prints every 50 iterations can be read back.

Used by tests/test_gpu_reference_trainer.py and scripts/run_reference_trainer.py (the log committed under profiles/)."""
from __future__ import annotations

import json
import math
import os
import re
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class _Stop(Exception):
    pass


def surface_scene(P: int, seed: int = 0):
    """Flat Gaussians on a bumpy closed surface (radius about 0.8), colours a smooth function of position: something an SDF
    regulariser has a surface to find in.  Returns a sugar_amd.synthetic.Scene (activated values)."""
    from sugar_amd import synthetic as syn
    g = torch.Generator().manual_seed(seed)
    d = torch.randn(P, 3, generator=g)
    d = d / d.norm(dim=1, keepdim=True)
    means = d * surface_radius(d)[:, None]
    # tangent frame: normal ~ radial direction (good enough for a bumpy sphere); quaternion rotating z onto the normal
    n = d
    z = torch.tensor([0.0, 0.0, 1.0]).expand_as(n)
    axis = torch.linalg.cross(z, n)
    s = axis.norm(dim=1, keepdim=True).clamp_min(1e-8)
    ang = torch.atan2(s, (z * n).sum(1, keepdim=True))
    axis = axis / s
    quat = torch.cat([torch.cos(ang / 2), axis * torch.sin(ang / 2)], dim=1)
    quat = quat / quat.norm(dim=1, keepdim=True)
    spacing = 0.8 * math.sqrt(4 * math.pi / P)
    tang = spacing * (0.7 + 0.9 * torch.rand(P, 2, generator=g))
    scales = torch.cat([tang, 0.15 * spacing * torch.ones(P, 1)], dim=1)   # third axis (local z = normal) is the thin one
    opac = torch.sigmoid(2.0 + torch.randn(P, 1, generator=g))
    rgb = 0.5 + 0.35 * torch.stack([torch.sin(4 * means[:, 0]), torch.sin(5 * means[:, 1] + 1.0), torch.cos(3 * means[:, 2])], dim=1)
    shs = torch.zeros(P, 16, 3)
    shs[:, 0] = (rgb - 0.5) / syn.SH_C0
    shs[:, 1:4] = 0.03 * torch.randn(P, 3, 3, generator=g)
    return syn.Scene(means.contiguous(), scales.contiguous(), quat.contiguous(), opac.contiguous(), shs.contiguous())


def write_dataset(root: str, P: int = 60_000, n_cams: int = 48, W: int = 480, H: int = 320, seed: int = 0, device: str = "cuda:0"):
    """the on-disk scene described in the module docstring; GT images come from the HIP rasterizer (GPU needed)"""
    from PIL import Image
    from sugar_amd import io as sio, synthetic as syn
    from sugar_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    scene = surface_scene(P, seed)
    cams = syn.scattered_cameras(W, H, n=n_cams, seed=seed + 1)
    img_dir = os.path.join(root, "scene", "images")
    gs_dir = os.path.join(root, "gs")
    os.makedirs(img_dir, exist_ok=True)
    os.makedirs(gs_dir, exist_ok=True)
    dev = torch.device(device)
    records = []
    on_dev = {k: getattr(scene, k).to(dev) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    with torch.no_grad():
        for i, c in enumerate(cams):
            settings = GaussianRasterizationSettings(
                image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=torch.zeros(3, device=dev), scale_modifier=1.0,
                viewmatrix=c.viewmatrix.to(dev), projmatrix=c.projmatrix.to(dev), sh_degree=3, campos=c.campos.to(dev),
                prefiltered=False, debug=False)
            img, _ = GaussianRasterizer(settings)(
                means3D=on_dev["means3D"], means2D=torch.zeros(P, 3, device=dev), shs=on_dev["shs"], colors_precomp=None,
                opacities=on_dev["opacities"], scales=on_dev["scales"], rotations=on_dev["rotations"], cov3D_precomp=None)
            arr = (img.clamp(0, 1).permute(1, 2, 0).cpu().numpy() * 255.0 + 0.5).astype(np.uint8)
            name = f"view_{i:03d}"
            Image.fromarray(arr, "RGB").save(os.path.join(img_dir, name + ".png"))
            w2c = c.viewmatrix.t().double().numpy()
            records.append(sio.camera_to_json(i, name, w2c[:3, :3].T.copy(), w2c[:3, 3].copy(), 2 * math.atan(c.tanfovx),
                                              2 * math.atan(c.tanfovy), W, H))
    with open(os.path.join(gs_dir, "cameras.json"), "w") as f:
        json.dump(records, f)
    # the "trained 3DGS" checkpoint: the true scene, perturbed, as raw (pre-activation) parameters
    g = torch.Generator().manual_seed(seed + 2)
    shs = scene.shs.clone()
    shs[:, 0] += 0.25 * torch.randn(P, 3, generator=g)
    o = scene.opacities.clamp(1e-4, 1 - 1e-4)
    raw_o = torch.log(o / (1 - o)) + 0.5 * torch.randn(P, 1, generator=g)
    raw_s = torch.log(scene.scales) + 0.15 * torch.randn(P, 3, generator=g)
    sio.save_gaussian_ply(os.path.join(gs_dir, "point_cloud", "iteration_7000", "point_cloud.ply"),
                          scene.means3D, shs, raw_o, raw_s, scene.rotations)
    return types.SimpleNamespace(scene_path=os.path.join(root, "scene"), checkpoint_path=gs_dir + "/", n_cams=n_cams, P=P, W=W, H=H)


TRAINERS = {"coarse_sdf": "coarse_training_with_sdf_regularization",            # sugar_trainers/coarse_sdf.py:17
            "coarse_density": "coarse_training_with_density_regularization"}    # sugar_trainers/coarse_density.py:17


def import_trainer(patch_sugar: bool, trainer: str = "coarse_sdf"):
    """the reference's `sugar_trainers.<trainer>`, from /root/reference or the staged snapshot oracle/_ref/pysrc"""
    import importlib
    from tests import ref_env
    sm = ref_env.import_sugar_model(patch_sugar=patch_sugar)
    tr = importlib.import_module("sugar_trainers." + trainer)
    ref = ref_env.reference_root()
    assert os.path.abspath(tr.__file__).startswith(os.path.abspath(ref)), tr.__file__
    import diff_gaussian_rasterization as dgr
    import simple_knn
    for m in (dgr, simple_knn):
        assert os.path.abspath(m.__file__).startswith(ROOT), m.__file__   # the HIP drop-ins, not some other install
    return tr, sm


def _adam_stats():
    from sugar_amd import fused_adam
    return dict(fused_adam.STATS)


LOSS_LINE = re.compile(r"loss:\s*([-+0-9.eE]+|nan|inf)\s*\[\s*(\d+)/\s*(\d+)\]")


def run(data, out_dir: str, stop_at: int, patch_sugar: bool = True, log_path: str | None = None, patch_losses: bool = False,
        patch_optimizer: bool = False, trainer: str = "coarse_sdf", profile_window=None, patch_gathers: bool = False,
        patch_densifier: bool = False):
    """Runs the unmodified trainer on `data` until its iteration counter reaches `stop_at` (or 15 000).  Returns a dict with the
    (iteration, loss) pairs the trainer printed, the host time stamps of each iteration and the events it announced."""
    from rich.console import Console
    tr, sm = import_trainer(patch_sugar, trainer)
    if patch_gathers:
        from sugar_amd import sugar_patch as _sp
        _sp.install_row_gathers(sm)   # SuGaR.points / scaling / quaternions / get_normals(): row gathers with a HIP backward
    if patch_losses:
        from sugar_amd import shims
        shims.install_losses()   # the trainer's module-level `ssim` -> HIP loss kernels (before the counter wraps it)
    if patch_optimizer:
        from sugar_amd import shims as _shims
        _shims.install_optimizer()   # SuGaROptimizer's torch.optim.Adam -> FusedAdam
    if patch_densifier:
        from sugar_amd import shims as _dshims
        _dshims.install_densifier()  # SuGaRDensifier.update_densification_stats without boolean-mask indexing
    log_path = log_path or os.path.join(out_dir, "trainer_console.log")
    os.makedirs(out_dir, exist_ok=True)
    log_file = open(log_path, "w")
    saved = (tr.Console, tr.ssim)
    stamps = []
    first_iteration = 7000   # coarse_sdf.py:472-473 + the `iteration += 1` at :486

    prof = {"window": profile_window, "p": None, "table": None}

    def counted_ssim(*a, **k):
        stamps.append(time.time())
        it = first_iteration + len(stamps) - 1
        if prof["window"] is not None:      # torch profiler over [a, b): started / stopped at the one call per iteration
            if it == prof["window"][0]:
                from torch.profiler import profile, ProfilerActivity
                torch.cuda.synchronize()
                prof["p"] = profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA])
                prof["p"].__enter__()
            elif it == prof["window"][1] and prof["p"] is not None:
                torch.cuda.synchronize()
                prof["p"].__exit__(None, None, None)
                prof["table"] = prof["p"].key_averages().table(sort_by="cuda_time_total", row_limit=60, max_name_column_width=90)
                prof["p"] = None
        if it > stop_at:
            raise _Stop()
        return saved[1](*a, **k)

    tr.Console = lambda *a, **k: Console(file=log_file, width=200, force_terminal=False)
    tr.ssim = counted_ssim
    args = types.SimpleNamespace(gpu=0, scene_path=data.scene_path, checkpoint_path=data.checkpoint_path, iteration_to_load=7000,
                                 estimation_factor=0.2, normal_factor=0.2, output_dir=os.path.join(out_dir, "coarse"), eval=True,
                                 white_background=False)
    finished, model_path = False, None
    t0 = time.time()
    try:
        try:
            model_path = getattr(tr, TRAINERS[trainer])(args)
            finished = True
        except _Stop:
            pass
        torch.cuda.synchronize()
    finally:
        tr.Console, tr.ssim = saved
        log_file.close()
        if patch_losses:
            shims.uninstall_losses()
        if patch_optimizer:
            _shims.uninstall_optimizer()
        if patch_gathers:
            _sp.uninstall_row_gathers(sm)
        if patch_densifier:
            _dshims.uninstall_densifier()
        if patch_sugar:
            from sugar_amd import sugar_patch
            sugar_patch.uninstall(sm)
    wall = time.time() - t0
    text = open(log_path).read()
    losses = [(int(m.group(2)), float(m.group(1))) for m in LOSS_LINE.finditer(text)]
    events = [e for e in ("Starting entropy regularization", "Stopping entropy regularization", "Pruning gaussians with low-opacity",
                          "Starting SDF regularization", "Starting SDF estimation loss", "Starting SDF better normal loss",
                          "Resetting neighbors", "Training finished") if e in text]
    left = re.findall(r"Pruning finished: (\d+) gaussians left", text)
    its = np.asarray(stamps)
    def rate(lo, hi):
        a, b = lo - first_iteration, min(hi - first_iteration, len(its) - 1)
        return float((b - a) / (its[b] - its[a])) if b > a else None
    return dict(finished=finished, iterations_run=len(stamps) - (0 if finished else 1), last_iteration=first_iteration + len(stamps) - 2
                if not finished else 15_000, wall_s=wall, losses=losses, events=events,
                gaussians_after_pruning=int(left[-1]) if left else None,
                it_per_s_before_9000=rate(7050, 8950), it_per_s_after_9000=rate(9050, 15_000), log=log_path,
                patch_sugar=patch_sugar, patch_losses=patch_losses, patch_optimizer=patch_optimizer, patch_gathers=patch_gathers, model_path=model_path, trainer=trainer,
                profile_table=prof["table"], adam_stats=_adam_stats())


# ---------------------------------------------------------------------------------------------------------------------------
# The coarse-mesh extractor: sugar_extractors/coarse_mesh.py::extract_mesh_from_coarse_sugar(args), untouched.  Everything up to
# its first open3d call runs on the drop-ins: loading the 3DGS checkpoint and the coarse SuGaR model, pruning, building the splat
# mesh and pytorch3d's MeshRasterizer (coarse_mesh.py:203-225), and the loop over ALL training cameras (:243-327): an RGB render and
# `compute_level_surface_points_from_camera_fast(use_gaussian_depth=False, rasterizer=...)` per view, three level sets each.  The
# Poisson reconstruction that follows is open3d's (out of scope): the stand-in `open3d.geometry.PointCloud` raises, and the point
# clouds the loop accumulated are read out of the function's frame.
class _Open3DReached(Exception):
    pass


def _open3d_that_stops():
    o3d = types.ModuleType("open3d")
    o3d.geometry = types.ModuleType("open3d.geometry")
    o3d.utility = types.ModuleType("open3d.utility")
    o3d.io = types.ModuleType("open3d.io")

    def stop(*a, **k):
        raise _Open3DReached("open3d.geometry.PointCloud(): Poisson reconstruction is open3d's, outside this package")
    o3d.geometry.PointCloud = stop
    return o3d


def run_extractor(data, out_dir: str, coarse_model_path: str | None, patch_sugar: bool = True, log_path: str | None = None,
                  patch_gathers: bool = False):
    """Runs the unmodified extractor up to the Poisson step.  `coarse_model_path=None`: `--use_vanilla_3dgs True` (the model is
    built from the 3DGS checkpoint, coarse_mesh.py:141-165), else the `.pt` the coarse trainer saved.  Returns per level the
    sampled points / normals / Gaussian ids (GPU tensors) and the loop's wall time per camera."""
    from rich.console import Console
    from tests import ref_env
    sm = ref_env.import_sugar_model(patch_sugar=patch_sugar)
    if patch_gathers:
        from sugar_amd import sugar_patch as _sp
        _sp.install_row_gathers(sm)
    saved_o3d = sys.modules.get("open3d")
    sys.modules["open3d"] = _open3d_that_stops()
    sys.modules.pop("sugar_extractors.coarse_mesh", None)   # (it binds `o3d` at import)
    import sugar_extractors.coarse_mesh as ex
    assert os.path.abspath(ex.__file__).startswith(os.path.abspath(ref_env.reference_root())), ex.__file__
    os.makedirs(out_dir, exist_ok=True)
    log_path = log_path or os.path.join(out_dir, "extractor_console.log")
    log_file = open(log_path, "w")
    saved_console = ex.Console
    ex.Console = lambda *a, **k: Console(file=log_file, width=200, force_terminal=False)
    stamps = []
    orig_fast = sm.SuGaR.compute_level_surface_points_from_camera_fast

    def timed_fast(self, *a, **k):
        stamps.append(time.time())
        return orig_fast(self, *a, **k)
    sm.SuGaR.compute_level_surface_points_from_camera_fast = timed_fast
    args = types.SimpleNamespace(scene_path=data.scene_path, checkpoint_path=data.checkpoint_path, iteration_to_load=7000, eval=True,
                                 coarse_model_path=coarse_model_path, surface_level=None, decimation_target=None,
                                 mesh_output_dir=os.path.join(out_dir, "coarse_mesh"), bboxmin=None, bboxmax=None, center_bbox=True,
                                 use_centers_to_extract_mesh=False, use_marching_cubes=False, use_vanilla_3dgs=coarse_model_path is None,
                                 gpu=0)
    outputs, reached = None, False
    # The trainer's checkpoint holds numpy scalars next to the state dict (sugar_model.py:2296-2301); the reference pins torch 2.0.1
    # (environment.yml), whose torch.load unpickled them -- torch >= 2.6 refuses unless told otherwise.  An environment setting of
    # this harness, not a change to the extractor.
    saved_env = os.environ.get("TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD")
    os.environ["TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD"] = "1"
    t0 = time.time()
    try:
        try:
            ex.extract_mesh_from_coarse_sugar(args)
        except _Open3DReached as e:
            reached = True
            tb = e.__traceback__
            while tb is not None:
                if tb.tb_frame.f_code.co_name == "extract_mesh_from_coarse_sugar":
                    outputs = tb.tb_frame.f_locals.get("surface_levels_outputs")
                tb = tb.tb_next
        torch.cuda.synchronize()
    finally:
        t1 = time.time()
        sm.SuGaR.compute_level_surface_points_from_camera_fast = orig_fast
        if saved_env is None:
            os.environ.pop("TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD", None)
        else:
            os.environ["TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD"] = saved_env
        ex.Console = saved_console
        log_file.close()
        if saved_o3d is not None:
            sys.modules["open3d"] = saved_o3d
        else:
            sys.modules.pop("open3d", None)
        sys.modules.pop("sugar_extractors.coarse_mesh", None)
        if patch_gathers:
            _sp.uninstall_row_gathers(sm)
        if patch_sugar:
            from sugar_amd import sugar_patch
            sugar_patch.uninstall(sm)
    n = len(stamps)
    per_cam = (t1 - stamps[1]) / (n - 1) if n > 2 else None   # (the first camera also resets the neighbours and warms allocations)
    return dict(reached_poisson=reached, cameras=n, wall_s=t1 - t0, ms_per_camera=None if per_cam is None else 1e3 * per_cam,
                outputs=outputs, log=log_path, patch_sugar=patch_sugar)


def surface_radius(d):
    """the radius of surface_scene's surface in unit direction d [N,3]"""
    return 0.8 + 0.12 * torch.sin(3.0 * d[:, 0]) * torch.cos(2.0 * d[:, 1]) + 0.08 * torch.sin(5.0 * d[:, 2])


# ---------------------------------------------------------------------------------------------------------------------------
# The refinement trainer: sugar_trainers/refine.py::refined_training(args), untouched -- the surface-bound model of BASELINE config 4:
# Gaussians tied to the triangles of a mesh (six per triangle, flat, their positions / scales / rotations functions of the mesh
# vertices and two in-plane parameters), optimised through the rasterizer with pytorch3d's mesh_normal_consistency on the mesh.  It
# gets its mesh from `open3d.io.read_triangle_mesh` (open3d is absent: the harness's stand-in reads the arrays this file wrote) and
# ends by itself after `args.refinement_iterations`; at the end it exports the Gaussians through GaussianModel.save_ply (plyfile
# stand-in).
def icosphere(level: int):
    """unit icosphere: (vertices [V,3] float64, faces [F,3] int64), F = 20 * 4**level, outward orientation"""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1),
         (-t, 0, -1), (-t, 0, 1)]
    verts = [np.asarray(p, dtype=np.float64) / np.linalg.norm(p) for p in v]
    faces = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
             (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(level):
        cache, out = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = verts[a] + verts[b]
                verts.append(m / np.linalg.norm(m))
                cache[key] = len(verts) - 1
            return cache[key]
        for a, b, c in faces:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            out += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        faces = out
    return np.stack(verts), np.asarray(faces, dtype=np.int64)


def write_surface_mesh(path: str, level: int = 5):
    """the surface of `surface_scene` as a triangle mesh with vertex colours (arrays in an .npz: what the open3d stand-in reads)"""
    d, faces = icosphere(level)
    dt = torch.from_numpy(d).float()
    verts = (dt * surface_radius(dt)[:, None]).numpy().astype(np.float64)
    rgb = 0.5 + 0.35 * np.stack([np.sin(4 * verts[:, 0]), np.sin(5 * verts[:, 1] + 1.0), np.cos(3 * verts[:, 2])], axis=1)
    np.savez(path, vertices=verts, triangles=faces, vertex_colors=rgb)
    return faces.shape[0]


def _open3d_that_reads_npz():
    o3d = types.ModuleType("open3d")
    o3d.io = types.ModuleType("open3d.io")

    def read_triangle_mesh(path, *a, **k):
        z = np.load(path)
        return types.SimpleNamespace(vertices=z["vertices"], triangles=z["triangles"], vertex_colors=z["vertex_colors"],
                                     vertex_normals=z["vertices"] / np.linalg.norm(z["vertices"], axis=1, keepdims=True))
    o3d.io.read_triangle_mesh = read_triangle_mesh
    return o3d


def run_refine(data, out_dir: str, iterations: int = 400, gaussians_per_triangle: int = 6, mesh_level: int = 5,
               patch_sugar: bool = True, patch_losses: bool = False, patch_optimizer: bool = False, log_path: str | None = None,
               patch_gathers: bool = False):
    import importlib
    from rich.console import Console
    from tests import ref_env
    from sugar_amd import shims
    sm = ref_env.import_sugar_model(patch_sugar=patch_sugar)
    saved_o3d = sys.modules.get("open3d")
    sys.modules["open3d"] = _open3d_that_reads_npz()
    sys.modules.pop("sugar_trainers.refine", None)   # (it binds `o3d` at import)
    tr = importlib.import_module("sugar_trainers.refine")
    assert os.path.abspath(tr.__file__).startswith(os.path.abspath(ref_env.reference_root())), tr.__file__
    if patch_losses:
        shims.install_losses()
    if patch_optimizer:
        shims.install_optimizer()
    if patch_gathers:
        from sugar_amd import sugar_patch as _sp
        _sp.install_row_gathers(sm)
    os.makedirs(out_dir, exist_ok=True)
    out_dir = os.path.abspath(out_dir)
    mesh_path = os.path.join(out_dir, "surface_mesh.npz")
    n_faces = write_surface_mesh(mesh_path, mesh_level)
    log_path = os.path.abspath(log_path or os.path.join(out_dir, "refine_console.log"))
    log_file = open(log_path, "w")
    saved = (tr.Console, tr.ssim)
    stamps = []

    def counted_ssim(*a, **k):
        stamps.append(time.time())
        return saved[1](*a, **k)
    tr.Console = lambda *a, **k: Console(file=log_file, width=200, force_terminal=False)
    tr.ssim = counted_ssim
    args = types.SimpleNamespace(gpu=0, scene_path=data.scene_path, checkpoint_path=data.checkpoint_path, mesh_path=mesh_path,
                                 iteration_to_load=7000, normal_consistency_factor=0.1, gaussians_per_triangle=gaussians_per_triangle,
                                 n_vertices_in_fg=1_000_000, refinement_iterations=iterations, bboxmin=None, bboxmax=None,
                                 output_dir=os.path.join("output", "refined", "scene"), eval=True, white_background=False,
                                 export_ply=True)
    # (a RELATIVE output directory, as the reference's command line gives it: the export path at refine.py:880-885 is rebuilt with
    # os.path.join(*model_path.split(os.sep)), which drops the root of an absolute path)
    model_path, t0, cwd = None, time.time(), os.getcwd()
    os.chdir(out_dir)
    try:
        model_path = tr.refined_training(args)
        torch.cuda.synchronize()
    finally:
        wall = time.time() - t0
        os.chdir(cwd)
        tr.Console, tr.ssim = saved
        log_file.close()
        if patch_losses:
            shims.uninstall_losses()
        if patch_optimizer:
            shims.uninstall_optimizer()
        if patch_gathers:
            _sp.uninstall_row_gathers(sm)
        if saved_o3d is not None:
            sys.modules["open3d"] = saved_o3d
        else:
            sys.modules.pop("open3d", None)
        sys.modules.pop("sugar_trainers.refine", None)
        if patch_sugar:
            from sugar_amd import sugar_patch
            sugar_patch.uninstall(sm)
    text = open(log_path).read()
    losses = [(int(m.group(2)), float(m.group(1))) for m in LOSS_LINE.finditer(text)]
    its = np.asarray(stamps)
    rate = float((len(its) - 51) / (its[-1] - its[50])) if len(its) > 60 else None
    ply = None
    for root, _, files in os.walk(os.path.join(out_dir, "output")):
        for f in files:
            if f.endswith(".ply"):
                ply = os.path.join(root, f)
    return dict(model_path=model_path, iterations_run=len(stamps), wall_s=wall, losses=losses, it_per_s=rate, mesh_faces=n_faces,
                gaussians=n_faces * gaussians_per_triangle, exported_ply=ply, finished="Final model saved" in text, log=log_path,
                patch_sugar=patch_sugar, patch_losses=patch_losses, patch_optimizer=patch_optimizer)


# ---------------------------------------------------------------------------------------------------------------------------
# The vanilla 3DGS trainer as a command line: gaussian_splatting/train.py, untouched, started the way a user would start it on this
# stack -- `python -m sugar_amd.launch <reference>/gaussian_splatting/train.py -s <scene> -m <out> --iterations N ...` -- on a scene
# in the COLMAP text layout its `Scene` class reads: cameras.txt / images.txt / points3D.txt + PNG views; the reader converts the
# points to sparse/0/points3D.ply through `storePly` and reads them back with `fetchPly` (plyfile stand-in).  (The reader of the
# NeRF-synthetic layout hands an int8 array to PIL with an explicit mode, dataset_readers.py:210, which the Pillow of this image
# rejects -- a version matter of the reference, not of this stack.)
# Everything the script does runs: `Scene` / camera loading, `create_from_pcd` (distCUDA2), the loop of train.py:69-128 with its
# densification and pruning (`densify_and_prune`: optimiser state cut and concatenated -- on the FusedAdam instance when
# patch_optimizer is on), opacity reset, the evaluation passes, `scene.save` (GaussianModel.save_ply).
def _rotmat_to_qvec(R):
    """unit quaternion (w, x, y, z) of a rotation matrix (the inverse of scene/colmap_loader.py:43-53 qvec2rotmat)"""
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = math.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    return q / np.linalg.norm(q)


def write_colmap_dataset(root: str, P: int = 60_000, n_cams: int = 48, W: int = 480, H: int = 320, n_sfm_points: int = 20_000, seed: int = 0,
                         device: str = "cuda:0"):
    """a scene in the COLMAP text layout `readColmapSceneInfo` reads (scene/dataset_readers.py:132-177, scene/colmap_loader.py:83-130,
    156-178,244-270): images/<name>.png, sparse/0/cameras.txt (one PINHOLE camera), images.txt (world-to-camera quaternion and
    translation per view), points3D.txt (a sparse, slightly noisy sample of the true surface: what SfM would leave)"""
    from PIL import Image
    from sugar_amd import synthetic as syn
    from sugar_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    scene = surface_scene(P, seed)
    cams = syn.scattered_cameras(W, H, n=n_cams, seed=seed + 1)
    dev = torch.device(device)
    on_dev = {k: getattr(scene, k).to(dev) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    os.makedirs(os.path.join(root, "images"), exist_ok=True)
    os.makedirs(os.path.join(root, "sparse", "0"), exist_ok=True)
    fx, fy = W / (2 * cams[0].tanfovx), H / (2 * cams[0].tanfovy)
    with open(os.path.join(root, "sparse", "0", "cameras.txt"), "w") as f:
        f.write("# Camera list with one line of data per camera:\n#   CAMERA_ID, MODEL, WIDTH, HEIGHT, PARAMS[]\n")
        f.write(f"1 PINHOLE {W} {H} {float(fx)!r} {float(fy)!r} {W / 2} {H / 2}\n")
    lines = ["# Image list with two lines of data per image:\n#   IMAGE_ID, QW, QX, QY, QZ, TX, TY, TZ, CAMERA_ID, NAME\n#   POINTS2D[] as (X, Y, POINT3D_ID)\n"]
    with torch.no_grad():
        for i, c in enumerate(cams):
            settings = GaussianRasterizationSettings(
                image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=torch.zeros(3, device=dev), scale_modifier=1.0,
                viewmatrix=c.viewmatrix.to(dev), projmatrix=c.projmatrix.to(dev), sh_degree=3, campos=c.campos.to(dev),
                prefiltered=False, debug=False)
            img, _ = GaussianRasterizer(settings)(
                means3D=on_dev["means3D"], means2D=torch.zeros(P, 3, device=dev), shs=on_dev["shs"], colors_precomp=None,
                opacities=on_dev["opacities"], scales=on_dev["scales"], rotations=on_dev["rotations"], cov3D_precomp=None)
            arr = (img.clamp(0, 1).permute(1, 2, 0).cpu().numpy() * 255.0 + 0.5).astype(np.uint8)
            name = f"view_{i:03d}.png"
            Image.fromarray(arr, "RGB").save(os.path.join(root, "images", name))
            w2c = c.viewmatrix.t().double().numpy()
            q, t = _rotmat_to_qvec(w2c[:3, :3]), w2c[:3, 3]
            nums = " ".join(repr(float(v)) for v in (*q, *t))
            lines.append(f"{i + 1} {nums} 1 {name}\n\n")
    with open(os.path.join(root, "sparse", "0", "images.txt"), "w") as f:
        f.writelines(lines)
    g = torch.Generator().manual_seed(seed + 3)
    pick = torch.randperm(P, generator=g)[:n_sfm_points]
    xyz = (scene.means3D[pick] + 0.004 * torch.randn(len(pick), 3, generator=g)).numpy()
    rgb = (scene.shs[pick, 0] * syn.SH_C0 + 0.5).clamp(0, 1).numpy()
    with open(os.path.join(root, "sparse", "0", "points3D.txt"), "w") as f:
        f.write("# 3D point list with one line of data per point:\n#   POINT3D_ID, X, Y, Z, R, G, B, ERROR, TRACK[] as (IMAGE_ID, POINT2D_IDX)\n")
        for k in range(len(pick)):
            f.write(f"{k + 1} {float(xyz[k, 0])!r} {float(xyz[k, 1])!r} {float(xyz[k, 2])!r} {int(rgb[k, 0] * 255)} {int(rgb[k, 1] * 255)} "
                    f"{int(rgb[k, 2] * 255)} 0.5\n")
    return root


def run_vanilla_cli(dataset_dir: str, out_dir: str, iterations: int = 1000, launcher_flags=(), extra_args=(), timeout: int = 900):
    """`python -m sugar_amd.launch [launcher_flags] <reference>/gaussian_splatting/train.py -s dataset -m out --iterations N --eval ...`
    in a subprocess.  Returns the exit code, the text it printed, the evaluation lines and the saved point cloud's path."""
    import socket
    import subprocess
    from tests import ref_env
    ref = ref_env.reference_root()
    script = os.path.join(ref, "gaussian_splatting", "train.py")
    with socket.socket() as sock:                  # a free port for the script's network GUI listener (train.py:214)
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "sugar_amd.launch", *launcher_flags, script, "-s", dataset_dir, "-m", out_dir, "--iterations", str(iterations),
           "--eval", "--test_iterations", str(iterations // 2), str(iterations), "--save_iterations", str(iterations), "--port", str(port),
           *extra_args]
    t0 = time.time()
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=dict(os.environ, PYTHONPATH=""))
    wall = time.time() - t0
    text = p.stdout + "\n" + p.stderr
    evals = [(int(m.group(1)), m.group(2), float(m.group(3)), float(m.group(4)))
             for m in re.finditer(r"\[ITER (\d+)\] Evaluating (\w+): L1 ([-+0-9.eE]+) PSNR ([-+0-9.eE]+)", text)]
    ply = os.path.join(out_dir, "point_cloud", f"iteration_{iterations}", "point_cloud.ply")
    return dict(returncode=p.returncode, wall_s=wall, evals=evals, complete="Training complete." in text, ply=ply if os.path.exists(ply) else None,
                launch_line=next((l for l in text.splitlines() if l.startswith("[sugar_amd.launch]")), None), text=text, cmd=cmd)
