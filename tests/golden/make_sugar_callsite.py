#!/usr/bin/env python
"""Golden fixture from the UNMODIFIED reference caller: sugar_scene/sugar_model.py is imported from /root/reference (with
the stand-in `pytorch3d` of sugar_amd.shims and this repository's `diff_gaussian_rasterization` / `simple_knn` packages on
the path), a SuGaR model is built and `SuGaR.render_image_gaussian_rasterizer` (:2085-2294) is run on the CPU.  Test-only
substitutions, none of which touch the reference's files:
  * the rasterizer class seen by sugar_model is the CPU-oracle-backed stand-in (tests/oracle_rasterizer.py) wrapped in a
    recorder, so the fixture holds exactly what SuGaR hands to the rasterizer boundary and what comes back;
  * `knn_points` is an exact scipy cKDTree stand-in (the HIP k-NN has no CPU path);
  * `Tensor.cuda()` is the identity (the caller hard-codes `.cuda()`, :2143-2150); `open3d` / `plyfile` are empty modules;
  * the cameras are duck-typed (CamerasWrapper needs pytorch3d's camera classes, which are out of scope).
The GPU test replays the recorded boundary inputs through the HIP rasterizer (tests/test_gpu_sugar_callsite.py); the CPU test
re-runs this script's `run()` when /root/reference is present (tests/test_shims.py).

    python tests/golden/make_sugar_callsite.py      -> tests/golden/sugar_callsite.npz
"""
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("SUGAR_REFERENCE", "/root/reference")
W, H, P = 200, 152, 3000


def _import_reference_model():
    for p in (os.path.join(REF, "gaussian_splatting"), REF, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    from sugar_amd import shims
    shims.install()
    for name in ("open3d", "plyfile"):
        if name not in sys.modules:
            try:
                __import__(name)
            except ImportError:
                m = types.ModuleType(name)
                m.PlyData = m.PlyElement = object
                sys.modules[name] = m
    import sugar_scene.sugar_model as sm
    assert os.path.abspath(sm.__file__).startswith(os.path.abspath(REF)), sm.__file__
    return sm


def _scipy_knn_points(p1, p2, K=1, **_):
    from scipy.spatial import cKDTree
    from sugar_amd.knn import _KNN
    d, i = cKDTree(p2[0].detach().double().numpy()).query(p1[0].detach().double().numpy(), k=K)
    d = np.asarray(d).reshape(p1.shape[1], K); i = np.asarray(i).reshape(p1.shape[1], K)
    return _KNN(torch.as_tensor(d ** 2, dtype=torch.float32)[None], torch.as_tensor(i, dtype=torch.int64)[None], None)


class _P3DCamera:
    def __init__(self, center):
        self.znear = torch.tensor([0.01]); self.zfar = torch.tensor([100.0])
        self.K = torch.zeros(1, 4, 4)
        self.K[0, 0, 0] = self.K[0, 1, 1] = 1.7
        self._c = center

    def get_camera_center(self):
        return self._c.reshape(1, 3)


class _P3DCameras(list):
    @property
    def K(self):
        return torch.cat([c.K for c in self])


class _Cameras:
    """the attributes of CamerasWrapper that SuGaR.__init__ and the renderer read"""
    def __init__(self, cams):
        n = len(cams)
        self.height = torch.tensor([cams[0].image_height] * n); self.width = torch.tensor([cams[0].image_width] * n)
        fx = cams[0].image_width / (2.0 * cams[0].tanfovx); fy = cams[0].image_height / (2.0 * cams[0].tanfovy)
        self.fx = torch.tensor([fx] * n); self.fy = torch.tensor([fy] * n)
        c2ws = []
        for c in cams:
            w2c = c.viewmatrix.t().double().numpy()          # COLMAP-convention world-to-camera
            c2w = np.linalg.inv(w2c)
            c2w[:3, 1:3] *= -1                               # to the nerfstudio convention the caller expects (:2133-2134)
            c2ws.append(torch.tensor(c2w[:3], dtype=torch.float32))
        self.camera_to_worlds = torch.stack(c2ws)
        self.p3d_cameras = _P3DCameras(_P3DCamera(c2w[:, 3].clone()) for c2w in self.camera_to_worlds)

    def get_spatial_extent(self):
        return 3.0


class _Recorder:
    """wraps the rasterizer class the caller instantiates; keeps the boundary tensors of every call"""
    calls = []

    def __init__(self, raster_settings):
        from tests.oracle_rasterizer import GaussianRasterizer
        self.inner = GaussianRasterizer(raster_settings)
        self.settings = raster_settings

    def __call__(self, **kw):
        for k, v in list(kw.items()):
            if torch.is_tensor(v) and v.requires_grad:
                # an alias per input: its .grad is the gradient that crosses the boundary (a parameter such as the positions
                # also receives gradient along other paths, e.g. through the view-dependent colours)
                kw[k] = v.view_as(v)
                kw[k].retain_grad()
        image, radii = self.inner(**kw)
        _Recorder.calls.append(dict(settings=self.settings, inputs=kw, image=image, radii=radii))
        return image, radii


STATE = ("_points", "_scales", "_quaternions", "all_densities", "_sh_coordinates_dc", "_sh_coordinates_rest")


def model_state(model):
    """the model's raw parameters (+ its neighbour table): what a GPU run of the same reference class is loaded with, so that
    both start from bit-identical parameters (tests/test_gpu_reference_sugar.py)"""
    st = {"state" + n: getattr(model, n).detach().cpu().numpy().copy() for n in STATE}
    if getattr(model, "knn_idx", None) is not None:
        st["state_knn_idx"] = model.knn_idx.cpu().numpy().copy()
    return st


def run():
    """returns {name: np.ndarray}"""
    sm = _import_reference_model()
    from sugar_amd import synthetic as syn
    torch.manual_seed(0)
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    sm.knn_points = _scipy_knn_points
    sm.GaussianRasterizer = _Recorder
    _Recorder.calls = []
    try:
        cams = syn.orbit_cameras(W, H)
        nerf = types.SimpleNamespace(device=torch.device("cpu"), training_cameras=_Cameras(cams))
        g = torch.Generator().manual_seed(77)
        pts = (torch.rand(P, 3, generator=g) * 2 - 1) * 0.7
        cols = torch.rand(P, 3, generator=g)
        model = sm.SuGaR(nerfmodel=nerf, points=pts, colors=cols, initialize=True, sh_levels=4, keep_track_of_knn=True,
                         knn_to_track=16)
        with torch.no_grad():  # a mid-training state: the initial one has identity rotations and isotropic scales
            model._scales += 0.35 * torch.randn(P, 3, generator=g) + 0.6
            model._quaternions += 0.8 * torch.randn(P, 4, generator=g)
            model.all_densities += 2.0 * torch.randn(P, 1, generator=g) + 1.5
            model._sh_coordinates_rest += 0.15 * torch.randn(P, 15, 3, generator=g)
        wimg = torch.randn(H, W, 3, generator=g)
        out = {"W": np.int32(W), "H": np.int32(H), "dL_dimage_hw3": wimg.numpy()}
        out.update(model_state(model))
        bgs = [None, torch.tensor([1.0, 1.0, 1.0])]
        for ci, (cam_idx, in_rast) in enumerate(((1, False), (5, True))):
            model.zero_grad(set_to_none=True)
            res = model.render_image_gaussian_rasterizer(camera_indices=cam_idx, bg_color=bgs[ci], sh_deg=3,
                                                         compute_color_in_rasterizer=in_rast, return_2d_radii=True)
            (res["image"] * wimg).sum().backward()
            call = _Recorder.calls[-1]
            s = call["settings"]
            pre = f"c{ci}_"
            out[pre + "tanfov"] = np.array([s.tanfovx, s.tanfovy], dtype=np.float64)
            out[pre + "sh_degree"] = np.int32(s.sh_degree)
            for k in ("bg", "viewmatrix", "projmatrix", "campos"):
                out[pre + k] = getattr(s, k).detach().numpy().astype(np.float32)
            for k, v in call["inputs"].items():
                if v is None:
                    continue
                out[pre + "in_" + k] = v.detach().numpy()
                if v.grad is not None:
                    out[pre + "grad_" + k] = v.grad.detach().numpy()
            assert np.array_equal(out[pre + "grad_means2D"], res["viewspace_points"].grad.numpy())
            out[pre + "image_hw3"] = res["image"].detach().numpy()
            out[pre + "radii"] = res["radii"].numpy()
            # gradients on SuGaR's own parameters (through the caller's activations)
            for name in ("_points", "_scales", "_quaternions", "all_densities", "_sh_coordinates_dc", "_sh_coordinates_rest"):
                out[pre + "param_grad" + name] = getattr(model, name).grad.detach().numpy()
        return out
    finally:
        torch.Tensor.cuda = real_cuda


class _TriangleMesh:
    """the attributes of an open3d TriangleMesh that SuGaR.__init__ reads when binding (:160, :213-216)"""
    def __init__(self, vertices, triangles, vertex_colors):
        self.vertices, self.triangles, self.vertex_colors = vertices, triangles, vertex_colors


def _bumpy_sphere(n_lat=14, n_lon=24, seed=3):
    """a closed, consistently oriented triangle mesh: latitude rings between two poles, radius modulated"""
    rng = np.random.default_rng(seed)
    verts = [[0.0, 0.0, 1.0]]
    for i in range(1, n_lat):
        th = math.pi * i / n_lat
        for j in range(n_lon):
            ph = 2 * math.pi * j / n_lon
            verts.append([math.sin(th) * math.cos(ph), math.sin(th) * math.sin(ph), math.cos(th)])
    verts.append([0.0, 0.0, -1.0])
    v = np.asarray(verts)
    r = 0.6 + 0.08 * np.sin(3 * v[:, :1]) * np.cos(2 * v[:, 1:2]) + 0.01 * rng.standard_normal((len(v), 1))
    v = v * r
    ring = lambda i, j: 1 + (i - 1) * n_lon + (j % n_lon)
    tris = []
    for j in range(n_lon):
        tris.append([0, ring(1, j), ring(1, j + 1)])
        tris.append([len(v) - 1, ring(n_lat - 1, j + 1), ring(n_lat - 1, j)])
    for i in range(1, n_lat - 1):
        for j in range(n_lon):
            a, b, c, d = ring(i, j), ring(i, j + 1), ring(i + 1, j), ring(i + 1, j + 1)
            tris.append([a, c, d]); tris.append([a, d, b])
    return _TriangleMesh(v.astype(np.float64), np.asarray(tris, dtype=np.int64), rng.random((len(v), 3)))


def run_bound():
    """The refine-mode model (BASELINE.json config 4): Gaussians bound to the triangles of a surface mesh
    (SuGaR.__init__(surface_mesh_to_bind=...), :150-226, :326-350; positions from barycentric coordinates :384-397, flat
    scales :399-430, rotations from the face frame + a learned in-plane complex number :432-475), rendered through the same
    boundary.  `Meshes` / `TexturesVertex` are the stand-ins of sugar_amd.shims (parity-unpinned against pytorch3d)."""
    sm = _import_reference_model()
    from sugar_amd import synthetic as syn
    torch.manual_seed(0)
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    sm.knn_points = _scipy_knn_points
    sm.GaussianRasterizer = _Recorder
    _Recorder.calls = []
    try:
        cams = syn.orbit_cameras(W, H)
        nerf = types.SimpleNamespace(device=torch.device("cpu"), training_cameras=_Cameras(cams))
        mesh = _bumpy_sphere()
        g = torch.Generator().manual_seed(78)
        model = sm.SuGaR(nerfmodel=nerf, points=None, colors=None, initialize=False, sh_levels=4, keep_track_of_knn=False,
                         surface_mesh_to_bind=mesh, n_gaussians_per_surface_triangle=6, learn_surface_mesh_positions=True,
                         learn_surface_mesh_opacity=True, learn_surface_mesh_scales=True)
        n = model._n_points
        assert model.binded_to_surface_mesh and n == 6 * len(mesh.triangles)
        with torch.no_grad():  # a mid-refinement state
            model._scales += 0.3 * torch.randn(n, 2, generator=g) + 0.5
            model._quaternions += 0.7 * torch.randn(n, 2, generator=g)
            model.all_densities += 2.0 * torch.randn(n, 1, generator=g) + 2.5
            model._sh_coordinates_rest += 0.15 * torch.randn(n, 15, 3, generator=g)
        wimg = torch.randn(H, W, 3, generator=g)
        out = {"W": np.int32(W), "H": np.int32(H), "dL_dimage_hw3": wimg.numpy(), "n_faces": np.int32(len(mesh.triangles))}
        out.update(model_state(model))
        for ci, (cam_idx, in_rast) in enumerate(((2, False), (6, True))):
            model.zero_grad(set_to_none=True)
            res = model.render_image_gaussian_rasterizer(camera_indices=cam_idx, bg_color=None, sh_deg=3,
                                                         compute_color_in_rasterizer=in_rast, return_2d_radii=True)
            (res["image"] * wimg).sum().backward()
            call = _Recorder.calls[-1]
            s = call["settings"]
            pre = f"c{ci}_"
            out[pre + "tanfov"] = np.array([s.tanfovx, s.tanfovy], dtype=np.float64)
            out[pre + "sh_degree"] = np.int32(s.sh_degree)
            for k in ("bg", "viewmatrix", "projmatrix", "campos"):
                out[pre + k] = getattr(s, k).detach().numpy().astype(np.float32)
            for k, v in call["inputs"].items():
                if v is None:
                    continue
                out[pre + "in_" + k] = v.detach().numpy()
                if v.grad is not None:
                    out[pre + "grad_" + k] = v.grad.detach().numpy()
            out[pre + "image_hw3"] = res["image"].detach().numpy()
            out[pre + "radii"] = res["radii"].numpy()
            # gradients on the bound model's own parameters: mesh vertices, in-plane scales, in-plane rotation
            for name in ("_points", "_scales", "_quaternions", "all_densities", "_sh_coordinates_dc", "_sh_coordinates_rest"):
                out[pre + "param_grad" + name] = getattr(model, name).grad.detach().numpy()
        assert out["c0_in_scales"].shape == (n, 3) and float(out["c0_in_scales"][:, 0].max()) < 1e-5  # flat Gaussians
        return out
    finally:
        torch.Tensor.cuda = real_cuda


def main():
    for fn, name in ((run, "sugar_callsite.npz"), (run_bound, "sugar_callsite_bound.npz")):
        _write(fn(), name)


def _write(out, name):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")
    for k in sorted(out):
        if "image" in k or "radii" in k:
            print(" ", k, out[k].shape, float(np.asarray(out[k], dtype=np.float64).mean()))


if __name__ == "__main__":
    sys.exit(main())
