"""CPU tests of the mesh z-buffer ORACLE (oracle/mesh_rasterizer.c) and of the host side of the stand-in
`pytorch3d.renderer.MeshRasterizer` (sugar_amd/shims/pytorch3d/renderer/mesh, run here on the oracle backend).

pytorch3d is absent, so the oracle is PARITY-UNPINNED against it; these tests pin what can be pinned without it:
  * geometry: the faces named for a pixel are exactly the K nearest faces whose projection contains the pixel centre, in
    ascending perspective-correct depth (independent float64 brute force);
  * conventions: a tiny world-space triangle lands on the pixel the GAUSSIAN rasterizer's camera projects its centre to (the
    sampler mixes both: it unprojects the mesh depth map and splats with the Gaussian rasterizer);
  * near-plane clipping: faces crossing z = znear / 2 render as the part in front of the plane, with the ORIGINAL face index and
    barycentric coordinates w.r.t. the original corners (float64 ray casting against the unclipped 3-D triangles);
  * batches, defaults, error behaviour of the host code."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import mesh_oracle as mo
from tests import mesh_scenes as ms
from tests.mesh_backend import oracle_mesh_rasterizer

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def p3d():
    from sugar_amd import shims
    shims.install()
    import pytorch3d.renderer as r
    import pytorch3d.structures as s
    return r, s


def _pix_ndc(H, W):
    xs = np.array([mo.pix_to_ndc(W - 1 - c, W, H) for c in range(W)], dtype=np.float64)
    ys = np.array([mo.pix_to_ndc(H - 1 - r, H, W) for r in range(H)], dtype=np.float64)
    return xs, ys


def test_pixel_centres_follow_the_non_square_ndc_convention():
    # the shorter side spans [-1, 1], centres at half-pixel offsets, +x to the LEFT and +y UP
    H, W = 90, 160
    xs, ys = _pix_ndc(H, W)
    assert np.allclose(ys, 1 - (2 * np.arange(H) + 1) / H, atol=1e-6)
    assert np.allclose(xs, (W / H) * (1 - (2 * np.arange(W) + 1) / W), atol=1e-6)
    xs, ys = _pix_ndc(W, H)  # portrait
    assert np.allclose(xs, 1 - (2 * np.arange(H) + 1) / H, atol=1e-6)
    assert np.allclose(ys, (W / H) * (1 - (2 * np.arange(W) + 1) / W), atol=1e-6)


@pytest.mark.parametrize("H,W,K,F,seed,persp", [(48, 64, 4, 600, 1, True), (64, 48, 10, 900, 2, True), (40, 40, 3, 400, 3, False)])
def test_oracle_names_the_k_nearest_covering_faces(H, W, K, F, seed, persp):
    fv = ms.soup(F, seed, size=0.15)
    p2f, zbuf, bary, dists = mo.rasterize_meshes_naive(fv, (H, W), 0.0, K, persp)
    v = fv.astype(np.float64)
    xs, ys = _pix_ndc(H, W)
    X, Y = np.meshgrid(xs, ys)                                          # [H, W]
    X, Y = X[..., None], Y[..., None]

    def edge(a, b):
        return (X - v[:, a, 0]) * (v[:, b, 1] - v[:, a, 1]) - (Y - v[:, a, 1]) * (v[:, b, 0] - v[:, a, 0])
    area = (v[:, 2, 0] - v[:, 0, 0]) * (v[:, 1, 1] - v[:, 0, 1]) - (v[:, 2, 1] - v[:, 0, 1]) * (v[:, 1, 0] - v[:, 0, 0])
    w = np.stack([edge(1, 2), edge(2, 0), edge(0, 1)], -1) / area[:, None]   # [H, W, F, 3]
    margin = w.min(-1)
    cover = margin > 0
    ambiguous = (np.abs(margin) < 1e-5).any(-1)                          # a pixel centre within rounding of some edge
    z = v[:, :, 2]
    if persp:
        depth = w.sum(-1) / (w / z).sum(-1)
    else:
        depth = (w * z).sum(-1)
    depth = np.where(cover, depth, np.inf)
    order = np.argsort(depth, axis=-1, kind="stable")[..., :K]
    want_z = np.take_along_axis(depth, order, -1)
    want_f = np.where(np.isfinite(want_z), order, -1)
    ok = ~ambiguous
    assert ok.mean() > 0.95 and (want_f[..., 0] >= 0).mean() > 0.3
    assert np.array_equal(p2f[ok], want_f[ok])
    filled = want_f >= 0
    assert np.allclose(zbuf[ok][filled[ok]], want_z[ok][filled[ok]], rtol=1e-4)
    assert (zbuf[ok][~filled[ok]] == -1).all() and (dists[ok][~filled[ok]] == -1).all() and (bary[ok][~filled[ok]] == -1).all()
    # barycentric coordinates reproduce the depth and sum to one; inside a face the signed distance is negative
    bz = (bary * np.where(p2f[..., None] >= 0, v[np.maximum(p2f, 0)][..., 2], 0)).sum(-1)
    assert np.allclose(bz[p2f >= 0], zbuf[p2f >= 0], rtol=1e-4)
    assert np.allclose(bary[p2f >= 0].sum(-1), 1, atol=(1e-5 if persp else 1e-3)) and (dists[p2f >= 0] <= 0).all()


def test_tiny_faces_land_on_the_pixel_the_gaussian_camera_projects_to(p3d):
    """sugar_scene/cameras.py:262-326 builds the pytorch3d camera from the Gaussian-splatting one; a tiny triangle around a 3-D
    point must cover the pixel ndc2Pix (DGR/cuda_rasterizer/auxiliary.h:41-44) puts the point in"""
    R, S = p3d
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from make_sugar_field import p3d_cameras_like_the_reference
    from sugar_amd import synthetic as syn
    W, H = 120, 88
    cams = syn.orbit_cameras(W, H)
    pc = p3d_cameras_like_the_reference(cams)
    g = torch.Generator().manual_seed(0)
    pts = (torch.rand(300, 3, generator=g) * 2 - 1) * 0.6
    for ci in (0, 3):
        cam = cams[ci]
        ph = torch.cat([pts, torch.ones(300, 1)], 1) @ cam.projmatrix          # row vectors (auxiliary.h:68-77)
        ndc = ph[:, :2] / (ph[:, 3:4] + 1e-7)
        pix = ((ndc + 1) * torch.tensor([W, H]) - 1) / 2
        depth = (torch.cat([pts, torch.ones(300, 1)], 1) @ cam.viewmatrix)[:, 2]
        col, row = pix[:, 0].round().long(), pix[:, 1].round().long()
        inside = (col >= 0) & (col < W) & (row >= 0) & (row < H) & ((pix - pix.round()).abs().max(1).values < 0.3)
        # one small camera-facing triangle per point (in view space, mapped back to the world)
        w2v = pc[ci].get_world_to_view_transform()
        pv = w2v.transform_points(pts)
        r = 1.3 * depth / (W / (2 * cam.tanfovx))                              # circumradius 1.3 px: inradius 0.65 px
        tri = torch.stack([pv + torch.stack([r, torch.zeros_like(r), torch.zeros_like(r)], 1),
                           pv + torch.stack([-0.5 * r, 0.87 * r, torch.zeros_like(r)], 1),
                           pv + torch.stack([-0.5 * r, -0.87 * r, torch.zeros_like(r)], 1)], 1)
        verts = w2v.inverse().transform_points(tri.reshape(-1, 3))
        faces = torch.arange(900).reshape(300, 3)
        rast = R.MeshRasterizer(cameras=pc[ci], raster_settings=R.RasterizationSettings(image_size=(H, W), faces_per_pixel=4))
        with oracle_mesh_rasterizer():
            fr = rast(S.Meshes(verts=[verts], faces=[faces]))
        assert fr.pix_to_face.shape == (1, H, W, 4) and fr.zbuf.shape == (1, H, W, 4) and fr.bary_coords.shape == (1, H, W, 4, 3)
        hit = 0
        for i in inside.nonzero()[:, 0].tolist():
            stack = fr.pix_to_face[0, row[i], col[i]].tolist()
            assert i in stack, (ci, i, stack)
            k = stack.index(i)
            assert abs(float(fr.zbuf[0, row[i], col[i], k]) - float(depth[i])) < 1e-4 * float(depth[i])
            hit += 1
        assert hit > 40


def _ray_cast(tri_view, H, W, fx, fy, z_clip):
    """float64: nearest intersection per pixel of the pixel's view ray with 3-D triangles (view space, pytorch3d axes), keeping
    only hits at depth >= z_clip.  Returns (face, depth, barycentric) per pixel; face -1 where nothing is hit."""
    xs, ys = _pix_ndc(H, W)
    X, Y = np.meshgrid(xs, ys)
    d = np.stack([X / fx, Y / fy, np.ones_like(X)], -1)                          # direction with unit z
    best_f = -np.ones((H, W), np.int64); best_z = np.full((H, W), np.inf); best_b = np.zeros((H, W, 3))
    amb = np.zeros((H, W), bool)
    for f, t in enumerate(tri_view):
        e1, e2 = t[1] - t[0], t[2] - t[0]
        n = np.cross(e1, e2)
        den = d @ n
        with np.errstate(divide="ignore", invalid="ignore"):
            s = (t[0] @ n) / den                                                 # depth along the ray (z of the hit)
            p = d * s[..., None]
            # barycentric coordinates of p in the triangle
            q = p - t[0]
            d11, d12, d22 = e1 @ e1, e1 @ e2, e2 @ e2
            q1, q2 = q @ e1, q @ e2
            det = d11 * d22 - d12 * d12
            b1 = (d22 * q1 - d12 * q2) / det
            b2 = (d11 * q2 - d12 * q1) / det
        b0 = 1 - b1 - b2
        mn = np.minimum(np.minimum(b0, b1), b2)
        ok = (mn > 0) & (s >= z_clip) & np.isfinite(s)
        amb |= (np.abs(mn) < 1e-4) | (np.abs(s - z_clip) < 1e-4 * z_clip)
        upd = ok & (s < best_z)
        best_f[upd] = f; best_z[upd] = s[upd]
        best_b[upd] = np.stack([b0, b1, b2], -1)[upd]
    return best_f, best_z, best_b, amb


def test_faces_crossing_the_near_plane_are_clipped_and_mapped_back(p3d):
    R, S = p3d
    H, W = 60, 80
    fx = fy = 1.4
    from pytorch3d.renderer.cameras import _get_sfm_calibration_matrix
    K = _get_sfm_calibration_matrix(1, "cpu", torch.tensor([[fx, fy]]), torch.zeros(1, 2))
    znear = 0.4
    cam = R.FoVPerspectiveCameras(R=torch.eye(3)[None], T=torch.zeros(1, 3), K=K, znear=znear)
    rng = np.random.default_rng(4)
    n = 24
    c = np.stack([rng.uniform(-0.4, 0.4, n), rng.uniform(-0.3, 0.3, n), rng.uniform(0.15, 0.6, n)], -1)
    tri = c[:, None] + rng.normal(0, 1, (n, 3, 3)) * np.array([0.25, 0.25, 0.3])
    behind = (tri[..., 2] < znear / 2).sum(1)
    assert set(behind.tolist()) >= {0, 1, 2}                                     # every clipping case occurs
    verts = torch.tensor(tri.reshape(-1, 3), dtype=torch.float32)
    faces = torch.arange(3 * n).reshape(n, 3)
    rast = R.MeshRasterizer(cameras=cam, raster_settings=R.RasterizationSettings(image_size=(H, W), faces_per_pixel=1))
    with oracle_mesh_rasterizer():
        fr = rast(S.Meshes(verts=[verts], faces=[faces]))
    f_ref, z_ref, b_ref, amb = _ray_cast(tri, H, W, fx, fy, znear / 2)
    got_f = fr.pix_to_face[0, ..., 0].numpy()
    ok = ~amb
    assert ok.mean() > 0.9 and (f_ref[ok] >= 0).mean() > 0.2
    assert np.array_equal(got_f[ok], f_ref[ok])
    m = ok & (f_ref >= 0)
    assert np.allclose(fr.zbuf[0, ..., 0].numpy()[m], z_ref[m], rtol=2e-4)
    # barycentric coordinates are those of the ORIGINAL corners: they reproduce the depth of the hit
    bz = (fr.bary_coords[0, ..., 0, :].numpy() * tri[np.maximum(got_f, 0)][..., 2]).sum(-1)
    pc = fr.bary_coords[0, ..., 0, :].numpy()
    assert np.allclose(pc[m], b_ref[m], atol=5e-3)
    hits_of_clipped = np.isin(got_f[m], np.nonzero(behind > 0)[0]).sum()
    assert hits_of_clipped > 30                                                  # the clipped faces are actually seen
    # for unclipped faces the depth is the barycentric combination of the corner depths (pz = sum b_i z_i)
    front = m & np.isin(got_f, np.nonzero(behind == 0)[0])
    assert front.sum() > 50 and np.allclose(bz[front], fr.zbuf[0, ..., 0].numpy()[front], rtol=1e-4)


def test_batches_number_faces_across_meshes(p3d):
    R, S = p3d
    H, W = 32, 48
    a, b = ms.soup(200, 1, size=0.2), ms.soup(150, 2, size=0.2)
    va, vb = torch.tensor(a.reshape(-1, 3)), torch.tensor(b.reshape(-1, 3))
    fa, fb = torch.arange(600).reshape(200, 3), torch.arange(450).reshape(150, 3)
    from pytorch3d.renderer.mesh import rasterize_meshes
    with oracle_mesh_rasterizer():
        p2f, z, bary, d = rasterize_meshes(S.Meshes(verts=[va, vb], faces=[fa, fb]), image_size=(H, W), faces_per_pixel=2,
                                           perspective_correct=True)
    ra = mo.rasterize_meshes_naive(a, (H, W), 0.0, 2, True)
    rb = mo.rasterize_meshes_naive(b, (H, W), 0.0, 2, True)
    assert p2f.shape == (2, H, W, 2)
    assert np.array_equal(p2f[0].numpy(), ra[0])
    assert np.array_equal(p2f[1].numpy(), np.where(rb[0] >= 0, rb[0] + 200, -1))  # packed-face numbering
    assert np.array_equal(z[1].numpy(), rb[1]) and np.array_equal(bary[0].numpy(), ra[2]) and np.array_equal(d[1].numpy(), rb[3])


def test_host_code_has_no_cpu_path_and_rejects_what_the_kernel_does_not_do():
    from sugar_amd.mesh_raster import rasterize_face_verts
    t = torch.tensor(ms.soup(10, 1))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        rasterize_face_verts(t, [0], [10], (16, 16), 0.0, 1, True)
    with pytest.raises(NotImplementedError):
        rasterize_face_verts(t, [0], [10], (16, 16), 1e-3, 1, True)
    with pytest.raises(NotImplementedError):
        rasterize_face_verts(t, [0], [10], (16, 16), 0.0, 1, True, True)
    with pytest.raises(ValueError):
        rasterize_face_verts(t, [0], [10], (16, 16), 0.0, 0, True)
    with pytest.raises(ValueError):
        rasterize_face_verts(t.reshape(-1, 3), [0], [10], (16, 16), 0.0, 1, True)
