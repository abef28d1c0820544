"""GPU parity of the HIP triangle-mesh z-buffer (sgr_rasterize_meshes, through the C ABI) against the CPU oracle
(oracle/mesh_rasterizer.c: the restated pytorch3d naive rasterizer): pix_to_face, zbuf, barycentric coordinates and distances
BIT FOR BIT -- both sides evaluate the same individually rounded float operations.  At BASELINE size (2M faces @ 1080p, the
splat mesh of 1M Gaussians) the oracle is out of reach: size-independent properties are checked instead."""
import numpy as np
import pytest
import torch

from oracle import mesh_oracle as mo
from tests import mesh_scenes as ms

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _hip(fv, H, W, K, persp=True, cull=False, attrs=True):
    from sugar_amd.mesh_raster import rasterize_face_verts
    t = torch.as_tensor(fv, dtype=torch.float32, device=DEV).reshape(-1, 3, 3)
    F = t.shape[0]
    p2f, z, b, d = rasterize_face_verts(t, [0], [F], (H, W), 0.0, K, persp, False, cull, want_bary=attrs, want_dists=attrs)
    torch.cuda.synchronize()
    return (p2f[0].cpu().numpy(), z[0].cpu().numpy(), None if b is None else b[0].cpu().numpy(), None if d is None else d[0].cpu().numpy())


def _same(a, b, what):
    if a.dtype.kind == "f":
        ok = np.array_equal(a.view(np.uint32), b.view(np.uint32))
    else:
        ok = np.array_equal(a, b)
    if not ok:
        bad = np.argwhere(a != b)
        raise AssertionError(f"{what}: {len(bad)} of {a.size} elements differ, first at {bad[0]}: hip {a[tuple(bad[0])]} oracle {b[tuple(bad[0])]}")


@pytest.mark.parametrize("H,W,K,F,seed", [(96, 128, 1, 3000, 1), (128, 96, 4, 3000, 2), (200, 152, 10, 8000, 3), (75, 211, 16, 5000, 4),
                                          (64, 64, 3, 500, 5), (300, 300, 10, 40000, 6)])
def test_bit_identical_to_the_oracle_on_random_soups(hip_lib, H, W, K, F, seed):
    fv = ms.soup(F, seed)
    got = _hip(fv, H, W, K)
    ref = mo.rasterize_meshes_naive(fv, (H, W), 0.0, K, True)
    assert (ref[0][..., 0] >= 0).mean() > 0.2
    for a, b, n in zip(got, ref, ("pix_to_face", "zbuf", "bary_coords", "dists")):
        _same(a, b, n)


@pytest.mark.parametrize("persp,cull", [(True, False), (False, False), (True, True)])
def test_bit_identical_on_large_degenerate_and_sliver_faces(hip_lib, persp, cull):
    H, W, K = 176, 240, 10
    fv = ms.mixed(6000, 11)
    got = _hip(fv, H, W, K, persp, cull)
    ref = mo.rasterize_meshes_naive(fv, (H, W), 0.0, K, persp, False, cull)
    if not cull:
        assert (ref[0][..., K - 1] >= 0).mean() > 0.5  # the screen-filling faces give every pixel a deep stack
    for a, b, n in zip(got, ref, ("pix_to_face", "zbuf", "bary_coords", "dists")):
        _same(a, b, n)


def test_many_screen_filling_faces_take_the_single_level_binning(hip_lib):
    """more (face, super-tile) pairs than the level-1 list holds -> the single-level fallback; same result"""
    H, W, K = 480, 640, 4
    rng = np.random.default_rng(3)
    F = 3000
    fv = np.empty((F, 3, 3), np.float32)
    fv[..., :2] = rng.uniform(-2.5, 2.5, (F, 3, 2))
    fv[..., 2] = rng.uniform(1, 5, (F, 3))
    got = _hip(fv, H, W, K, attrs=False)
    ref = mo.rasterize_meshes_naive(fv, (H, W), 0.0, K, True)
    _same(got[0], ref[0], "pix_to_face")
    _same(got[1], ref[1], "zbuf")


def test_empty_and_invisible_meshes(hip_lib):
    H, W, K = 40, 56, 3
    p2f, z, b, d = _hip(np.zeros((0, 3, 3), np.float32), H, W, K)
    assert (p2f == -1).all() and (z == -1).all() and (b == -1).all() and (d == -1).all()
    fv = ms.soup(300, 9)
    fv[..., 2] = -1.0  # everything behind the camera
    p2f, z, b, d = _hip(fv, H, W, K)
    assert (p2f == -1).all() and (z == -1).all() and (b == -1).all() and (d == -1).all()
    fv = ms.soup(300, 9)
    fv[..., 0] += 50.0  # everything off screen
    p2f, z, _, _ = _hip(fv, H, W, K)
    assert (p2f == -1).all() and (z == -1).all()


def test_unsupported_modes_fail_loudly(hip_lib):
    from sugar_amd.mesh_raster import rasterize_face_verts
    t = torch.as_tensor(ms.soup(10, 1), device=DEV)
    with pytest.raises(NotImplementedError):
        rasterize_face_verts(t, [0], [10], (32, 32), 1e-4, 1, True)
    with pytest.raises(NotImplementedError):
        rasterize_face_verts(t, [0], [10], (32, 32), 0.0, 1, True, True)
    with pytest.raises(ValueError):
        rasterize_face_verts(t, [0], [10], (32, 32), 0.0, 17, True)
    with pytest.raises(RuntimeError):
        rasterize_face_verts(t.cpu(), [0], [10], (32, 32), 0.0, 1, True)


def test_splat_mesh_of_a_million_gaussians_at_1080p_properties(hip_lib):
    """BASELINE size: 2M faces (two per Gaussian) at 1920 x 1080, faces_per_pixel 10 as SuGaR asks.  Properties that do not need
    the oracle: (i) the K depths of a pixel ascend and filled slots come first; (ii) every named face strictly contains the
    pixel centre (float64 edge functions, up to rounding of the edges) and its depth is the perspective-correct plane depth;
    (iii) for sampled pixels the named faces are exactly the K nearest of ALL faces covering the pixel (brute force over the 2M
    faces on the device in float64); (iv) K = 1 returns the first column of K = 10."""
    H, W, K, P = 1080, 1920, 10, 1_000_000
    fv = ms.splat_like(P, 21, W, H)
    t = torch.as_tensor(fv, device=DEV)
    from sugar_amd.mesh_raster import rasterize_face_verts
    p2f, z, bary, dists = rasterize_face_verts(t, [0], [2 * P], (H, W), 0.0, K, True)
    p2f, z, bary = p2f[0], z[0], bary[0]
    have = p2f >= 0
    assert float(have[..., 0].float().mean()) > 0.5
    # (i)
    assert bool((have[..., 1:] <= have[..., :-1]).all())
    zz = torch.where(have, z, torch.full_like(z, float("inf")))
    assert bool((zz[..., 1:] >= zz[..., :-1]).all())
    assert bool((z[~have] == -1).all())
    # (ii) on a sample of filled slots
    g = torch.Generator(device="cpu").manual_seed(0)
    idx = have.nonzero()
    sel = idx[torch.randint(0, idx.shape[0], (400_000,), generator=g).to(DEV)]
    r, c, k = sel[:, 0], sel[:, 1], sel[:, 2]
    f = p2f[r, c, k]
    v = t[f].double()
    px = torch.tensor([mo.pix_to_ndc(i, W, H) for i in range(W)], dtype=torch.float64, device=DEV)[W - 1 - c]
    py = torch.tensor([mo.pix_to_ndc(i, H, W) for i in range(H)], dtype=torch.float64, device=DEV)[H - 1 - r]

    def edge(ax, ay, bx, by):
        return (px - ax) * (by - ay) - (py - ay) * (bx - ax)
    area = (v[:, 2, 0] - v[:, 0, 0]) * (v[:, 1, 1] - v[:, 0, 1]) - (v[:, 2, 1] - v[:, 0, 1]) * (v[:, 1, 0] - v[:, 0, 0])
    w0 = edge(v[:, 1, 0], v[:, 1, 1], v[:, 2, 0], v[:, 2, 1]) / area
    w1 = edge(v[:, 2, 0], v[:, 2, 1], v[:, 0, 0], v[:, 0, 1]) / area
    w2 = edge(v[:, 0, 0], v[:, 0, 1], v[:, 1, 0], v[:, 1, 1]) / area
    assert float(torch.stack([w0, w1, w2]).min()) > -1e-4
    inv_z = (w0 / v[:, 0, 2] + w1 / v[:, 1, 2] + w2 / v[:, 2, 2]) / (w0 + w1 + w2)
    assert float(((1.0 / inv_z - z[r, c, k].double()).abs() / z[r, c, k].double()).max()) < 1e-4
    bsum = bary[r, c, k].double().sum(-1)
    assert float((bsum - 1).abs().max()) < 1e-5
    # (iii) brute force for a few pixels
    tv = t.double()
    a0 = (tv[:, 2, 0] - tv[:, 0, 0]) * (tv[:, 1, 1] - tv[:, 0, 1]) - (tv[:, 2, 1] - tv[:, 0, 1]) * (tv[:, 1, 0] - tv[:, 0, 0])
    rows = torch.randint(0, H, (24,), generator=g).tolist(); cols = torch.randint(0, W, (24,), generator=g).tolist()
    checked = 0
    for rr, cc in zip(rows, cols):
        x = mo.pix_to_ndc(W - 1 - cc, W, H); y = mo.pix_to_ndc(H - 1 - rr, H, W)
        e0 = ((x - tv[:, 1, 0]) * (tv[:, 2, 1] - tv[:, 1, 1]) - (y - tv[:, 1, 1]) * (tv[:, 2, 0] - tv[:, 1, 0])) / a0
        e1 = ((x - tv[:, 2, 0]) * (tv[:, 0, 1] - tv[:, 2, 1]) - (y - tv[:, 2, 1]) * (tv[:, 0, 0] - tv[:, 2, 0])) / a0
        e2 = ((x - tv[:, 0, 0]) * (tv[:, 1, 1] - tv[:, 0, 1]) - (y - tv[:, 0, 1]) * (tv[:, 1, 0] - tv[:, 0, 0])) / a0
        margin = torch.minimum(torch.minimum(e0, e1), e2)
        cover = margin > 0
        if bool(((margin.abs() < 1e-5) & (a0.abs() > 1e-12)).any()):
            continue  # a face edge within rounding of the pixel centre: float32 and float64 may disagree
        fi = cover.nonzero()[:, 0]
        zf = 1.0 / ((e0[fi] / tv[fi, 0, 2] + e1[fi] / tv[fi, 1, 2] + e2[fi] / tv[fi, 2, 2]) / (e0[fi] + e1[fi] + e2[fi]))
        order = torch.argsort(zf)[:K]
        want = fi[order]
        got = p2f[rr, cc][: len(want)]
        assert torch.equal(got, want), (rr, cc, got, want)
        assert bool((p2f[rr, cc][len(want):] == -1).all())
        checked += 1
    assert checked >= 12
    # (iv)
    p1, z1, _, _ = rasterize_face_verts(t, [0], [2 * P], (H, W), 0.0, 1, True, want_bary=False, want_dists=False)
    assert torch.equal(p1[0, ..., 0], p2f[..., 0]) and torch.equal(z1[0, ..., 0], z[..., 0])
