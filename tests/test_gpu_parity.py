"""GPU parity tests (run with `pytest -m gpu` on an MI355X): the HIP path, called through the reference-shaped Python
API and the C ABI, against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star):
  * tile assignment pixel-exact: radii, projected means / conics / depth keys, per-tile list lengths and the
    depth-sorted per-tile Gaussian lists are compared BIT-EXACTLY;
  * rendered RGB and gradients within 1e-4 relative.  (Since round 6 the blend evaluates alpha exactly as the reference does;
    against THIS oracle -- gcc's expf on the host -- the last bit of exp may still differ.)  The backward sums in a different order (the reference's own atomics are order-nondeterministic), so
    these are checked norm-wise (<= 1e-5 image, <= 1e-4 gradients) and element-wise with a floor of 1e-3 * max|ref|
    (>= 99.9 % of the elements within 1e-4, image and every gradient tensor alike).
"""
import numpy as np
import pytest
import torch

from oracle import cpu_oracle as orc
from sugar_amd import synthetic as syn
from tests import parity_utils as pu

pytestmark = pytest.mark.gpu

GRAD_NAMES = dict(means3D="dL_dmeans3D", means2D="dL_dmeans2D", opacities="dL_dopacity", shs="dL_dsh",
                  colors_precomp="dL_dcolors", scales="dL_dscales", rotations="dL_drotations", cov3D_precomp="dL_dcov3D")


def _check(scene, cam, bg, grads=True, grad_frac=1e-3, **opts):
    st = pu.run_oracle(scene, cam, bg, **opts)
    H, W = cam.image_height, cam.image_width
    g = np.random.default_rng(0).standard_normal((3, H, W)).astype(np.float32)
    hp = pu.run_hip(scene, cam, bg, grad_out=g if grads else None, **opts)
    # ---- pixel-exact part
    assert hp["num_rendered"] == st["num_rendered"]
    assert np.array_equal(hp["radii"], st["radii"])
    vis = st["radii"] > 0
    rec = hp["rec"]
    assert np.array_equal(rec[vis, 0:2].view(np.uint32), st["means2D"][vis].view(np.uint32))
    assert np.array_equal(rec[vis][:, [2, 3, 4]].view(np.uint32), st["conic_opacity"][vis][:, :3].view(np.uint32))
    assert np.array_equal(rec[vis, pu.REC_DEPTH].view(np.uint32), st["depths"][vis].view(np.uint32))
    assert np.array_equal(rec[:, pu.REC_RADIUS].view(np.int32), st["radii"])
    assert np.array_equal(np.diff(hp["tile_start"]), st["ranges"][:, 1] - st["ranges"][:, 0])
    assert np.array_equal(hp["point_list"], st["point_list"])
    if "shs" in pu.scene_kwargs(scene, cam, bg, **opts):
        np.testing.assert_allclose(rec[vis, pu.REC_RGB], st["rgb"][vis], rtol=1e-6, atol=1e-7)
        cl = rec[:, pu.REC_CLAMPED].view(np.uint32)
        assert np.array_equal((cl[vis, None] >> np.arange(3)) & 1, st["clamped"][vis])
    # ---- tolerance part
    assert (hp["n_contrib"] != st["n_contrib"]).mean() <= 1e-4
    e = pu.rel_stats(hp["color"], st["color"])
    assert e["norm_rel"] <= 1e-5 and e["frac_gt_1e4"] <= 1e-3 and e["max_abs"] <= 5e-3, e
    e = pu.rel_stats(hp["final_T"], st["final_T"])
    assert e["norm_rel"] <= 1e-5 and e["frac_gt_1e4"] <= 1e-3, e
    # tile_maxc is the per-tile maximum of n_contrib
    gx, gy = st["grid"]
    nc = np.zeros((gy * 16, gx * 16), np.uint32); nc[:H, :W] = hp["n_contrib"].reshape(H, W)
    assert np.array_equal(hp["tile_maxc"], nc.reshape(gy, 16, gx, 16).max(axis=(1, 3)).reshape(-1))
    if grads:
        gr = orc.backward(st, g)
        for k, v in hp["grads"].items():
            ref = gr[GRAD_NAMES[k]]
            e = pu.rel_stats(v.reshape(ref.shape), ref)
            # the north star's bar: 1e-4 relative on every gradient tensor -- norm-wise, and >= 99.9 % of the elements
            # within 1e-4 (relative, floor 1e-3 * max|ref|)
            # (small scenes: ONE flipped alpha < 1/255 or T < 1e-4 decision moves the three or four gradient elements of a
            # Gaussian that covers a handful of pixels, and a few thousand elements cannot average that away: the count of
            # elements beyond 1e-4 may reach 30 before the fraction counts; the full-size cases of tests/test_gpu_fullsize.py
            # sit at 1e-5 .. 6e-5 against the 1e-3 allowed)
            assert e["norm_rel"] <= 1e-4 and e["frac_gt_1e4"] <= max(grad_frac, 30.0 / v.size), (k, e)
    return st, hp


def test_config1_sh_degree3():
    scene, cams, bg = syn.make_config("config1")
    _check(scene, cams[0], bg)


def test_config1_all_cameras_forward():
    scene, cams, bg = syn.make_config("config1")
    for cam in cams[1:]:
        _check(scene, cam, bg, grads=False)


def test_precomputed_colours_white_background():
    scene, cams, _ = syn.make_config("config1")
    _check(scene, cams[3], torch.ones(3), use_sh=False)


def test_precomputed_covariance_low_sh_degree_scale_modifier():
    scene, cams, _ = syn.make_config("config1")
    _check(scene, cams[5], torch.tensor([0.2, 0.5, 0.7]), use_cov=True, sh_degree=1, scale_modifier=1.3)


@pytest.mark.parametrize("deg", [0, 2])
def test_sh_degrees(deg):
    scene = syn.make_scene(3000, 31, 0.01, 0.1)
    _check(scene, syn.orbit_cameras(128, 96)[deg], torch.zeros(3), sh_degree=deg)


def test_partial_tiles_and_large_gaussians():
    scene = syn.make_scene(3000, 11, 0.01, 0.3)
    # splats hundreds of pixels wide: thousands of per-pixel terms of either sign cancel in every gradient element, and ANY
    # float summation order (the reference's atomics included) scatters a fraction of a percent of the elements by more than
    # 1e-4 around the double-summed oracle -- tests/test_gpu_reference.py::test_large_splat_gradient_spread measures the
    # reference's own scatter on this scene and holds the product to it; here the element-wise bar is 99.5 %
    scene_bar = 5e-3
    _check(scene, syn.orbit_cameras(250, 190)[2], torch.tensor([0.1, 0.2, 0.3]), grad_frac=scene_bar)


def test_camera_inside_cloud_near_culling():
    scene = syn.make_scene(2000, 12, 0.05, 0.5)
    _check(scene, syn.look_at_camera((0.1, 0.0, 0.0), (1.0, 0.2, 0.0), 200, 120), torch.zeros(3))


def test_long_tile_lists_use_large_sort_paths():
    """> 2048 and > 16384 instances in one tile: the 128 KB LDS sort and the global-memory fallback."""
    g = torch.Generator().manual_seed(3)
    P = 40000
    base = syn.make_scene(P, 13, 0.004, 0.01)
    m = torch.randn(P, 3, generator=g) * 0.002  # two tight clusters: one tile with 30000, one with 10000 instances
    m[: P // 4] += torch.tensor([0.0, 0.6, 0.3])
    scene = base._replace(means3D=m.contiguous(), opacities=torch.full((P, 1), 0.02))
    # gradients too: 30 000-entry walks put the backward's T * rcp(1 - alpha) reconstruction (blend.hip, backward.cu:503)
    # through its longest chains
    st, hp = _check(scene, syn.orbit_cameras(160, 128)[0], torch.zeros(3), grads=True)
    lens = st["ranges"][:, 1] - st["ranges"][:, 0]
    assert lens.max() > 16384 and ((lens > 2048) & (lens <= 16384)).any()


def test_everything_culled():
    scene = syn.make_scene(100, 4, 0.01, 0.05)
    cam = syn.look_at_camera((0.0, -3.0, 0.0), (0.0, -6.0, 0.0), 64, 64)
    st, hp = _check(scene, cam, torch.tensor([0.2, 0.4, 0.6]))
    assert hp["num_rendered"] == 0
    assert all(np.all(v == 0) for v in hp["grads"].values() if v is not None)


def test_mark_visible_matches_oracle():
    from sugar_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    scene = syn.make_scene(5000, 12, 0.05, 0.5)
    cam = syn.look_at_camera((0.1, 0.0, 0.0), (1.0, 0.2, 0.0), 200, 120)
    dev = torch.device("cuda:0")
    r = GaussianRasterizer(GaussianRasterizationSettings(120, 200, cam.tanfovx, cam.tanfovy, torch.zeros(3, device=dev), 1.0,
                                                         cam.viewmatrix.to(dev), cam.projmatrix.to(dev), 3,
                                                         cam.campos.to(dev), False, False))
    vis = r.markVisible(scene.means3D.to(dev)).cpu().numpy()
    ref = orc.mark_visible(scene.means3D.numpy(), cam.viewmatrix.numpy(), cam.projmatrix.numpy())
    assert vis.dtype == np.bool_ and np.array_equal(vis, ref) and 0 < ref.sum() < 5000


def test_forward_is_bit_stable_and_debug_mode_runs():
    scene, cams, bg = syn.make_config("config1")
    a = pu.run_hip(scene, cams[2], bg)
    b = pu.run_hip(scene, cams[2], bg, debug=True)
    assert np.array_equal(a["color"], b["color"]) and np.array_equal(a["point_list"], b["point_list"])


def test_non_default_stream():
    scene, cams, bg = syn.make_config("config1")
    ref = pu.run_hip(scene, cams[4], bg)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        out = pu.run_hip(scene, cams[4], bg)
    s.synchronize()
    assert np.array_equal(ref["color"], out["color"])


def _check_list_properties(hp, W, H):
    """Size-independent properties of the binning: every tile list sorted by (depth bits, index), lists hold exactly the
    Gaussians whose rectangle (getRect, auxiliary.h:46-56, recomputed on the host) covers the tile, R = sum of areas."""
    rec, pl, ts = hp["rec"], hp["point_list"].astype(np.int64), hp["tile_start"].astype(np.int64)
    R = hp["num_rendered"]
    assert ts[-1] == R == len(pl)
    depth_bits = rec[:, pu.REC_DEPTH].view(np.uint32).astype(np.uint64)
    key = (depth_bits[pl] << np.uint64(32)) | pl.astype(np.uint64)
    tile_of = np.repeat(np.arange(len(ts) - 1), np.diff(ts))
    same_tile = tile_of[1:] == tile_of[:-1]
    assert np.all(key[1:][same_tile] > key[:-1][same_tile]), "a tile list is not sorted by (depth, index)"
    # rectangle membership: recompute getRect on the host from the record and compare per-Gaussian tile counts
    radii = hp["radii"].astype(np.float32)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    x, y = rec[:, 0], rec[:, 1]
    f16, f1 = np.float32(16), np.float32(1)  # getRect, auxiliary.h:46-56: (p + r + BLOCK - 1) / BLOCK, evaluated left to right
    minx = np.clip(np.trunc((x - radii) / f16), 0, gx); maxx = np.clip(np.trunc((((x + radii) + f16) - f1) / f16), 0, gx)
    miny = np.clip(np.trunc((y - radii) / f16), 0, gy); maxy = np.clip(np.trunc((((y + radii) + f16) - f1) / f16), 0, gy)
    area = ((maxx - minx) * (maxy - miny)).astype(np.int64) * (hp["radii"] > 0)
    assert np.array_equal(np.bincount(pl, minlength=len(radii)), area)
    txs, tys = tile_of % gx, tile_of // gx
    assert np.all((txs >= minx[pl]) & (txs < maxx[pl]) & (tys >= miny[pl]) & (tys < maxy[pl]))


@pytest.mark.parametrize("config", ["config2", "metric"])
def test_full_size_properties(config):
    """BASELINE config 2 (300k Gaussians, 800x800) and the metric case (1M Gaussians, 1920x1080) at full size:
    size-independent properties instead of the oracle: every tile list is sorted by (depth, index), lists hold exactly the
    Gaussians whose rectangle covers the tile, R = sum of rectangle areas, T in [0,1], colour bounded, backward finite and
    zero for culled Gaussians."""
    scene, cams, bg = syn.make_config(config)
    cam = cams[0]
    H, W = cam.image_height, cam.image_width
    g = np.random.default_rng(0).standard_normal((3, H, W)).astype(np.float32)
    hp = pu.run_hip(scene, cam, bg, grad_out=g)
    _check_list_properties(hp, W, H)
    radii = hp["radii"]
    assert np.all((hp["final_T"] >= 0) & (hp["final_T"] <= 1)) and np.isfinite(hp["color"]).all()
    assert hp["color"].min() >= 0 and hp["color"].max() <= 3.0
    for k, v in hp["grads"].items():
        assert np.isfinite(v).all(), k
        if k != "means2D":
            assert np.all(v.reshape(len(radii), -1)[hp["radii"] == 0] == 0), k


def test_depth_sort_with_more_chunks_than_resident_workgroups():
    """4.6M Gaussians = 1124 sort chunks, more than the ~1000 workgroups the chip holds at once: the look-back of the
    one-kernel radix passes (binning.hip) then waits on chunks of earlier dispatch waves.  Forward only, tiny splats on a
    small image; the tile lists must come out sorted by (depth bits, index) and complete."""
    P = 4_600_000
    scene = syn.make_scene(P, 21, 0.0004, 0.0015)
    cam = syn.orbit_cameras(160, 128)[1]
    hp = pu.run_hip(scene, cam, torch.zeros(3), use_sh=False)
    assert (hp["radii"] > 0).sum() > P // 4
    _check_list_properties(hp, 160, 128)


def test_knn_and_dist2_match_oracle():
    import ctypes as C
    from sugar_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    pts = torch.randn(3000, 3, generator=torch.Generator().manual_seed(4))
    p = pts.to(dev)
    out = torch.empty(3000, device=dev)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.sgr_dist2(3000, C.c_void_p(p.data_ptr()), C.c_void_p(out.data_ptr()), s) == 0
    assert np.array_equal(out.cpu().numpy(), orc.dist2(pts.numpy()))  # same arithmetic, same value bit for bit
    K = 16
    d = torch.empty(3000, K, device=dev); i = torch.empty(3000, K, dtype=torch.int64, device=dev)
    assert lib.sgr_knn(3000, C.c_void_p(p.data_ptr()), 3000, C.c_void_p(p.data_ptr()), K, C.c_void_p(d.data_ptr()),
                       C.c_void_p(i.data_ptr()), s) == 0
    from scipy.spatial import cKDTree
    dd, ii = cKDTree(pts.numpy().astype(np.float64)).query(pts.numpy().astype(np.float64), k=K)
    np.testing.assert_allclose(d.cpu().numpy(), dd ** 2, rtol=1e-4, atol=1e-7)
    assert (i.cpu().numpy() == ii).mean() > 0.999 and np.array_equal(i[:, 0].cpu().numpy(), np.arange(3000))


def test_large_tile_grid_4k():
    """BASELINE config-5 image size (3840x2160 = 32 400 tiles, 127 KB of LDS tile counters per workgroup)."""
    scene = syn.make_scene(3000, 33, 0.02, 0.2)
    _check(scene, syn.orbit_cameras(3840, 2160)[1], torch.zeros(3), grads=False)


def test_8k_image_runs_on_the_two_level_binning_only():
    """7680x4320 = 129 600 tiles: beyond the single-level path's one-LDS-counter-per-tile limit; the default path has none"""
    from sugar_amd import _lib
    scene = syn.make_scene(20000, 35, 0.002, 0.01)
    cam = syn.orbit_cameras(7680, 4320)[2]
    from sugar_amd.diff_gaussian_rasterization import _C, grad_sink
    _check(scene, cam, torch.zeros(3), grads=False)
    assert _C.last_forward["binning_mode"] == 0
    with grad_sink(single_level_binning=True):
        with pytest.raises(RuntimeError, match="too large for the single-level"):
            pu.run_hip(scene, cam, torch.zeros(3))


def test_knn_with_an_uninstantiated_k_is_a_prefix_of_the_next_one():
    from sugar_amd.knn import knn_points
    dev = torch.device("cuda:0")
    pts = syn.make_scene(6000, 36, 0.01, 0.02).means3D.to(dev)
    a = knn_points(pts[None], pts[None], K=5, return_nn=True)
    b = knn_points(pts[None], pts[None], K=8)
    assert a.dists.shape == (1, 6000, 5) and a.knn.shape == (1, 6000, 5, 3)
    assert torch.equal(a.idx, b.idx[..., :5]) and torch.equal(a.dists, b.dists[..., :5])
    with pytest.raises(RuntimeError):
        knn_points(pts[None], pts[None], K=33)


def test_tiny_image():
    scene = syn.make_scene(2000, 34, 0.02, 0.2)
    _check(scene, syn.orbit_cameras(100, 20)[0], torch.tensor([0.5, 0.1, 0.9]))


@pytest.mark.parametrize("dist", ["uniform", "clustered"])
def test_grid_knn_is_identical_to_exhaustive(dist):
    """sgr_knn_grid / sgr_dist2_grid return the same values AND indices as the exhaustive kernels (ties on index)."""
    from sugar_amd.knn import knn_points, distCUDA2
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    P = 30000
    if dist == "uniform":
        pts = torch.rand(P, 3, generator=g) * 2 - 1
    else:  # dense clusters, duplicates and far outliers stretching the bounding box
        c = torch.randn(20, 3, generator=g)
        pts = c[torch.randint(0, 20, (P,), generator=g)] + 0.01 * torch.randn(P, 3, generator=g)
        pts[:50] = pts[50:100]
        pts[100] = torch.tensor([40.0, -35.0, 22.0])
    p = pts.to(dev)
    for K in (1, 4, 16):
        a = knn_points(p[None], p[None], K=K, method="brute")
        b = knn_points(p[None], p[None], K=K, method="grid")
        assert torch.equal(a.dists, b.dists) and torch.equal(a.idx, b.idx)
    q = (torch.rand(5000, 3, generator=g) * 3 - 1.5).to(dev)  # queries outside the reference bounding box too
    a = knn_points(q[None], p[None], K=8, method="brute")
    b = knn_points(q[None], p[None], K=8, method="grid")
    assert torch.equal(a.dists, b.dists) and torch.equal(a.idx, b.idx)
    # (query sets up to 262144 go to the wave-per-query kernel directly; a larger one takes the lane-per-query ring walk first)
    q = (torch.rand(300_000, 3, generator=g) * 3 - 1.5).to(dev)
    a = knn_points(q[None], p[None], K=8, method="brute")
    b = knn_points(q[None], p[None], K=8, method="grid")
    assert torch.equal(a.dists, b.dists) and torch.equal(a.idx, b.idx)
    assert torch.equal(distCUDA2(p, method="brute"), distCUDA2(p, method="grid"))
    # queries far outside the cloud (the level-set sampler's silhouette pixels), along faces, edges and corners of the bounding
    # box, on its boundary planes and a hair outside them: the ring walk's stop bound uses the distance to the box there
    lo, hi = pts.min(0).values, pts.max(0).values
    ext = float((hi - lo).max())
    far = []
    for scale in (1e-6, 0.01, 0.3, 1.0, 5.0):
        d = torch.randn(600, 3, generator=g)
        d = d / d.norm(dim=1, keepdim=True)
        base = lo + (hi - lo) * torch.rand(600, 3, generator=g)
        out = base + d * ext * scale
        axis = torch.randint(0, 3, (600,), generator=g)
        side = torch.randint(0, 2, (600,), generator=g).bool()
        snap = torch.where(side, hi[axis], lo[axis]) + torch.where(side, 1.0, -1.0) * ext * scale  # straight out of one face
        out[torch.arange(600), axis] = snap
        far.append(out)
    far.append(torch.stack([lo, hi, torch.tensor([float(lo[0]), float(hi[1]), float(lo[2])]), (lo + hi) / 2]))
    qf = torch.cat(far).to(dev)
    for K in (1, 16):
        a = knn_points(qf[None], p[None], K=K, method="brute")
        b = knn_points(qf[None], p[None], K=K, method="grid")
        assert torch.equal(a.dists, b.dists) and torch.equal(a.idx, b.idx)


def _lists(scene, cam, bg, single_level=False):
    from sugar_amd.diff_gaussian_rasterization import _C, grad_sink
    with grad_sink(single_level_binning=single_level):
        h = pu.run_hip(scene, cam, bg)
    return h, _C.last_forward["binning_mode"]


def test_two_level_and_single_level_binning_give_identical_lists():
    """binning2.hip (super-tiles, then tiles) against binning.hip (single-level ordered scatter): same ranges, same lists,
    same image -- at a size where every level-2 code path runs (several chunks per super-tile, partial border super-tiles)."""
    scene = syn.make_scene(120000, 21, 0.004, 0.05)
    cam = syn.orbit_cameras(1000, 600)[3]   # 63 x 38 tiles: the last super-tile column / row are partial
    bg = torch.tensor([0.1, 0.2, 0.3])
    a, mode_a = _lists(scene, cam, bg)
    b, mode_b = _lists(scene, cam, bg, single_level=True)  # per call (SGR_FLAG_SINGLE_LEVEL_BINNING): no process-wide switch
    assert (mode_a, mode_b) == (0, 1)
    assert a["num_rendered"] == b["num_rendered"] and a["num_rendered"] > 1_000_000
    assert np.array_equal(a["tile_start"], b["tile_start"])
    assert np.array_equal(a["point_list"], b["point_list"])
    assert np.array_equal(a["color"], b["color"]) and np.array_equal(a["n_contrib"], b["n_contrib"])


def test_level1_overflow_falls_back_to_single_level_binning():
    """a few huge splats on a 4K grid touch more (Gaussian, super-tile) pairs than the level-1 list holds: the forward must
    notice, take the single-level path and still match the oracle bit for bit"""
    scene = syn.make_scene(3000, 22, 0.4, 0.9)
    cam = syn.orbit_cameras(3840, 2160)[0]
    bg = torch.zeros(3)
    h, mode = _lists(scene, cam, bg)
    assert mode == 1
    o = pu.run_oracle(scene, cam, bg)
    assert np.array_equal(h["radii"], o["radii"])
    assert h["num_rendered"] == o["num_rendered"]
    assert np.array_equal(np.diff(h["tile_start"]), o["ranges"][:, 1] - o["ranges"][:, 0])
    assert np.array_equal(h["point_list"], o["point_list"])


@pytest.mark.parametrize("P", [1, 2, 63, 64, 65, 511, 513, 1023, 1025, 2049])
def test_counts_around_the_chunk_boundaries(P):
    """Gaussian counts around the 64-lane, 512-entry and 1024-key chunk sizes of the sort and binning kernels"""
    scene = syn.make_scene(P, 100 + P, 0.02, 0.25)
    _check(scene, syn.orbit_cameras(176, 144)[P % 8], torch.tensor([0.3, 0.2, 0.1]))


def test_empty_input_short_circuits_like_the_reference():
    """P == 0 never reaches the core (rasterize_points.cu:68-69,81): the image stays at its torch::full(0.0) initial value
    (NOT the background), radii are empty"""
    from sugar_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device("cuda:0")
    cam = syn.orbit_cameras(64, 48)[0]
    bg = torch.tensor([0.25, 0.5, 0.75], device=dev)
    st = GaussianRasterizationSettings(48, 64, cam.tanfovx, cam.tanfovy, bg, 1.0, cam.viewmatrix.to(dev), cam.projmatrix.to(dev), 3,
                                       cam.campos.to(dev), False, False)
    e = lambda *s: torch.zeros(*s, device=dev, requires_grad=True)
    color, radii = GaussianRasterizer(st)(e(0, 3), e(0, 3), e(0, 1), shs=e(0, 16, 3), scales=e(0, 3), rotations=e(0, 4))
    assert color.shape == (3, 48, 64) and radii.shape == (0,)
    assert torch.equal(color, torch.zeros(3, 48, 64, device=dev))
    color.sum().backward()  # and the backward of the empty call runs


def test_backward_is_linear_in_the_pixel_gradient_at_full_size():
    """size-independent property at the metric workload (1M Gaussians @ 1080p): the backward is linear in dL/dpixel, so
    B(g1 + 2 g2) = B(g1) + 2 B(g2) for every gradient tensor (float-atomic summation order is the only difference)"""
    from sugar_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device("cuda:0")
    scene, cams, bg = syn.make_config("metric")
    cam = cams[3]
    H, W = cam.image_height, cam.image_width
    st = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, bg.to(dev), 1.0, cam.viewmatrix.to(dev), cam.projmatrix.to(dev), 3,
                                       cam.campos.to(dev), False, False)
    leaves = [t.to(dev).requires_grad_(True) for t in (scene.means3D, scene.opacities, scene.shs, scene.scales, scene.rotations)]
    m2 = torch.zeros_like(leaves[0], requires_grad=True)
    color, _ = GaussianRasterizer(st)(leaves[0], m2, leaves[1], shs=leaves[2], scales=leaves[3], rotations=leaves[4])
    gen = torch.Generator().manual_seed(11)
    g1 = torch.randn(3, H, W, generator=gen).to(dev); g2 = torch.randn(3, H, W, generator=gen).to(dev)
    B = lambda g: torch.autograd.grad(color, leaves + [m2], grad_outputs=g, retain_graph=True)
    b1, b2, b12 = B(g1), B(g2), B(g1 + 2.0 * g2)
    for a, b, c in zip(b1, b2, b12):
        ref = a + 2.0 * b
        assert torch.isfinite(c).all()
        assert float((c - ref).norm() / ref.norm()) < 2e-5


def test_grid_knn_on_a_surface_cloud_with_queries_off_the_surface():
    """BASELINE config 4's geometry: the reference set lies on a 2-D surface (most grid cells are empty) and the queries -- pixels
    of the level-set sampler unprojected from a blended depth -- sit anywhere from on the surface to a scene diameter away from it,
    INSIDE the bounding box.  Such queries meet fewer than K points in their first ring: k_knn_ball takes their bound from a spread
    sample of the set.  Identical values and indices to the exhaustive kernel; also for sets smaller than K and than the sample."""
    from sugar_amd.knn import knn_points
    dev = torch.device("cuda:0")
    b = syn.make_bound_scene(200_000, seed=4)
    p = b.scene.means3D.to(dev)
    g = torch.Generator().manual_seed(3)
    n = 6000
    base = b.scene.means3D[torch.randint(0, p.shape[0], (n,), generator=g)]
    off = torch.cat([torch.zeros(1000), 10 ** (torch.rand(n - 1000, generator=g) * 4 - 4) * 2.0])     # 0 ... 2 scene units
    q = (base * (1.0 - off[:, None] / base.norm(dim=1, keepdim=True))).to(dev)                        # towards the centre of the shape
    for K in (1, 16, 32):
        a = knn_points(q[None], p[None], K=K, method="brute")
        c = knn_points(q[None], p[None], K=K, method="grid")
        assert torch.equal(a.dists, c.dists) and torch.equal(a.idx, c.idx), K
    for M in (5, 40, 3000):   # fewer points than K; fewer than the sample; a sample of every point
        small = p[:M].contiguous()
        a = knn_points(q[None], small[None], K=16, method="brute")
        c = knn_points(q[None], small[None], K=16, method="grid")
        assert torch.equal(a.dists, c.dists) and torch.equal(a.idx, c.idx), M


def test_the_samplers_two_per_view_kernels_equal_the_torch_formulas_they_replace():
    """sgr_view_std / sgr_unproject_pixels (csrc/field.hip) against sugar_model.py:1934-1972 written out with torch ops"""
    from sugar_amd import sampler, synthetic as syn
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    P = 20_000
    means = torch.randn(P, 3, generator=g).to(dev)
    q = torch.nn.functional.normalize(torch.randn(P, 4, generator=g), dim=-1).to(dev)
    sc = torch.exp(torch.randn(P, 3, generator=g) - 3).to(dev)
    c = syn.orbit_cameras(640, 360, 8)[2]
    cam = c._replace(viewmatrix=c.viewmatrix.to(dev), projmatrix=c.projmatrix.to(dev), campos=c.campos.to(dev))
    # gaussian_std
    to_cam = torch.nn.functional.normalize(cam.campos.reshape(1, 3) - means, dim=-1)
    r, x, y, z = q.unbind(-1)
    vx, vy, vz = to_cam.unbind(-1)
    ox = (1 - 2 * (y * y + z * z)) * vx + 2 * (x * y + r * z) * vy + 2 * (x * z - r * y) * vz
    oy = 2 * (x * y - r * z) * vx + (1 - 2 * (x * x + z * z)) * vy + 2 * (y * z + r * x) * vz
    oz = 2 * (x * z + r * y) * vx + 2 * (y * z - r * x) * vy + (1 - 2 * (x * x + y * y)) * vz
    want = (sc * torch.stack([ox, oy, oz], dim=-1)).norm(dim=-1)
    got = sampler.view_std(means, q, sc, cam.campos)
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-9)
    # the depth render's colours
    from sugar_amd import _lib
    import ctypes as C
    rgb = torch.empty(P, 3, device=dev)
    rc = _lib.load().sgr_view_depth_rgb(P, C.c_void_p(means.data_ptr()), C.c_void_p(cam.viewmatrix.contiguous().data_ptr()),
                                        C.c_void_p(rgb.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert rc == 0 and torch.allclose(rgb, sampler.view_depth(means, cam.viewmatrix).expand(-1, 3), rtol=1e-5, atol=1e-6)
    # back-projection
    H, W = int(cam.image_height), int(cam.image_width)
    depth = (torch.rand(H * W, generator=g) * 5 + 0.5).to(dev)
    picked = torch.randperm(H * W, generator=g)[:5000].to(dev)
    m = min(W, H)
    rows = torch.div(picked, W, rounding_mode="floor")
    cols = picked - rows * W
    ndc_x = W / m - (cols.to(torch.float32) / (m - 1)) * 2
    ndc_y = H / m - (rows.to(torch.float32) / (m - 1)) * 2
    zz = depth[picked]
    xc = -ndc_x * zz / ((W / (2.0 * cam.tanfovx)) * 2.0 / m)
    yc = -ndc_y * zz / ((H / (2.0 * cam.tanfovy)) * 2.0 / m)
    c2w = torch.linalg.inv(cam.viewmatrix.double())
    want = (torch.stack([xc, yc, zz, torch.ones_like(zz)], dim=-1).double() @ c2w)[:, :3]
    got = sampler.unproject_pixels(picked, depth, cam)
    assert (got.double() - want).abs().max() < 1e-5 * float(want.abs().max())
