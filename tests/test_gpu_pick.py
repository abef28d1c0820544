"""The sampling pass without host round trips (csrc/pick.hip, sugar_amd.sampler.sample_level_sets(sync_free=True)):
device-side pixel subset (sugar_model.py:1929-1957) and per-level row compaction."""
import ctypes as C

import numpy as np
import pytest
import torch

from sugar_amd import _lib, synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pick(depth, k, seed):
    lib = _lib.load()
    n = depth.numel()
    picked = torch.full((k,), -7, dtype=torch.int64, device=DEV)
    words = torch.zeros(2, dtype=torch.int32, device=DEV)
    scratch = torch.empty(int(lib.sgr_pick_pixels_scratch_bytes(n)), dtype=torch.uint8, device=DEV)
    rc = lib.sgr_pick_pixels(n, C.c_void_p(depth.data_ptr()), k, C.c_uint32(seed), C.c_void_p(picked.data_ptr()), C.c_void_p(words.data_ptr()),
                             C.c_void_p(words.data_ptr() + 4), C.c_void_p(scratch.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    return picked.cpu().numpy(), int(words[0]), int(words[1])


@pytest.mark.parametrize("n,frac_valid,k", [(1920 * 1080, 0.35, 124_000), (640 * 400, 0.9, 5000), (100_003, 0.5, 1), (3000, 0.2, 50)])
def test_pick_is_a_k_subset_of_the_valid_pixels_in_raster_order(n, frac_valid, k):
    g = torch.Generator().manual_seed(n)
    depth = torch.rand(n, generator=g) * 5.0
    depth[torch.rand(n, generator=g) > frac_valid] = -1.0
    valid = (~(depth < 0)).nonzero()[:, 0].numpy()
    d = depth.to(DEV)
    a, ca, nv = _pick(d, k, 11)
    assert nv == valid.size and ca == min(k, valid.size)
    sel = a[:ca]
    assert np.all(np.diff(sel) > 0)                       # distinct, raster order
    assert np.isin(sel, valid).all()                      # only pixels with a depth
    assert np.all(a[ca:] == (sel[0] if ca else 0))        # padding: the first picked pixel
    b, cb, _ = _pick(d, k, 11)
    assert np.array_equal(a, b)                           # a function of (seed, validity)
    c, _, _ = _pick(d, k, 12)
    assert not np.array_equal(a, c) or k >= valid.size


def test_pick_takes_every_valid_pixel_when_fewer_than_k_and_copes_with_none():
    n, k = 50_000, 4096
    depth = torch.full((n,), -1.0)
    depth[torch.arange(0, n, 37)] = 2.0
    valid = (~(depth < 0)).nonzero()[:, 0].numpy()
    a, c, nv = _pick(depth.to(DEV), k, 3)
    assert c == valid.size == nv and np.array_equal(a[:c], valid) and np.all(a[c:] == valid[0])
    a, c, nv = _pick(torch.full((n,), -1.0, device=DEV), k, 3)
    assert c == 0 and nv == 0 and np.all(a == 0)


def test_pick_is_uniform_over_the_valid_pixels():
    n, k = 20_000, 2000
    depth = torch.rand(n, generator=torch.Generator().manual_seed(1))
    depth[::3] = -1.0
    d = depth.to(DEV)
    hits = np.zeros(n)
    trials = 200
    for s in range(trials):
        a, c, _ = _pick(d, k, 1000 + s)
        hits[a[:c]] += 1
    valid = (~(depth < 0)).numpy()
    p = k / valid.sum()
    assert hits[~valid].sum() == 0
    z = (hits[valid] - trials * p) / np.sqrt(trials * p * (1 - p))   # per-pixel inclusion counts ~ Binomial(trials, p)
    assert abs(z.mean()) < 0.05 and 0.9 < z.std() < 1.1 and np.abs(z).max() < 6.0


def test_the_sync_free_sampling_pass_equals_the_gathering_one_on_the_same_pixels():
    """sample_level_sets(sync_free=True): fixed-size stages and device-side compaction -- against the same stages with the
    host-side gathers (field.level_set_points), on the pixels the device-side subset chose"""
    from sugar_amd import sampler
    from sugar_amd.field import level_set_points, scaled_rotation
    from sugar_amd.knn import knn_points
    dev = torch.device(DEV)
    sc = syn.make_scene(60_000, 9, 0.01, 0.05)   # (volumetric Gaussians: plenty of level crossings along the rays)
    cams = syn.orbit_cameras(640, 360)
    cam = cams[2]._replace(viewmatrix=cams[2].viewmatrix.to(dev), projmatrix=cams[2].projmatrix.to(dev), campos=cams[2].campos.to(dev))
    m, s_, q, o = (t.to(dev) for t in (sc.means3D, sc.scales, sc.rotations, sc.opacities))
    depth = sampler.render_depth(m, s_, q, o, cam)
    n = 20_000
    res = sampler.sample_level_sets(m, s_, q, o, cam, n_surface_points=n, depth=depth, sync_free=True, seed=5)
    first = next(iter(res.values()))
    picked, n_picked = first["picked"], int(first["n_picked"])
    assert 0 < n_picked <= n and int(first["n_valid_pixels"]) == int((~(depth.reshape(-1) < 0)).sum())
    trimmed = sampler.trim(res)
    # the gathering path on exactly those pixels
    pk = picked[:n_picked]
    world = sampler.unproject_pixels(pk, depth.reshape(-1), cam)
    nbr = knn_points(world[None], m[None], K=16).idx[0]
    ref = level_set_points(world, nbr, cam.campos.reshape(1, 3), m, scaled_rotation(q, s_, inverse_scales=True), o.reshape(-1, 1),
                           sampler.view_std(m, q, s_, cam.campos))
    total = 0
    for lv, r in ref.items():
        t = trimmed[lv]
        rows = r["valid_idx"]
        assert t["intersection_points"].shape[0] == rows.shape[0]
        assert torch.equal(t["intersection_points"], r["intersection_points"]) and torch.equal(t["normals"], r["normals"])
        assert torch.equal(t["pixel_idx"], pk[rows]) and torch.equal(t["gaussian_idx"], nbr[rows, 0])
        total += rows.shape[0]
    assert total > 50
