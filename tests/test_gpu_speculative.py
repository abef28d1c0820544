"""GPU tests of the speculative forward (SGR_FLAG_SPECULATIVE, include/sugar_raster.h): what every unmodified caller of the
reference-shaped API gets from its second call with the same (device, P, W, H) on.  The reference waits for num_rendered in the
middle of the forward (rasterizer_impl.cu:280-281); the speculative forward enqueues everything with a guessed list capacity and
waits at the END of the call.  Contract: the same num_rendered, the same lists, the same image bit for bit, and gradients equal to
the plain forward's up to the float-atomic order -- whether the guess holds (hit) or not (miss: the tail kernels run twice)."""
import numpy as np
import pytest
import torch

from sugar_amd import synthetic as syn
from tests import parity_utils as pu

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _same(a, b, grads=True):
    assert a["num_rendered"] == b["num_rendered"]
    assert np.array_equal(a["radii"], b["radii"]) and np.array_equal(a["tile_start"], b["tile_start"])
    assert np.array_equal(a["point_list"], b["point_list"])
    assert np.array_equal(a["color"], b["color"]) and np.array_equal(a["n_contrib"], b["n_contrib"])
    if grads:
        for k in a["grads"]:
            if a["grads"][k] is not None:
                assert pu.rel_stats(b["grads"][k], a["grads"][k])["norm_rel"] < 2e-5, k


@pytest.mark.parametrize("use_sh", [True, False])
def test_speculative_hit_and_miss_equal_the_plain_forward(use_sh):
    from sugar_amd import diff_gaussian_rasterization as dgr
    from sugar_amd.diff_gaussian_rasterization import _C, grad_sink
    W, H, P = 640, 400, 80_000 if use_sh else 80_001   # (a (P, W, H) no other test of this process has used)
    scene = syn.make_scene(P, 31, 0.004, 0.05)
    cams = syn.orbit_cameras(W, H)
    bg = torch.tensor([0.2, 0.1, 0.3])
    g = np.random.default_rng(3).standard_normal((3, H, W)).astype(np.float32)
    key = (torch.device(DEV).index if torch.device(DEV).index is not None else 0, P, W, H)
    with grad_sink(speculative=False):
        plain = [pu.run_hip(scene, cams[i], bg, use_sh=use_sh, grad_out=g) for i in range(3)]
        assert not _C.last_forward["speculative"]
    dgr._SPEC_CAP.pop(key, None)
    # first ordinary call: nothing known about this size yet -> host round trip; it leaves the count behind
    a0 = pu.run_hip(scene, cams[0], bg, use_sh=use_sh, grad_out=g)
    assert not _C.last_forward["speculative"] and dgr._SPEC_CAP[key] == a0["num_rendered"]
    _same(plain[0], a0)
    # second call: speculative, the capacity (1.5 x + 64k) holds
    a1 = pu.run_hip(scene, cams[1], bg, use_sh=use_sh, grad_out=g)
    lf = _C.last_forward
    assert lf["speculative"] and not lf["speculation_missed"] and lf["list_capacity"] >= a1["num_rendered"]
    _same(plain[1], a1)
    # a guess that is far too small: the tail runs twice, the result is the same
    dgr._SPEC_CAP[key] = 1000
    a2 = pu.run_hip(scene, cams[2], bg, use_sh=use_sh, grad_out=g)
    lf = _C.last_forward
    assert lf["speculative"] and lf["speculation_missed"]
    _same(plain[2], a2)
    assert dgr._SPEC_CAP[key] == a2["num_rendered"]   # ... and the next guess has learnt from it
    a3 = pu.run_hip(scene, cams[2], bg, use_sh=use_sh, grad_out=g)
    assert _C.last_forward["speculative"] and not _C.last_forward["speculation_missed"]
    _same(plain[2], a3)


def test_speculative_forward_with_a_level1_overflow_falls_back_to_the_single_level_binning():
    """splats hundreds of pixels wide: the level-1 list overflows; a speculative call must notice at its end-of-call check and take
    the single-level path exactly as the plain call does"""
    from sugar_amd import diff_gaussian_rasterization as dgr
    from sugar_amd.diff_gaussian_rasterization import _C, grad_sink
    W, H, P = 1024, 640, 6001
    scene = syn.make_scene(P, 8, 0.3, 0.9)
    cam = syn.orbit_cameras(W, H)[2]
    bg = torch.zeros(3)
    with grad_sink(speculative=False):
        plain = pu.run_hip(scene, cam, bg)
        mode = _C.last_forward["binning_mode"]
    if mode != 1:
        pytest.skip("this scene does not overflow the level-1 list")
    key = (0, P, W, H)
    dgr._SPEC_CAP[key] = plain["num_rendered"]
    spec = pu.run_hip(scene, cam, bg)
    lf = _C.last_forward
    assert lf["speculative"] and lf["speculation_missed"] and lf["binning_mode"] == 1
    _same(plain, spec, grads=False)


def test_a_trainer_shaped_loop_never_takes_the_mid_forward_round_trip_after_its_first_call():
    """forward -> loss -> backward -> (parameters move) through the reference-shaped API, 12 iterations over 4 cameras: from the
    second call on every forward is speculative, and none misses while the scene drifts slowly"""
    from sugar_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _C
    dev = torch.device(DEV)
    W, H, P = 480, 320, 50_003
    scene = syn.make_scene(P, 12, 0.004, 0.04)
    cams = syn.orbit_cameras(W, H)[:4]
    leaves = [getattr(scene, k).to(dev).clone().requires_grad_(True) for k in ("means3D", "opacities", "shs", "scales", "rotations")]
    m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
    bg = torch.zeros(3, device=dev)
    spec, missed = [], []
    for it in range(12):
        c = cams[it % 4]
        st = GaussianRasterizationSettings(H, W, c.tanfovx, c.tanfovy, bg, 1.0, c.viewmatrix.to(dev), c.projmatrix.to(dev), 3, c.campos.to(dev),
                                           False, False)
        img, radii = GaussianRasterizer(st)(leaves[0], m2, leaves[1], shs=leaves[2], scales=leaves[3], rotations=leaves[4])
        spec.append(_C.last_forward["speculative"]); missed.append(_C.last_forward["speculation_missed"])
        (img ** 2).mean().backward()
        with torch.no_grad():
            leaves[0] -= 1e-3 * leaves[0].grad.sign()
            for t in leaves + [m2]:
                t.grad = None
    assert spec == [False] + [True] * 11 and not any(missed)


@pytest.mark.parametrize("use_sh,use_cov", [(True, False), (False, False), (True, True)])
def test_the_torch_extension_path_equals_the_ctypes_path(use_sh, use_cov):
    """The plain reference-shaped call goes through the PyTorch C++ extension (csrc/torch_ext.cpp: DGR/ext.cpp's `_C` over the C
    ABI); with a grad_sink extension in effect, or SGR_TORCH_EXT=0, through the ctypes binding.  Same library underneath: same lists,
    same image bit for bit, gradients equal up to the float-atomic order."""
    from sugar_amd import diff_gaussian_rasterization as dgr
    from sugar_amd.diff_gaussian_rasterization import _C
    ext = dgr._ext()
    assert ext is not None, "sugar_amd/_C_ext.so missing: python -m sugar_amd.build"
    W, H, P = 512, 320, 40_007
    scene = syn.make_scene(P, 5, 0.004, 0.05)
    cam = syn.orbit_cameras(W, H)[6]
    bg = torch.tensor([0.0, 0.5, 1.0])
    g = np.random.default_rng(1).standard_normal((3, H, W)).astype(np.float32)
    a = pu.run_hip(scene, cam, bg, use_sh=use_sh, use_cov=use_cov, grad_out=g)
    assert _C.last_forward["torch_ext"]
    saved = (dgr._EXT, dgr._EXT_TRIED)
    dgr._EXT, dgr._EXT_TRIED = None, True
    try:
        b = pu.run_hip(scene, cam, bg, use_sh=use_sh, use_cov=use_cov, grad_out=g)
        assert not _C.last_forward["torch_ext"]
    finally:
        dgr._EXT, dgr._EXT_TRIED = saved
    _same(b, a)
    assert set(a["grads"]) == set(b["grads"])


def test_the_backward_finds_the_forwards_layout_in_cloned_scratch_buffers():
    """The three scratch tensors are opaque but ordinary tensors: a caller may clone, checkpoint or move them between forward and
    backward (the reference's own debug path copies them: DGR/__init__.py:105-118).  After a speculative HIT the instance list is
    laid out for the guessed capacity while the call returns the true count; the backward must find that layout in the buffers
    themselves (device header word SGR_HDR_LAYOUT_CAP), not in a side table keyed by the original buffer's address."""
    from sugar_amd import diff_gaussian_rasterization as dgr
    from sugar_amd.diff_gaussian_rasterization import _C
    dev = torch.device(DEV)
    W, H, P = 560, 336, 60_011
    scene = syn.make_scene(P, 17, 0.004, 0.05)
    cam = syn.orbit_cameras(W, H)[3]
    bg = torch.tensor([0.3, 0.2, 0.1], device=dev)
    t = {k: getattr(scene, k).to(dev) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    empty = torch.empty(0, device=dev)
    fwd = (bg, t["means3D"], empty, t["opacities"], t["scales"], t["rotations"], 1.0, empty, cam.viewmatrix.to(dev), cam.projmatrix.to(dev),
           cam.tanfovx, cam.tanfovy, H, W, t["shs"], 3, cam.campos.to(dev), False, False)
    dgr._SPEC_CAP.pop((0, P, W, H), None)
    _C.rasterize_gaussians(*fwd)                                   # learns the count
    R, color, radii, geom, binning, img = _C.rasterize_gaussians(*fwd)
    lf = _C.last_forward
    assert lf["speculative"] and not lf["speculation_missed"] and lf["list_capacity"] > R   # laid out for MORE than R
    g = torch.randn(3, H, W, generator=torch.Generator().manual_seed(5)).to(dev)

    def bwd(geom_, binning_, img_):
        return _C.rasterize_gaussians_backward(bg, t["means3D"], radii, empty, t["scales"], t["rotations"], 1.0, empty, cam.viewmatrix.to(dev),
                                               cam.projmatrix.to(dev), cam.tanfovx, cam.tanfovy, g, t["shs"], 3, cam.campos.to(dev), geom_, R,
                                               binning_, img_, False)
    want = [x.clone() for x in bwd(geom, binning, img)]
    g2, b2, i2 = geom.clone(), binning.clone(), img.clone()
    geom.fill_(0xEE); binning.fill_(0xEE); img.fill_(0xEE)         # the originals are gone
    del geom, binning, img
    got = bwd(g2, b2, i2)
    for a, b in zip(want, got):
        if a.numel():
            assert float((a - b).norm() / a.norm().clamp_min(1e-30)) < 2e-5   # (float atomics: same sums, another order)
