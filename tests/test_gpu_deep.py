"""The eight-wave blend kernel for long lists (csrc/blend.hip: k_blend_fwd_deep; include/sugar_raster.h: sgr_set_deep_min): with a
walk hint the 8x8 blocks of tiles whose hinted list exceeds the threshold are blended by eight waves -- alpha for eight batches in
parallel, then the sequential transmittance chain -- beside the one-wave kernel.  Contract: BIT-IDENTICAL image, final_T, n_contrib;
the backward (which reads the forward's survivor masks) returns the same gradients up to the float-atomic order."""
import numpy as np
import pytest
import torch

import sugar_amd
from sugar_amd import _lib, synthetic as syn
from tests import parity_utils as pu

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("exact", [True, False])
def test_the_eight_wave_kernel_equals_the_one_wave_walk_bit_for_bit(exact):
    from sugar_amd.diff_gaussian_rasterization import grad_sink
    dev = torch.device(DEV)
    W, H = 640, 360
    T = ((W + 15) // 16) * ((H + 15) // 16)
    scene = syn.make_scene(200_000, 33, 0.01, 0.06)
    # faint splats: transmittance decays slowly and every pixel walks thousands of entries (the regime of config 4's silhouette tiles)
    scene = scene._replace(opacities=torch.full_like(scene.opacities, 0.01) + 0.02 * torch.rand(scene.opacities.shape, generator=torch.Generator().manual_seed(1)))
    cam = syn.orbit_cameras(W, H)[2]
    bg = torch.tensor([0.2, 0.4, 0.1])
    g = np.random.default_rng(4).standard_normal((3, H, W)).astype(np.float32)
    hint = torch.zeros(T, dtype=torch.int32, device=dev)
    hdr = torch.zeros(16, dtype=torch.int32).pin_memory()
    ev = torch.cuda.Event()
    sugar_amd.set_exact_alpha(exact)
    old = _lib.load().sgr_get_deep_min()
    try:
        sugar_amd.set_deep_min(0)
        with grad_sink(tile_need_out=hint, header_out=hdr, header_event=ev):
            pu.run_hip(scene, cam, bg)
        assert int(hint.max()) > 1500, int(hint.max())          # lists of thousands of entries are walked
        with grad_sink(tile_need=hint, header_out=hdr, header_event=ev):
            a = pu.run_hip(scene, cam, bg, grad_out=g)           # the one-wave walk
        ev.synchronize()
        sugar_amd.set_deep_min(256)
        with grad_sink(tile_need=hint, header_out=hdr, header_event=ev):
            b = pu.run_hip(scene, cam, bg, grad_out=g)           # blocks beyond 256 hinted entries: eight waves
        ev.synchronize()
        assert int(hdr[8 + 3]) == 0
        n_deep = int((hint.cpu() > 256).sum())
        assert n_deep > 100
        assert np.array_equal(a["color"], b["color"]) and np.array_equal(a["n_contrib"], b["n_contrib"])
        assert np.array_equal(a["final_T"], b["final_T"]) and np.array_equal(a["tile_start"], b["tile_start"])
        for k in a["grads"]:
            if a["grads"][k] is not None:
                assert pu.rel_stats(b["grads"][k], a["grads"][k])["norm_rel"] < 2e-5, k
        # a threshold nothing reaches: the side kernel finds an empty list
        sugar_amd.set_deep_min(1 << 30)
        with grad_sink(tile_need=hint, header_out=hdr, header_event=ev):
            c = pu.run_hip(scene, cam, bg)
        assert np.array_equal(a["color"], c["color"])
    finally:
        sugar_amd.set_deep_min(old)
        sugar_amd.set_exact_alpha(True)
