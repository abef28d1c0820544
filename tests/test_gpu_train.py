"""GPU tests of the train-step plumbing around the rasterizer: gradient sinks, the compact SH-gradient mode
(sgr_backward with dL_dsh == NULL + sgr_sh_grad_from_views) and the trainer using them."""
import numpy as np
import pytest
import torch

from sugar_amd import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _render_grads(scene, cam, bg, g, compact):
    from sugar_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, grad_sink
    dev = torch.device(DEV)
    st = GaussianRasterizationSettings(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy, bg.to(dev), 1.0,
                                       cam.viewmatrix.to(dev), cam.projmatrix.to(dev), 3, cam.campos.to(dev), False, False)
    m = scene.means3D.to(dev).requires_grad_(True); sh = scene.shs.to(dev).requires_grad_(True)
    op = scene.opacities.to(dev).requires_grad_(True); sc = scene.scales.to(dev).requires_grad_(True)
    ro = scene.rotations.to(dev).requires_grad_(True)
    holder = {}
    with grad_sink(compact_sh=compact, out=holder):
        color, _ = GaussianRasterizer(st)(m, torch.zeros_like(m, requires_grad=True), op, shs=sh, scales=sc, rotations=ro)
        grads = torch.autograd.grad((color * g).sum(), [m, sh, op, sc, ro], allow_unused=True)
    return grads, holder


def test_compact_sh_mode_rebuilds_the_sh_gradient_over_views():
    from sugar_amd.train_step import sh_grad_from_views
    dev = torch.device(DEV)
    scene = syn.make_scene(20000, 8, 0.01, 0.08)
    cams = syn.orbit_cameras(320, 200)
    bg = torch.tensor([0.2, 0.3, 0.4])
    g = torch.randn(3, 200, 320, generator=torch.Generator().manual_seed(2)).to(dev)
    full, masked, campos = [], [], []
    for cam in (cams[1], cams[4], cams[6]):
        gr, _ = _render_grads(scene, cam, bg, g, compact=False)
        gc, holder = _render_grads(scene, cam, bg, g, compact=True)
        assert gc[1] is None and "masked_colors" in holder
        for a, b in zip([gr[0], gr[2], gr[3], gr[4]], [gc[0], gc[2], gc[3], gc[4]]):
            # the other gradients do not depend on the mode; two backward runs differ only by atomic ordering
            assert float((a - b).norm() / a.norm()) < 1e-5
        full.append(gr[1]); masked.append(holder["masked_colors"]); campos.append(cam.campos.to(dev))
    ref = full[0] + full[1] + full[2]
    out = torch.empty_like(ref)
    sh_grad_from_views(scene.means3D.to(dev), torch.stack(campos), torch.stack(masked), 3, out)
    err = float((out - ref).norm() / ref.norm())
    assert err < 2e-5, err  # two independent backward runs (atomic ordering) + float summation order
    assert float(ref.abs().max()) > 0


def test_trainer_compact_and_flat_paths_agree():
    from sugar_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from sugar_amd.train_step import GaussianParams, ViewShardedTrainer
    dev = torch.device(DEV)
    scene = syn.make_scene(30000, 5, 0.01, 0.06)
    cams = [c._replace(viewmatrix=c.viewmatrix.to(dev), projmatrix=c.projmatrix.to(dev), campos=c.campos.to(dev))
            for c in syn.orbit_cameras(400, 240)]
    gts = [torch.rand(3, 240, 400, generator=torch.Generator().manual_seed(i)).to(dev) for i in range(3)]
    flats = []
    # flat SH gradient | compact exchange + SH gradient rebuilt | compact exchange + SH gradient consumed inside the Adam kernel
    for compact, fused in ((False, False), (True, False), (True, True)):
        p = GaussianParams(scene, dev)
        tr = ViewShardedTrainer(p, GaussianRasterizer, GaussianRasterizationSettings, torch.zeros(3, device=dev),
                                compact_sh=compact, fused_sh_adam=fused)
        assert tr.fused_sh_adam == fused
        for i in range(3):
            loss, _ = tr.step(cams[i], gts[i])
            assert torch.isfinite(loss)
        flats.append(p.flat.clone())
    start = GaussianParams(scene, dev).flat
    assert float((flats[0] - start).abs().max()) > 1e-4
    # Adam normalises tiny gradients to +-lr, so compare the update, not bit patterns
    for other in flats[1:]:
        assert float((flats[0] - other).abs().max()) < 2e-2 * float((flats[0] - start).abs().max())
        assert float((flats[0] - other).norm() / (flats[0] - start).norm()) < 1e-3
    tr = ViewShardedTrainer(GaussianParams(scene, dev), GaussianRasterizer, GaussianRasterizationSettings, torch.zeros(3, device=dev))
    assert tr.compact_sh and tr.fused_sh_adam and tr.sync_free  # the defaults on a ROCm device


def test_view_direction_term_formed_in_the_sh_adam_kernel_is_the_same_step():
    """SGR_MODE_SH_DIR_ELSEWHERE + sgr_sh_adam_from_views_ex + sgr_adam_step_ex: the backward never reads the SH tensor, the
    SH-Adam kernel forms dRGB/d(view direction) -> dL/dxyz and the flat Adam kernel adds it.  Same arithmetic in the same
    order: the parameters after three steps differ from the path that keeps the term in the backward preprocess kernel by
    no more than two runs of that path differ from each other (the blend backward's float atomics)."""
    from sugar_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from sugar_amd.train_step import GaussianParams, ViewShardedTrainer
    dev = torch.device(DEV)
    scene = syn.make_scene(30001, 5, 0.01, 0.06)
    cams = [c._replace(viewmatrix=c.viewmatrix.to(dev), projmatrix=c.projmatrix.to(dev), campos=c.campos.to(dev))
            for c in syn.orbit_cameras(400, 240)]
    gts = [torch.rand(3, 240, 400, generator=torch.Generator().manual_seed(i)).to(dev) for i in range(3)]
    flats, extras = [], []
    for elsewhere in (False, False, True):
        p = GaussianParams(scene, dev)
        tr = ViewShardedTrainer(p, GaussianRasterizer, GaussianRasterizationSettings, torch.zeros(3, device=dev),
                                sh_dir_in_adam=elsewhere)
        assert tr.fused_sh_adam and tr.sh_dir_in_adam == elsewhere
        for i in range(3):
            loss, _ = tr.step(cams[i], gts[i])
            assert torch.isfinite(loss)
        flats.append(p.flat.clone())
        extras.append(tr.opt._dmean_extra)
    assert extras[0] is None and float(extras[2].abs().max()) > 0  # the term is not identically zero
    start = GaussianParams(scene, dev).flat
    assert float((flats[0] - start).abs().max()) > 1e-4
    # the blend backward's float atomics make two runs of the SAME configuration differ in the last bits, and Adam turns that
    # into +-lr for near-zero gradients: the run-to-run spread is the yardstick
    noise = float((flats[0] - flats[1]).norm())
    diff = float((flats[0] - flats[2]).norm())
    upd = float((flats[0] - start).norm())
    assert diff <= max(3.0 * noise, 3e-4 * upd), (diff, noise, upd)
    assert not ViewShardedTrainer(GaussianParams(scene, dev), GaussianRasterizer, GaussianRasterizationSettings,
                                  torch.zeros(3, device=dev)).sh_dir_in_adam  # an option, off by default (no net gain measured)


def test_fused_activations_match_torch_ops():
    """exp / F.normalize / sigmoid of the raw parameters (gaussian_model.py:92-117) and their autograd gradients"""
    from sugar_amd.train_step import GaussianParams
    dev = torch.device(DEV)
    scene = syn.make_scene(50001, 4, 0.005, 0.2)  # odd count: the tail block is partial
    ws = None
    out = []
    for fused in (False, True):
        p = GaussianParams(scene, dev)
        with torch.no_grad():
            p.params["rotation"].mul_(3.7)            # un-normalised quaternions, as they are mid-training
            p.params["rotation"][:7] = 0.0            # |v| < eps: F.normalize divides by the clamped 1e-12
        a = p.activated(fused=fused)
        if ws is None:
            g = torch.Generator().manual_seed(0)
            ws = {k: torch.randn(a[k].shape, generator=g).to(dev) for k in ("scales", "rotations", "opacities")}
        loss = sum((a[k] * ws[k]).sum() for k in ws)
        grads = torch.autograd.grad(loss, [p.params["scaling"], p.params["rotation"], p.params["opacity"]])
        if fused:  # the gradients were written into the flat gradient buffer and returned as such
            for gr, name in zip(grads, ("scaling", "rotation", "opacity")):
                assert gr.data_ptr() == p.params[name].grad.data_ptr()
        out.append(([a[k].detach().clone() for k in ("scales", "rotations", "opacities")], [x.clone() for x in grads]))
    for x, y in zip(out[0][0] + out[0][1], out[1][0] + out[1][1]):
        assert torch.isfinite(y).all()
        # (the seven zero quaternions give 1e12-sized gradients: compare them apart from the regular rows)
        assert torch.allclose(x[:7], y[:7], rtol=2e-5, atol=1e-6 * float(x[:7].abs().max()))
        assert torch.allclose(x[7:], y[7:], rtol=2e-5, atol=2e-6 * float(x[7:].abs().max())), float((x[7:] - y[7:]).abs().max())


def test_view_stride_of_the_gathered_buffer():
    """the all-gather buffer holds P + 1 rows per rank (colours, then the camera centre): the kernels read it in place"""
    from sugar_amd.train_step import sh_grad_from_views, _view_rows
    dev = torch.device(DEV)
    g = torch.Generator().manual_seed(5)
    P, V = 7001, 3
    means = torch.randn(P, 3, generator=g).to(dev)
    buf = torch.randn(V, P + 1, 3, generator=g).to(dev)
    cams = buf[:, P].contiguous()
    view = buf[:, :P]
    assert not view.is_contiguous() and _view_rows(view)[1] == P + 1
    a = sh_grad_from_views(means, cams, view, 3, torch.empty(P, 16, 3, device=dev))
    b = sh_grad_from_views(means, cams, view.contiguous(), 3, torch.empty(P, 16, 3, device=dev))
    assert torch.equal(a, b) and float(a.abs().max()) > 0


def test_two_phase_backward_hands_over_final_colour_gradients():
    """sgr_backward_phase: the masked colour gradients seen by the `on_colors` hook (between the blend half and the
    preprocess half) are the final ones, and every other gradient equals the one-call backward's"""
    from sugar_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, grad_sink
    dev = torch.device(DEV)
    scene = syn.make_scene(30000, 9, 0.01, 0.08)
    cam = syn.orbit_cameras(320, 200)[2]
    bg = torch.tensor([0.2, 0.3, 0.4])
    g = torch.randn(3, 200, 320, generator=torch.Generator().manual_seed(3)).to(dev)
    ref, holder_ref = _render_grads(scene, cam, bg, g, compact=True)
    st = GaussianRasterizationSettings(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy, bg.to(dev), 1.0,
                                       cam.viewmatrix.to(dev), cam.projmatrix.to(dev), 3, cam.campos.to(dev), False, False)
    m = scene.means3D.to(dev).requires_grad_(True); sh = scene.shs.to(dev).requires_grad_(True)
    op = scene.opacities.to(dev).requires_grad_(True); sc = scene.scales.to(dev).requires_grad_(True)
    ro = scene.rotations.to(dev).requires_grad_(True)
    seen, holder = [], {}
    send = torch.full((scene.means3D.shape[0], 3), float("nan"), device=dev)

    def on_colors(c):
        assert c.data_ptr() == send.data_ptr()   # written straight into the caller's buffer
        seen.append(c.clone())                   # a snapshot at hand-over time (stream-ordered)

    with grad_sink(compact_sh=True, out=holder, colors=send, on_colors=on_colors):
        color, _ = GaussianRasterizer(st)(m, torch.zeros_like(m, requires_grad=True), op, shs=sh, scales=sc, rotations=ro)
        grads = torch.autograd.grad((color * g).sum(), [m, sh, op, sc, ro], allow_unused=True)
    assert len(seen) == 1 and grads[1] is None
    rel = lambda a, b: float((a - b).norm() / b.norm())
    assert rel(seen[0], holder_ref["masked_colors"]) < 1e-5
    assert torch.equal(seen[0], send) and torch.equal(holder["masked_colors"], send)  # the second half did not touch them
    for a, b in zip([grads[0], grads[2], grads[3], grads[4]], [ref[0], ref[2], ref[3], ref[4]]):
        assert rel(a, b) < 1e-5


def test_sync_free_forward_matches_and_recovers_from_a_small_capacity():
    """sgr_forward_ex with a binning capacity: same image and lists as the forward with the host round trip; a capacity
    that is too small is flagged in the header (and leaves the outputs alone) so the caller can repeat the forward"""
    from sugar_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, grad_sink, _C
    from sugar_amd import _lib
    dev = torch.device(DEV)
    scene = syn.make_scene(30000, 10, 0.01, 0.08)
    cam = syn.orbit_cameras(320, 200)[5]
    st = GaussianRasterizationSettings(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy, torch.zeros(3, device=dev),
                                       1.0, cam.viewmatrix.to(dev), cam.projmatrix.to(dev), 3, cam.campos.to(dev), False, False)
    args = dict(means3D=scene.means3D.to(dev), means2D=torch.zeros(30000, 3, device=dev), opacities=scene.opacities.to(dev),
                shs=scene.shs.to(dev), scales=scene.scales.to(dev), rotations=scene.rotations.to(dev))
    with torch.no_grad():
        color0, _ = GaussianRasterizer(st)(**args)
    R = _C.last_forward["num_rendered"]
    list0 = _C.last_forward["binning"][: 4 * R].clone()
    hdr, ev = torch.zeros(16, dtype=torch.int32).pin_memory(), torch.cuda.Event()
    for cap, ok in ((R + 1000, True), (R, True), (R - 1, False), (R // 3, False)):
        with torch.no_grad(), grad_sink(binning_capacity=cap, header_out=hdr, header_event=ev):
            color, _ = GaussianRasterizer(st)(**args)
        ev.synchronize()
        assert int(hdr[0]) == R and int(hdr[6]) == 0 and int(hdr[8]) == R and int(hdr[8 + 3]) == 0  # both header copies
        assert _C.last_forward["num_rendered"] == cap and _C.last_forward["sync_free"]
        if ok:
            assert torch.equal(color, color0)
            assert torch.equal(_C.last_forward["binning"][: 4 * R], list0)
        else:
            # a backward on the invalid forward is a no-op on the device: no fault, no atomics into the accumulators
            leaves = {k: v.clone().requires_grad_(True) for k, v in args.items()}
            with grad_sink(binning_capacity=cap, header_out=hdr, header_event=ev):
                c2, _ = GaussianRasterizer(st)(**leaves)
                c2.sum().backward()
            torch.cuda.synchronize()
            assert float(leaves["opacities"].grad.abs().max()) == 0.0 and float(leaves["means3D"].grad.abs().max()) == 0.0
    # the trainer notices and repeats the forward
    from sugar_amd.train_step import GaussianParams, ViewShardedTrainer
    cams = [c._replace(viewmatrix=c.viewmatrix.to(dev), projmatrix=c.projmatrix.to(dev), campos=c.campos.to(dev))
            for c in syn.orbit_cameras(320, 200)]
    gt = torch.rand(3, 200, 320, generator=torch.Generator().manual_seed(3)).to(dev)
    res = []
    for shrink in (False, True):
        p = GaussianParams(scene, dev)
        tr = ViewShardedTrainer(p, GaussianRasterizer, GaussianRasterizationSettings, torch.zeros(3, device=dev))
        assert tr.sync_free
        for i in range(3):
            if shrink and i == 2:
                tr._bin_cap = 1000  # far too small: step 3 must detect it and run the forward again
            tr.step(cams[i], gt)
        assert tr.redone == (1 if shrink else 0) and tr.last_num_rendered > 1000
        res.append(p.flat.clone())
    start = GaussianParams(scene, dev).flat
    assert float((res[0] - res[1]).norm() / (res[0] - start).norm()) < 1e-3


def test_raw_parameter_mode_equals_separate_activations():
    """sgr_forward_ex / sgr_backward_phase in raw-parameter mode == activations kernels + rasterizer"""
    from sugar_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from sugar_amd.train_step import GaussianParams, ViewShardedTrainer
    dev = torch.device(DEV)
    scene = syn.make_scene(30000, 6, 0.01, 0.06)
    cams = [c._replace(viewmatrix=c.viewmatrix.to(dev), projmatrix=c.projmatrix.to(dev), campos=c.campos.to(dev))
            for c in syn.orbit_cameras(400, 240)]
    gts = [torch.rand(3, 240, 400, generator=torch.Generator().manual_seed(i)).to(dev) for i in range(3)]
    flats, grads, images = [], [], []
    for fuse in (False, True):
        p = GaussianParams(scene, dev)
        with torch.no_grad():
            p.params["rotation"].mul_(2.5)  # un-normalised quaternions, as they are mid-training
        tr = ViewShardedTrainer(p, GaussianRasterizer, GaussianRasterizationSettings, torch.zeros(3, device=dev), fuse_activations=fuse)
        assert tr.fuse_activations == fuse
        _, pkg = tr.step(cams[0], gts[0])
        images.append(pkg["render"].detach().clone()); grads.append(p.flat_grad[: p.n_small].clone())
        for i in (1, 2):
            tr.step(cams[i], gts[i])
        flats.append(p.flat.clone())
    # (expf inlines differently into the two translation units -- one is built without FMA contraction --, so the activated
    # values can differ in the last bit)
    assert float((images[0] - images[1]).norm() / images[0].norm()) < 1e-5
    # the backward sums with atomics in a run-dependent order: compare norm-wise
    assert float((grads[0] - grads[1]).norm() / grads[0].norm()) < 1e-5
    start = GaussianParams(scene, dev).flat
    assert float((flats[0] - flats[1]).norm() / (flats[0] - start).norm()) < 1e-3
