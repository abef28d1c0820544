"""GPU tests of the native train step (sgr_trainer_*, sugar_amd.train_step.NativeTrainer), the walk hint of the list-write
pass (sgr_forward_opts.tile_need) and the fused densification statistics (sgr_backward_opts)."""
import numpy as np
import pytest
import torch

from sugar_amd import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cams(W, H):
    dev = torch.device(DEV)
    return [c._replace(viewmatrix=c.viewmatrix.to(dev), projmatrix=c.projmatrix.to(dev), campos=c.campos.to(dev))
            for c in syn.orbit_cameras(W, H)]


def _close(a, b, start):
    """two runs of the same steps: equal up to the float-atomic order of the blend backward (Adam turns the sign of a
    near-zero gradient into a +-lr step, so a handful of parameters may sit a full update apart)"""
    upd = float((b - start).abs().max())
    assert upd > 1e-4
    assert float(((a - b).abs() > 1e-2 * upd).float().mean()) < 1e-4
    assert float((a - b).norm() / (b - start).norm()) < 1e-3


def test_native_step_equals_the_autograd_trainer_and_repairs_itself():
    """Twelve steps over eight cameras (every camera is revisited: the second visits run with the walk hint), started with a
    list capacity that is far too small (the first forward overflows, the device skips that step, the host repeats it)."""
    from sugar_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from sugar_amd.train_step import GaussianParams, NativeTrainer, ViewShardedTrainer
    dev = torch.device(DEV)
    W, H = 400, 240
    scene = syn.make_scene(30000, 5, 0.01, 0.06)
    cams = _cams(W, H)
    gts = [torch.rand(3, H, W, generator=torch.Generator().manual_seed(i)).to(dev) for i in range(8)]
    pa = GaussianParams(scene, dev)
    start = pa.flat.clone()
    ref = ViewShardedTrainer(pa, GaussianRasterizer, GaussianRasterizationSettings, torch.zeros(3, device=dev))
    ref_losses = [float(ref.step(cams[i % 8], gts[i % 8])[0]) for i in range(12)]
    results = {}
    for name, kw in (("hint", dict(capacity=1000)), ("nohint", dict(walk_hint=False))):
        pb = GaussianParams(scene, dev)
        nt = NativeTrainer(pb, torch.zeros(3), W, H, **kw)
        losses = []
        for i in range(12):
            loss = nt.step(cams[i % 8], gts[i % 8], cam_key=i % 8)
            nt.synchronize()  # (validates the step, so that the loss read here belongs to a valid forward)
            losses.append(float(loss))
        if name == "hint":
            assert nt.redone >= 1 and nt.capacity > 1000          # the capacity repair ran
            assert all(ent[1] for ent in nt._hints.values())     # hints in use from the second visit on
        assert np.allclose(losses, ref_losses, rtol=2e-4), (losses, ref_losses)
        _close(pb.flat, pa.flat, start)
        results[name] = pb.flat.clone()
    _close(results["hint"], results["nohint"], start)


def test_native_step_without_the_job_in_the_loss_kernel(monkeypatch):
    """sgr_trainer_step normally lets the loss forward kernel carry the rasterizer's post-blend bookkeeping (launch order, walk hint,
    second header copy: SGR_FLAG_DEFER_POST).  When the pinned header is not mapped into the device's address space the forward
    does that itself; SGR_TRAINER_NO_DEFER forces that path: same training, hints and launch orders in use, no repairs."""
    from sugar_amd.train_step import GaussianParams, NativeTrainer
    dev = torch.device(DEV)
    W, H = 400, 240
    scene = syn.make_scene(30000, 7, 0.01, 0.06)
    cams = _cams(W, H)
    gts = [torch.rand(3, H, W, generator=torch.Generator().manual_seed(i)).to(dev) for i in range(8)]
    flats, losses = [], []
    for no_defer in (False, True):
        if no_defer:
            monkeypatch.setenv("SGR_TRAINER_NO_DEFER", "1")
        p = GaussianParams(scene, dev)
        start = p.flat.clone()
        nt = NativeTrainer(p, torch.zeros(3), W, H)
        ls = []
        for i in range(20):
            loss = nt.step(cams[i % 8], gts[i % 8], cam_key=i % 8)
            nt.synchronize()
            ls.append(float(loss))
        assert nt.redone == 0
        assert all(ent[1] and ent[4] for ent in nt._hints.values())   # walk hints and launch orders validated and in use
        for ent in nt._hints.values():
            assert np.array_equal(np.sort(ent[3].cpu().numpy()), np.arange(nt.T))   # every stored order is a permutation
        flats.append(p.flat.clone()); losses.append(ls)
    assert np.allclose(losses[0], losses[1], rtol=2e-4)
    _close(flats[1], flats[0], start)


def test_hints_that_are_too_short_are_repaired_in_place_and_a_capacity_miss_is_repeated():
    """Without synchronising after every step the validity check lags one step behind.  (a) Walk hints that have become far too
    short (forced here: every tile may walk ONE entry) no longer cost the step: the tiles that outrun their hint are listed on the
    device and rendered again inside the same forward -- no step is repeated, the training is the same.  (b) A list capacity that
    has become too small still makes the step a no-op on the device, and the trainer repeats it before the next one."""
    from sugar_amd.train_step import GaussianParams, NativeTrainer
    dev = torch.device(DEV)
    W, H = 400, 240
    scene = syn.make_scene(30000, 6, 0.01, 0.06)
    cams = _cams(W, H)
    gts = [torch.rand(3, H, W, generator=torch.Generator().manual_seed(i)).to(dev) for i in range(8)]
    flats = []
    for sabotage in (None, "hints", "capacity"):
        p = GaussianParams(scene, dev)
        start = p.flat.clone()
        nt = NativeTrainer(p, torch.zeros(3), W, H)
        for i in range(16):
            if sabotage and i == 10:
                nt.synchronize()
                if sabotage == "hints":
                    for ent in nt._hints.values():
                        ent[0].fill_(1)
                else:
                    nt.capacity = 1000
                    nt._lib.sgr_trainer_set_binning(nt._h, nt._binning.data_ptr(), nt._binning.numel(), 1000)
            nt.step(cams[i % 8], gts[i % 8], cam_key=i % 8)
        nt.synchronize()
        if sabotage == "hints":
            assert nt.redone == 0 and nt.hint_pauses == 0
            assert nt.repaired_tiles > 100          # most tiles of the six sabotaged views walk more than one entry
        elif sabotage == "capacity":
            assert nt.redone >= 1 and nt.capacity > 1000
        else:
            assert nt.redone == 0 and nt.repaired_tiles == 0
        flats.append(p.flat.clone())
    _close(flats[1], flats[0], start)
    _close(flats[2], flats[0], start)


def test_walk_hint_leaves_ranges_and_the_walked_prefix_bit_identical():
    """sgr_forward_opts.tile_need through the reference-shaped API: same image bit for bit, same ranges and num_rendered, and
    every tile's list identical to the unhinted one over the entries the tile walks (rasterizer_impl.cu:70-138 order)."""
    from sugar_amd import _lib
    from sugar_amd.diff_gaussian_rasterization import _C, grad_sink
    from tests import parity_utils as pu
    lib = _lib.load()
    dev = torch.device(DEV)
    scene = syn.make_scene(150000, 21, 0.004, 0.05)
    cam = syn.orbit_cameras(1000, 600)[3]
    W, H = 1000, 600
    T = ((W + 15) // 16) * ((H + 15) // 16)
    bg = torch.tensor([0.1, 0.2, 0.3])
    hint = torch.zeros(T, dtype=torch.int32, device=dev)
    hdr = torch.zeros(16, dtype=torch.int32).pin_memory()
    ev = torch.cuda.Event()
    with grad_sink(tile_need_out=hint, header_out=hdr, header_event=ev):
        a = pu.run_hip(scene, cam, bg)
    off = lib.sgr_img_tile_walked_offset(W, H)
    walked = _C.last_forward["img"][off: off + 4 * T].view(torch.int32).clone()
    assert torch.equal(hint, walked + (walked >> 2) + 64)
    g = np.random.default_rng(0).standard_normal((3, H, W)).astype(np.float32)
    with grad_sink(tile_need=hint, header_out=hdr, header_event=ev):
        b = pu.run_hip(scene, cam, bg, grad_out=g)
    ev.synchronize()
    assert int(hdr[8 + 3]) == 0 and int(hdr[0]) == a["num_rendered"]
    assert b["num_rendered"] == a["num_rendered"] and np.array_equal(a["tile_start"], b["tile_start"])
    assert np.array_equal(a["color"], b["color"]) and np.array_equal(a["n_contrib"], b["n_contrib"])
    ts, w = a["tile_start"].astype(np.int64), walked.cpu().numpy().astype(np.int64)
    keep = np.zeros(a["num_rendered"], dtype=bool)
    for t in np.nonzero(w)[0]:
        keep[ts[t]: ts[t] + w[t]] = True
    assert keep.sum() < 0.9 * keep.size  # the hint is worth something on this scene
    assert np.array_equal(a["point_list"][keep], b["point_list"][keep])
    # gradients agree with the unhinted forward's (the backward only reads what the forward walked)
    c = pu.run_hip(scene, cam, bg, grad_out=g)
    for k in c["grads"]:
        assert pu.rel_stats(b["grads"][k], c["grads"][k])["norm_rel"] < 2e-5, k
    # a hint that is too short on SOME tiles (here: a third of what 300 of them walk) is repaired inside the forward: those tiles
    # are rendered again over their full lists -- same image bit for bit, same gradients, header word 7 counts them
    some = hint.clone()
    idx = torch.argsort(walked, descending=True)[:300]
    some[idx] = torch.clamp(walked[idx] // 3, min=1)
    with grad_sink(tile_need=some, header_out=hdr, header_event=ev):
        d = pu.run_hip(scene, cam, bg, grad_out=g)
    ev.synchronize()
    assert int(hdr[8 + 3]) == 0 and 250 <= int(hdr[8 + 7]) <= 300, (int(hdr[8 + 3]), int(hdr[8 + 7]))
    assert np.array_equal(a["color"], d["color"]) and np.array_equal(a["n_contrib"], d["n_contrib"])
    for k in c["grads"]:
        assert pu.rel_stats(d["grads"][k], c["grads"][k])["norm_rel"] < 2e-5, k
    # more tiles than one repair launch covers (1024): reported, not rendered wrongly
    short = torch.ones_like(hint)
    with grad_sink(tile_need=short, header_out=hdr, header_event=ev):
        pu.run_hip(scene, cam, bg)
    ev.synchronize()
    assert int(hdr[8 + 3]) != 0


def test_launch_order_is_a_permutation_deepest_first_and_does_not_change_the_result():
    """sgr_forward_opts.tile_order / tile_order_out: the order written by a forward is a permutation of the tiles sorted by the
    depth of their deepest contributor (1024 classes), a forward launched in that order returns the same image bit for bit, and
    a backward told that the order is ready (SGR_BWD_TILE_ORDER_READY) returns the gradients of one that sorts itself."""
    from sugar_amd import _lib
    from sugar_amd.diff_gaussian_rasterization import _C, grad_sink
    from tests import parity_utils as pu
    lib = _lib.load()
    dev = torch.device(DEV)
    scene = syn.make_scene(150000, 21, 0.004, 0.05)
    W, H = 1000, 600
    cam = syn.orbit_cameras(W, H)[3]
    T = ((W + 15) // 16) * ((H + 15) // 16)
    bg = torch.tensor([0.1, 0.2, 0.3])
    g = np.random.default_rng(1).standard_normal((3, H, W)).astype(np.float32)
    order = torch.zeros(T, dtype=torch.int32, device=dev)
    a = pu.run_hip(scene, cam, bg, grad_out=g)
    with grad_sink(tile_order_out=order):
        b = pu.run_hip(scene, cam, bg, grad_out=g)          # sorts in the forward, the backward reuses it
    off = lib.sgr_img_tile_maxc_offset(W, H)
    o = order.cpu().numpy().astype(np.int64)
    assert np.array_equal(np.sort(o), np.arange(T))
    # (tile_maxc of the forward is gone -- its array holds the backward's order -- so the depth comes from n_contrib)
    depth = b["n_contrib"].reshape(H, W)
    per_tile = np.zeros(T, dtype=np.int64)
    gx = (W + 15) // 16
    for ty in range((H + 15) // 16):
        for tx in range(gx):
            per_tile[ty * gx + tx] = depth[16 * ty: 16 * ty + 16, 16 * tx: 16 * tx + 16].max()
    mc = int(np.diff(a["tile_start"].astype(np.int64)).max())
    shift = max(mc.bit_length() - 10, 0) if mc >= 1024 else 0
    cls = per_tile[o] >> shift
    assert np.all(np.diff(cls) <= 0), "not sorted by depth class, deepest first"
    assert np.array_equal(a["color"], b["color"])
    with grad_sink(tile_order=order, tile_order_out=order):   # in place: launched in the order of the previous visit
        c = pu.run_hip(scene, cam, bg, grad_out=g)
    assert np.array_equal(a["color"], c["color"]) and np.array_equal(a["n_contrib"], c["n_contrib"])
    assert np.array_equal(np.sort(order.cpu().numpy()), np.arange(T))
    for k in a["grads"]:
        assert pu.rel_stats(b["grads"][k], a["grads"][k])["norm_rel"] < 2e-5, k   # (atomic summation order differs)
        assert pu.rel_stats(c["grads"][k], a["grads"][k])["norm_rel"] < 2e-5, k


def test_fused_densification_statistics():
    """sgr_backward_opts against the reference's own bookkeeping (gaussian_splatting/train.py:111-123,
    scene/gaussian_model.py:405-407) applied to the gradients the plain API returns"""
    from sugar_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, grad_sink
    from sugar_amd.train_step import GaussianParams, NativeTrainer
    dev = torch.device(DEV)
    W, H = 320, 200
    scene = syn.make_scene(20000, 9, 0.01, 0.08)
    cams = _cams(W, H)
    gts = [torch.rand(3, H, W, generator=torch.Generator().manual_seed(i)).to(dev) for i in range(3)]
    P = 20000
    # reference bookkeeping on the plain API: two views, parameters fixed
    max_radii, accum, denom = torch.zeros(P, device=dev), torch.zeros(P, 1, device=dev), torch.zeros(P, 1, device=dev)
    m, op, sh, sc, ro = (t.to(dev).requires_grad_(True) for t in (scene.means3D, scene.opacities, scene.shs, scene.scales, scene.rotations))
    fused = [torch.zeros(P, device=dev) for _ in range(3)]
    for i in range(2):
        cam = cams[i]
        st = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, torch.zeros(3, device=dev), 1.0, cam.viewmatrix, cam.projmatrix, 3,
                                           cam.campos, False, False)
        for use_fused in (False, True):
            vs = torch.zeros(P, 3, device=dev, requires_grad=True)
            with grad_sink(dens_stats=tuple(fused) if use_fused else None):
                img, radii = GaussianRasterizer(st)(m, vs, op, shs=sh, scales=sc, rotations=ro)
                ((img - gts[i]) ** 2).sum().backward()
            if not use_fused:
                vis = radii > 0
                max_radii[vis] = torch.max(max_radii[vis], radii[vis].float())                 # train.py:114
                accum[vis] += torch.norm(vs.grad[vis, :2], dim=-1, keepdim=True)               # gaussian_model.py:406
                denom[vis] += 1                                                                # gaussian_model.py:407
    assert torch.equal(fused[0], max_radii) and torch.equal(fused[2], denom[:, 0])
    assert float((fused[1] - accum[:, 0]).norm() / accum.norm()) < 2e-5  # (two backward runs: atomic order)
    assert float(denom.sum()) > P // 4
    # and through the native trainer
    p = GaussianParams(scene, dev)
    nt = NativeTrainer(p, torch.zeros(3), W, H, densify_stats=True)
    nt.step(cams[0], gts[0], cam_key=0)
    nt.synchronize()
    assert float(nt.denom.sum()) == float((nt.radii > 0).sum()) > 0
    assert torch.equal(nt.max_radii2D, nt.radii.float().clamp_min(0))
    ref = nt.viewspace_grad[:, :2].norm(dim=-1)
    assert float((nt.xyz_gradient_accum - ref).abs().max()) <= 1e-6 * float(ref.max())


def test_densification_statistics_count_a_repaired_step_once():
    """A forward that outgrows the list capacity is a device no-op for EVERY later kernel of the step, the backward preprocess
    kernel and its fused statistics included (round-3 advisor finding: `denom` was incremented by the invalid attempt AND by the
    repeat, diluting xyz_gradient_accum / denom against train.py:111-123).  Capacity far too small on the first step: after N
    completed steps on one camera every visible Gaussian has denom == N, exactly as with ample capacity."""
    from sugar_amd.train_step import GaussianParams, NativeTrainer
    dev = torch.device(DEV)
    W, H = 320, 200
    scene = syn.make_scene(20000, 9, 0.01, 0.08)
    cam = _cams(W, H)[0]
    gt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(0)).to(dev)
    out = []
    for capacity in (None, 20000):  # ample; too small (this view renders several hundred thousand instances)
        p = GaussianParams(scene, dev)
        nt = NativeTrainer(p, torch.zeros(3), W, H, densify_stats=True, capacity=capacity)
        for _ in range(3):
            nt.step(cam, gt, cam_key=0)
        nt.synchronize()
        out.append((nt.redone, nt.denom.clone(), nt.xyz_gradient_accum.clone(), nt.max_radii2D.clone(), p.flat.detach().clone()))
    (r0, d0, a0, m0, f0), (r1, d1, a1, m1, f1) = out
    assert r0 == 0 and r1 >= 1
    assert float(d0.max()) == 3.0 and torch.equal(d0, d1)
    assert torch.equal(m0, m1)
    assert float((a0 - a1).norm() / a0.norm()) < 1e-4  # (atomic order of the blend backward)
    assert float((f0 - f1).abs().max()) < 1e-3


def test_densify_and_resize_keep_training_and_follow_the_reference_model():
    """NativeTrainer.densify_and_prune (sugar_amd/densify.py on the trainer's own buffers, then `resize`) against the reference's
    GaussianModel.densify_and_prune (gaussian_model.py:350-403) started from the same parameters, moments and statistics under the
    same seed: same clone / split / prune decisions, same new parameters and moments; and the trainer keeps stepping on the new
    topology with its hints and launch orders (per tile) kept."""
    from tests import ref_env
    from sugar_amd.train_step import GaussianParams, NativeTrainer
    dev = torch.device(DEV)
    W, H = 320, 200
    scene = syn.make_scene(20000, 9, 0.004, 0.08)
    cams = _cams(W, H)
    gts = [torch.rand(3, H, W, generator=torch.Generator().manual_seed(i)).to(dev) for i in range(8)]
    p = GaussianParams(scene, dev)
    nt = NativeTrainer(p, torch.zeros(3), W, H, densify_stats=True)
    for i in range(10):
        nt.step(cams[i % 8], gts[i % 8], cam_key=i % 8)
    nt.synchronize()
    P0 = p.P
    kw = dict(max_grad=float(torch.quantile((nt.xyz_gradient_accum / nt.denom.clamp_min(1))[nt.denom > 0], 0.9)),
              min_opacity=0.05, extent=3.0, max_screen_size=20)
    snap = dict(raw={k: v.clone() for k, v in p.raw().items()}, m1={k: v.clone() for k, v in p.split_flat(nt.exp_avg).items()},
                m2={k: v.clone() for k, v in p.split_flat(nt.exp_avg_sq).items()},
                stats=dict(xyz_gradient_accum=nt.xyz_gradient_accum.clone(), denom=nt.denom.clone(), max_radii2D=nt.max_radii2D.clone()))
    hints_before = {k: v[0].clone() for k, v in nt._hints.items()}
    nc, ns, npr = nt.densify_and_prune(seed=77, **kw)
    newp = nt.params
    assert nc > 0 and ns > 0 and npr > 0 and newp.P == P0 + nc + ns - npr and newp is not p
    assert float(nt.denom.sum()) == 0.0 and nt.radii.shape[0] == newp.P
    assert all(torch.equal(nt._hints[k][0], v) for k, v in hints_before.items())
    # ---- the reference's own GaussianModel from the same state (when its Python is on this box)
    if ref_env.reference_root() is not None:
        _, GaussianModel, _, _ = ref_env.import_gaussian_splatting()
        gm = GaussianModel(3)
        raw = snap["raw"]
        mk = lambda t: torch.nn.Parameter(t.clone().requires_grad_(True))
        gm._xyz, gm._opacity, gm._scaling, gm._rotation = mk(raw["xyz"]), mk(raw["opacity"]), mk(raw["scaling"]), mk(raw["rotation"])
        gm._features_dc, gm._features_rest = mk(raw["features"][:, :1]), mk(raw["features"][:, 1:])
        gm.percent_dense = 0.01
        groups = [("xyz", gm._xyz), ("f_dc", gm._features_dc), ("f_rest", gm._features_rest), ("opacity", gm._opacity),
                  ("scaling", gm._scaling), ("rotation", gm._rotation)]
        gm.optimizer = torch.optim.Adam([{"params": [t], "lr": 1e-3, "name": n} for n, t in groups], lr=0.0, eps=1e-15)
        mom = lambda d, n: (d["features"][:, :1] if n == "f_dc" else d["features"][:, 1:] if n == "f_rest" else d[n]).clone()
        for n, t in groups:
            gm.optimizer.state[t] = dict(step=torch.tensor(10.0), exp_avg=mom(snap["m1"], n), exp_avg_sq=mom(snap["m2"], n))
        gm.xyz_gradient_accum = snap["stats"]["xyz_gradient_accum"].clone().reshape(-1, 1)
        gm.denom = snap["stats"]["denom"].clone().reshape(-1, 1)
        gm.max_radii2D = snap["stats"]["max_radii2D"].clone()
        torch.manual_seed(77)   # (densify_and_split draws from the global generator, :360)
        gm.densify_and_prune(kw["max_grad"], kw["min_opacity"], kw["extent"], kw["max_screen_size"])
        assert gm._xyz.shape[0] == newp.P
        got = newp.raw()
        # decisions and row order: every tensor that is not drawn at random is bit-identical; the positions too, except (at most) the
        # 2 x n_split rows sampled with torch.normal, which are the reference's draws when both generators are in the same state
        assert torch.equal(got["scaling"], gm._scaling.detach())
        differing = int((got["xyz"] != gm._xyz.detach()).any(dim=1).sum())
        assert differing <= 2 * ns, (differing, ns)
        if differing:
            assert float((got["xyz"] - gm._xyz.detach()).abs().max()) < 3.0 * float(torch.exp(got["scaling"]).max()) * 8
        assert torch.equal(got["rotation"], gm._rotation.detach()) and torch.equal(got["opacity"], gm._opacity.detach())
        assert torch.equal(got["features"], torch.cat((gm._features_dc, gm._features_rest), dim=1).detach())
        m1 = newp.split_flat(nt.exp_avg)
        assert torch.equal(m1["xyz"], gm.optimizer.state[gm._xyz]["exp_avg"])
        assert torch.equal(newp.split_flat(nt.exp_avg_sq)["opacity"], gm.optimizer.state[gm._opacity]["exp_avg_sq"])
    # ---- the trainer goes on: losses keep falling over two more passes, statistics count again
    losses = []
    for i in range(16):
        loss = nt.step(cams[i % 8], gts[i % 8], cam_key=i % 8)
        nt.synchronize()
        losses.append(float(loss))
    assert np.mean(losses[8:]) < np.mean(losses[:8]) and float(nt.denom.max()) == 16.0   # (a Gaussian in front of all eight cameras was counted by every step)
