"""CPU tests (gloo, world size 2) of the multi-GPU semantics SURVEY.md section 8(e) lists beyond the vanilla gradient exchange:

  * densification across ranks: `xyz_gradient_accum` / `denom` all-reduce-SUM, `max_radii2D` all-reduce-MAX
    (sugar_densifier.py:156-164, train.py:111-123), then the identical, identically seeded clone / split / prune on every rank
    (sugar_amd.densify, restating gaussian_model.py:350-403): replicas stay BIT-identical across the event and equal
    single-process sequential accumulation of the same views;
  * a view-sharded gradient exchange for arbitrary parameter lists (sugar_amd.view_parallel.attach: a step pre-hook on the optimiser
    the trainer built) on the REFERENCE'S OWN surface-bound `SuGaR` class -- the refine-mode model of BASELINE config 4, whose
    parameters include the mesh vertices `_points[n_verts,3]` (sugar_model.py:222, refine.py:786-808) -- with the trainer code
    untouched.
The rasterizer underneath is the oracle-backed stand-in (tests/oracle_rasterizer.py): what is under test is host logic."""
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sugar_amd import densify, synthetic as syn, view_parallel as vp
from sugar_amd.train_step import GaussianParams, _torch_adam, photometric_loss, render
from tests import ref_env

P, W, H, STEPS = 400, 48, 32, 2
DENS = dict(max_grad=2e-5, min_opacity=0.02, extent=3.0, max_screen_size=20)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _setup():
    from tests import oracle_rasterizer as orast
    torch.set_num_threads(1)
    scene = syn.make_scene(P, 23, 0.01, 0.12)
    cams = syn.orbit_cameras(W, H)
    g = torch.Generator().manual_seed(9)
    gts = [torch.rand(3, H, W, generator=g) for _ in cams]
    return orast, scene, cams, gts


def _train_and_densify(world, rank, views_of_step):
    """STEPS optimisation steps with the reference's statistics bookkeeping, then one densification event.  `views_of_step(s)` lists
    the cameras THIS process renders in step s (one per rank when sharded; all of the batch when it accumulates alone)."""
    orast, scene, cams, gts = _setup()
    params = GaussianParams(scene, torch.device("cpu"))
    opt = _torch_adam(params)
    handle = vp.attach(opt)
    stats = dict(xyz_gradient_accum=torch.zeros(P), denom=torch.zeros(P), max_radii2D=torch.zeros(P))
    n_batch = 2
    for s in range(STEPS):
        params.flat_grad.zero_()
        for k in views_of_step(s):
            pkg = render(params, cams[k], torch.zeros(3), orast.GaussianRasterizer, orast.GaussianRasterizationSettings)
            scale = 1.0 if world > 1 else 1.0 / n_batch      # (sharded: the hook averages over the ranks)
            (photometric_loss(pkg["render"], gts[k]) * scale).backward()
            vis, radii = pkg["visibility_filter"], pkg["radii"]
            # train.py:111-114 / gaussian_model.py:405-407
            stats["max_radii2D"][vis] = torch.max(stats["max_radii2D"][vis], radii[vis].float())
            stats["xyz_gradient_accum"][vis] += torch.norm(pkg["viewspace_points"].grad[vis, :2], dim=-1) / scale
            stats["denom"][vis] += 1
        opt.step()
    handle.remove()
    vp.all_reduce_densification_stats(stats)
    m1 = {k: opt.state[p]["exp_avg"] for k, p in _named(opt)}
    m2 = {k: opt.state[p]["exp_avg_sq"] for k, p in _named(opt)}
    gen = torch.Generator().manual_seed(1234)  # every rank seeds alike (sugar_densifier.py:206 draws from the common global seed)
    raw = params.raw()
    mom1 = dict(xyz=m1["xyz"], opacity=m1["opacity"], scaling=m1["scaling"], rotation=m1["rotation"],
                features=torch.cat((m1["f_dc"], m1["f_rest"]), dim=1))
    mom2 = dict(xyz=m2["xyz"], opacity=m2["opacity"], scaling=m2["scaling"], rotation=m2["rotation"],
                features=torch.cat((m2["f_dc"], m2["f_rest"]), dim=1))
    t, a, b, nc, ns, npr = densify.densify_and_prune(raw, mom1, mom2, stats, generator=gen, **DENS)
    new = GaussianParams.from_raw(t, torch.device("cpu"))
    return dict(flat=new.flat.detach().numpy().copy(), m1=torch.cat([a[k].reshape(-1) for k in densify.NAMES]).numpy(),
                m2=torch.cat([b[k].reshape(-1) for k in densify.NAMES]).numpy(), counts=np.array([nc, ns, npr, new.P]),
                accum=stats["xyz_gradient_accum"].numpy().copy(), denom=stats["denom"].numpy().copy(),
                radii=stats["max_radii2D"].numpy().copy())


def _named(opt):
    return [(g["name"], g["params"][0]) for g in opt.param_groups]


def _densify_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = _train_and_densify(world, rank, lambda s: [(s * world + rank) % 8])
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **out)
    dist.destroy_process_group()


def test_replicas_stay_bit_identical_across_a_densification_event_and_equal_sequential_accumulation(tmp_path):
    world = 2
    mp.spawn(_densify_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(world)]
    for k in r[0].files:
        assert np.array_equal(r[0][k], r[1][k]), f"replicas diverged in {k}"
    nc, ns, npr, newP = (int(v) for v in r[0]["counts"])
    assert nc > 0 and ns > 0 and npr > 0 and newP == P + nc + ns - npr, (nc, ns, npr, newP)
    # single process, same views accumulated sequentially, one Adam step per batch
    seq = _train_and_densify(1, 0, lambda s: [(s * world + k) % 8 for k in range(world)])
    assert np.array_equal(seq["denom"], r[0]["denom"]) and np.array_equal(seq["radii"], r[0]["radii"])
    np.testing.assert_allclose(seq["accum"], r[0]["accum"], rtol=2e-4, atol=1e-9)
    assert np.array_equal(seq["counts"], r[0]["counts"]), (seq["counts"], r[0]["counts"])
    np.testing.assert_allclose(seq["flat"], r[0]["flat"], rtol=2e-4, atol=2e-6)


def test_densify_and_prune_follows_the_reference_procedure():
    """clone: small + high gradient -> an exact copy appended (moments zero); split: large + high gradient -> replaced by two
    samples with scales / 1.6; prune: opacity below the threshold or larger than 0.1 extent; same seed -> same result"""
    g = torch.Generator().manual_seed(0)
    n = 200
    t = dict(xyz=torch.randn(n, 3, generator=g), opacity=torch.randn(n, 1, generator=g) * 3, rotation=torch.randn(n, 4, generator=g),
             scaling=torch.log(torch.rand(n, 3, generator=g) * 0.05 + 1e-3), features=torch.randn(n, 16, 3, generator=g))
    t["scaling"][:40] = torch.log(torch.tensor(0.2))         # large: candidates for a split (percent_dense * extent = 0.03)
    m1 = {k: torch.ones_like(v) for k, v in t.items()}
    m2 = {k: 2 * torch.ones_like(v) for k, v in t.items()}
    stats = dict(xyz_gradient_accum=torch.zeros(n), denom=torch.ones(n), max_radii2D=torch.zeros(n))
    stats["xyz_gradient_accum"][::2] = 1.0                   # every other Gaussian has a large gradient
    stats["denom"][5] = 0.0                                  # 0/0 -> nan -> 0 (gaussian_model.py:391-392)
    kw = dict(max_grad=0.5, min_opacity=0.005, extent=3.0, max_screen_size=None)
    out = [densify.densify_and_prune(t, m1, m2, stats, generator=torch.Generator().manual_seed(3), **kw) for _ in range(2)]
    for a, b in zip(out[0][:3], out[1][:3]):
        for k in a:
            assert torch.equal(a[k], b[k])
    nt, a1, a2, nc, ns, npr = out[0]
    big = (torch.exp(t["scaling"]).max(1).values > 0.03)
    hi = stats["xyz_gradient_accum"] / stats["denom"].clamp_min(1) >= 0.5
    assert nc == int((hi & ~big).sum()) and ns == int((hi & big).sum()) and ns > 0 and nc > 0
    survivors = n - ns
    P2 = nt["xyz"].shape[0]
    assert P2 == n + nc + 2 * ns - ns - npr
    if npr == 0:
        # order: survivors of the original (split ones removed), clones, 2 x split samples
        clones = nt["xyz"][survivors: survivors + nc]
        assert torch.equal(clones, t["xyz"][hi & ~big])
        assert float(a1["xyz"][survivors:].abs().max()) == 0.0 and float(a2["features"][survivors:].abs().max()) == 0.0
        assert torch.equal(a1["xyz"][:survivors], torch.ones(survivors, 3))
        new_sc = torch.exp(nt["scaling"][survivors + nc:])
        assert torch.allclose(new_sc, torch.exp(t["scaling"][hi & big]).repeat(2, 1) / 1.6)
    # transparent Gaussians go
    t2 = {k: v.clone() for k, v in t.items()}
    t2["opacity"][:] = 4.0
    t2["opacity"][100:110] = -9.0
    stats0 = dict(xyz_gradient_accum=torch.zeros(n), denom=torch.ones(n), max_radii2D=torch.zeros(n))
    nt2, *_rest, npr2 = densify.densify_and_prune(t2, m1, m2, stats0, **kw)
    assert npr2 == 10 and nt2["xyz"].shape[0] == n - 10


# ------------------------------------------------------------------ the reference's surface-bound SuGaR class, view-sharded
def _bound_model():
    from tests.golden import make_sugar_callsite as mk
    sm = mk._import_reference_model()
    from tests import oracle_rasterizer as orast
    torch.manual_seed(0)
    torch.set_num_threads(1)
    sm.knn_points = mk._scipy_knn_points
    sm.GaussianRasterizer = orast.GaussianRasterizer
    cams = syn.orbit_cameras(mk.W, mk.H)
    nerf = types.SimpleNamespace(device=torch.device("cpu"), training_cameras=mk._Cameras(cams))
    model = sm.SuGaR(nerfmodel=nerf, points=None, colors=None, initialize=False, sh_levels=4, keep_track_of_knn=False,
                     surface_mesh_to_bind=mk._bumpy_sphere(), n_gaussians_per_surface_triangle=6, learn_surface_mesh_positions=True,
                     learn_surface_mesh_opacity=True, learn_surface_mesh_scales=True)
    g = torch.Generator().manual_seed(78)
    n = model._n_points
    with torch.no_grad():
        model._scales += 0.3 * torch.randn(n, 2, generator=g) + 0.5
        model._quaternions += 0.7 * torch.randn(n, 2, generator=g)
        model.all_densities += 2.0 * torch.randn(n, 1, generator=g) + 2.5
    wimg = torch.randn(mk.H, mk.W, 3, generator=g)
    names = ("_points", "_scales", "_quaternions", "all_densities", "_sh_coordinates_dc", "_sh_coordinates_rest")
    return model, wimg, names


def _bound_steps(world, rank, views_of_step, steps=2):
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        model, wimg, names = _bound_model()
        # (the optimiser the refine trainer builds is a torch.optim.Adam over these tensors: sugar_optimizer.py:60-85)
        opt = torch.optim.Adam([{"params": [getattr(model, n)], "lr": 1e-3, "name": n} for n in names], eps=1e-15)
        vp.attach(opt)
        for s in range(steps):
            opt.zero_grad(set_to_none=True)
            views = views_of_step(s)
            for k in views:
                res = model.render_image_gaussian_rasterizer(camera_indices=k, bg_color=None, sh_deg=3, compute_color_in_rasterizer=True)
                ((res * wimg).mean() / (1 if world > 1 else len(views))).backward()
            opt.step()
        return {n: getattr(model, n).detach().numpy().copy() for n in names}
    finally:
        torch.Tensor.cuda = real_cuda


def _bound_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = _bound_steps(world, rank, lambda s: [(2 + 3 * (s * world + rank)) % 8])
    np.savez(os.path.join(out_dir, f"bound{rank}.npz"), **out)
    dist.destroy_process_group()


@pytest.mark.skipif(ref_env.reference_root() is None, reason="needs the reference tree")
def test_the_reference_bound_sugar_class_trains_view_sharded_with_the_trainer_untouched(tmp_path):
    world = 2
    mp.spawn(_bound_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"bound{k}.npz") for k in range(world)]
    for n in r[0].files:
        assert np.array_equal(r[0][n], r[1][n]), f"replicas diverged in {n}"
    seq = _bound_steps(1, 0, lambda s: [(2 + 3 * (s * world + k)) % 8 for k in range(world)])
    start = _bound_steps(1, 0, lambda s: [], steps=0)
    for n in r[0].files:
        assert np.abs(seq[n] - start[n]).max() > 0, f"{n} was not optimised"   # the mesh vertices `_points` included
        np.testing.assert_allclose(r[0][n], seq[n], rtol=2e-4, atol=2e-6, err_msg=n)


def test_gradient_exchange_counts_missing_gradients_as_zeros_and_buckets():
    """single process, no group: the hook is a no-op; the bucket walk itself is exercised through a fake one-rank group"""
    a, b = torch.nn.Parameter(torch.ones(5)), torch.nn.Parameter(torch.ones(3, 2))
    opt = torch.optim.Adam([a, b], lr=0.1)
    h = vp.attach(opt)
    a.grad = torch.ones(5)
    opt.step()                                  # b.grad is None, world 1: untouched
    assert b.grad is None and float(a.detach()[0]) < 1.0
    h.remove()
    with pytest.raises(TypeError):
        vp.attach(object())


def _none_grad_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = torch.nn.Parameter(torch.ones(5)), torch.nn.Parameter(torch.ones(3))
    opt = torch.optim.Adam([a, b], lr=0.1)
    vp.attach(opt)
    # step 1: b has a gradient on rank 0 only -> zeros on rank 1, both step it with the mean
    a.grad = torch.full((5,), float(rank + 1))
    b.grad = torch.ones(3) if rank == 0 else None
    opt.step()
    b1 = b.detach().clone()
    # step 2: b has no gradient on ANY rank -> it stays None and Adam skips b (no momentum step), as on one GPU
    a.grad = torch.ones(5); b.grad = None
    opt.step()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), a=a.detach().numpy(), b1=b1.numpy(), b2=b.detach().numpy(),
             b_grad_none=np.array(b.grad is None), b_steps=np.array(float(opt.state[b]["step"])))
    dist.destroy_process_group()


def test_a_gradient_missing_on_every_rank_stays_missing_and_the_parameter_is_not_stepped(tmp_path):
    world = 2
    mp.spawn(_none_grad_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(world)]
    for k in r[0].files:
        assert np.array_equal(r[0][k], r[1][k]), k
    assert bool(r[0]["b_grad_none"]) and float(r[0]["b_steps"]) == 1.0
    assert np.all(r[0]["b1"] < 1.0) and np.array_equal(r[0]["b1"], r[0]["b2"])   # moved in step 1, untouched in step 2


def test_the_fallback_of_fused_adam_runs_the_step_hooks_once():
    """a CPU parameter makes FusedAdam fall back to torch's own step; the pre-hook (where the gradient exchange lives) must fire
    once per step, for a constructed FusedAdam and for a torch.optim.Adam converted in place (`adopt`)"""
    from sugar_amd import fused_adam
    for make in (lambda ps: fused_adam.FusedAdam(ps, lr=0.1), lambda ps: fused_adam.adopt(torch.optim.Adam(ps, lr=0.1))):
        p = torch.nn.Parameter(torch.ones(4))
        opt = make([p])
        calls = []
        opt.register_step_pre_hook(lambda o, a, k: calls.append("pre"))
        opt.register_step_post_hook(lambda o, a, k: calls.append("post"))
        p.grad = torch.ones(4)
        opt.step()
        assert calls == ["pre", "post"], calls
        assert float(p.detach()[0]) < 1.0


def test_the_pieces_of_the_gradient_exchange_tile_their_ranges():
    """sugar_amd.train_step.exchange_pieces (the rule sgr_trainer_step_exchange applies in C++): monotone cut points on the
    alignments the kernels need (SH-Adam works on 64-Gaussian waves, flat Adam on float4s), covering [0, P) and [0, n_small)"""
    from sugar_amd.train_step import exchange_pieces
    for P in (1, 255, 256, 20_000, 1_000_000, 6_000_001):
        for pieces in (1, 2, 4, 7, 16):
            g, f = exchange_pieces(P, 11 * P, pieces)
            assert len(g) == len(f) == pieces + 1 and g[0] == f[0] == 0 and g[-1] == P and f[-1] == 11 * P
            assert all(a <= b for a, b in zip(g, g[1:])) and all(a <= b for a, b in zip(f, f[1:]))
            assert all(x % 64 == 0 for x in g[:-1]) and all(x % 4 == 0 for x in f[:-1])
