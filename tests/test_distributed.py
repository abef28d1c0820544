"""World-size-2 rehearsal of the view-sharded data-parallel train step over gloo (CPU).  The rasterizer underneath is
the oracle-backed stand-in (tests/oracle_rasterizer.py); what is under test is the host logic of
sugar_amd/train_step.py: one view per rank, one flat all-reduce, identical Adam steps, replicas stay identical and
equal to single-process accumulation of the same views."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sugar_amd import synthetic as syn
from sugar_amd.train_step import GaussianParams, ViewShardedTrainer, render, photometric_loss

P, W, H, STEPS = 300, 48, 32, 2


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _setup():
    from tests import oracle_rasterizer as orast
    torch.set_num_threads(1)
    scene = syn.make_scene(P, 17, 0.03, 0.2)
    cams = syn.orbit_cameras(W, H)
    g = torch.Generator().manual_seed(5)
    gts = [torch.rand(3, H, W, generator=g) for _ in cams]
    return orast, scene, cams, gts


def _worker(rank, world, port, out_dir, compact):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orast, scene, cams, gts = _setup()
    params = GaussianParams(scene, torch.device("cpu"))
    kw = dict(compact_sh=True, sh_grad_fn=orast.sh_grad_from_views_torch, grad_sink_cm=orast.grad_sink) if compact else {}
    tr = ViewShardedTrainer(params, orast.GaussianRasterizer, orast.GaussianRasterizationSettings, torch.zeros(3), **kw)
    for s in range(STEPS):
        k = (s * world + rank) % len(cams)
        tr.step(cams[k], gts[k])
    np.save(os.path.join(out_dir, f"flat_{rank}.npy"), params.flat.detach().numpy())
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("compact", [False, True], ids=["flat_allreduce", "compact_sh_exchange"])
def test_view_sharded_step_matches_sequential_accumulation(tmp_path, compact):
    """Both gradient exchanges (one flat all-reduce of 59 floats/Gaussian; all-gather of 3 masked colour gradients per view
    + all-reduce of the other 11 + local SH reconstruction) must equal single-process accumulation of the same views."""
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), compact), nprocs=world, join=True)
    flats = [np.load(tmp_path / f"flat_{r}.npy") for r in range(world)]
    assert np.array_equal(flats[0], flats[1]), "replicas diverged"
    # single-process reference: accumulate the same views, mean gradient, one Adam step per batch
    orast, scene, cams, gts = _setup()
    params = GaussianParams(scene, torch.device("cpu"))
    opt = params.make_optimizer()
    for s in range(STEPS):
        params.flat_grad.zero_()
        for r in range(world):
            k = (s * world + r) % len(cams)
            pkg = render(params, cams[k], torch.zeros(3), orast.GaussianRasterizer, orast.GaussianRasterizationSettings)
            (photometric_loss(pkg["render"], gts[k]) / world).backward()
        opt.step()
    ref = params.flat.detach().numpy()
    assert not np.array_equal(ref, GaussianParams(scene, torch.device("cpu")).flat.numpy()), "nothing was optimised"
    np.testing.assert_allclose(flats[0], ref, rtol=2e-4, atol=2e-6)


def test_flat_parameter_views_and_grads_alias():
    scene = syn.make_scene(10, 1, 0.01, 0.1)
    p = GaussianParams(scene, torch.device("cpu"))
    assert sum(v.numel() for v in p.params.values()) == 10 * 59 and p.n_small == p.offsets["features"]
    assert p.params["features"].shape == (10, 16, 3)
    p.params["xyz"].grad.fill_(3.0)
    assert float(p.flat_grad.sum()) == 3.0 * 30
    a = p.activated()
    torch.testing.assert_close(a["scales"], scene.scales, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(a["opacities"], scene.opacities, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(a["shs"], scene.shs)
