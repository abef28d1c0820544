"""The reference's coarse trainer itself (sugar_trainers/coarse_sdf.py:17-825, untouched) on the HIP drop-ins: it loads a 3DGS
checkpoint and ground-truth images from disk (through the `plyfile` stand-in and the reference's own camera loader), builds the
reference's SuGaR model, optimiser and densifier, and optimises through entropy regularisation, the opacity pruning at 9 000 and
the first SDF / normal regularisation iterations (1M density samples per iteration, depth renders, neighbour resets).  The run
is cut at iteration 9 060 by a counter around the module-level `ssim` -- see oracle/reference_trainer.py.  What is asserted is
what a user switching rasterizers would look at: every logged loss finite, the photometric phase converging, the regularisers
announced when the schedule says, Gaussians pruned, no exception anywhere on the way."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_the_unmodified_coarse_trainer_runs_through_its_schedule_on_the_drop_ins(tmp_path):
    from tests import ref_env
    if ref_env.reference_root() is None:
        pytest.skip("the reference's Python is not staged (oracle/ref_build/build_ref.sh)")
    from oracle import reference_trainer as rt
    data = rt.write_dataset(str(tmp_path / "data"), P=30_000, n_cams=24, W=320, H=208)
    res = rt.run(data, str(tmp_path / "out"), stop_at=9060, patch_sugar=True)
    assert not res["finished"] and res["last_iteration"] == 9060
    its = [i for i, _ in res["losses"]]
    assert its[0] == 7000 and its[-1] == 9050 and len(its) == 42          # one line per 50 iterations (coarse_sdf.py:221,761)
    assert all(math.isfinite(v) for _, v in res["losses"])
    by_it = dict(res["losses"])
    # photometric + entropy phase: the perturbed checkpoint is pulled back onto the ground-truth views
    early = sum(by_it[i] for i in (7000, 7050, 7100)) / 3
    late = sum(by_it[i] for i in (8850, 8900, 8950)) / 3
    assert late < 0.8 * early, (early, late)
    for e in ("Starting entropy regularization", "Stopping entropy regularization", "Pruning gaussians with low-opacity",
              "Starting SDF regularization", "Starting SDF estimation loss", "Starting SDF better normal loss", "Resetting neighbors"):
        assert e in res["events"], e
    assert 0 < res["gaussians_after_pruning"] <= 30_000
    assert torch.cuda.is_available()
