"""GPU parity of the fused SH -> RGB op (SuGaR.get_points_rgb, sugar_scene/sugar_model.py:839-883) against golden vectors
made with the reference's own eval_sh (tests/golden/make_golden.py: forward colours and autograd gradients for
sh_levels 1..4) and, for the `directions` mode, against the float64 autograd restatement in oracle/."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_helpers.npz"))
DEV = "cuda:0"


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize("sh_levels", [1, 2, 3, 4])
def test_get_points_rgb_matches_reference_golden(sh_levels):
    from sugar_amd.shcolor import get_points_rgb

    class Model:  # the two attributes SuGaR.get_points_rgb reads from `self`
        pass
    m = Model()
    m.points = torch.tensor(GOLD["prgb_positions"], device=DEV, requires_grad=True)
    m.sh_coordinates = torch.tensor(GOLD["prgb_sh_coordinates"], device=DEV, requires_grad=True)
    cam = torch.tensor(GOLD["prgb_camera_center"], device=DEV)
    w = torch.tensor(GOLD["prgb_weights"], device=DEV)
    colors = get_points_rgb(m, camera_centers=cam, sh_levels=sh_levels)
    assert colors.shape == (m.points.shape[0], 3)
    (colors * w).sum().backward()
    np.testing.assert_allclose(colors.detach().cpu().numpy(), GOLD[f"prgb_colors_l{sh_levels}"], rtol=2e-5, atol=2e-6)
    assert m.sh_coordinates.grad.shape == (m.points.shape[0], 16, 3)
    assert _rel(m.sh_coordinates.grad.cpu().numpy(), GOLD[f"prgb_dsh_l{sh_levels}"]) < 1e-5
    if sh_levels > 1:
        assert _rel(m.points.grad.cpu().numpy(), GOLD[f"prgb_dpos_l{sh_levels}"]) < 1e-4
    else:
        assert float(m.points.grad.abs().max()) == 0.0
    # per-point camera centres and a pre-sliced coefficient tensor give the same colours
    c2 = get_points_rgb(m, positions=m.points.detach(), camera_centers=cam.expand(m.points.shape[0], 3),
                        sh_levels=sh_levels, sh_coordinates=m.sh_coordinates.detach()[:, :sh_levels ** 2].contiguous())
    assert torch.equal(c2, colors.detach())


def test_directions_mode_and_errors():
    from oracle.torch_cpu_rasterizer import eval_sh_color
    from sugar_amd.shcolor import sh_to_rgb, get_points_rgb
    g = torch.Generator().manual_seed(3)
    P = 5000
    sh = torch.randn(P, 16, 3, generator=g, dtype=torch.float64) * 0.5
    d = torch.nn.functional.normalize(torch.randn(P, 3, generator=g, dtype=torch.float64), dim=-1)
    w = torch.randn(P, 3, generator=g, dtype=torch.float64)
    shr = sh.clone().requires_grad_(True); dr = d.clone().requires_grad_(True)
    ref = eval_sh_color(3, shr, dr)  # includes the +0.5 and the clamp
    (ref * w).sum().backward()
    shd = sh.float().to(DEV).requires_grad_(True); dd = d.float().to(DEV).requires_grad_(True)
    out = sh_to_rgb(shd, 4, directions=dd)
    (out * w.float().to(DEV)).sum().backward()
    assert _rel(out.detach().cpu().numpy(), ref.detach().numpy()) < 1e-5
    assert _rel(shd.grad.cpu().numpy(), shr.grad.numpy()) < 1e-5
    assert _rel(dd.grad.cpu().numpy(), dr.grad.numpy()) < 1e-4
    with pytest.raises(ValueError):
        get_points_rgb(type("M", (), {"points": dd, "sh_coordinates": shd})(), sh_levels=4)
    with pytest.raises(RuntimeError):
        sh_to_rgb(sh.float(), 4, directions=d.float())  # CPU tensors: no fallback
