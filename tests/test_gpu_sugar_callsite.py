"""GPU replay of the reference's real call site: tests/golden/sugar_callsite.npz holds the tensors the UNMODIFIED
SuGaR.render_image_gaussian_rasterizer (sugar_scene/sugar_model.py:2085-2294) handed to the GaussianRasterizer boundary
(once with colours from get_points_rgb as `colors_precomp`, once with SH evaluated in the rasterizer) together with the image
and gradients the CPU oracle returned.  The HIP rasterizer must reproduce them from the same boundary inputs.
sugar_callsite_bound.npz is the same for the refine-mode model (BASELINE.json config 4: six flat Gaussians per triangle of a
surface mesh, thickness 3e-6 -- covariances of rank 2 up to rounding, the hardest conditioning this boundary sees)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDS = {name: np.load(os.path.join(os.path.dirname(__file__), "golden", f"sugar_callsite{suffix}.npz"))
         for name, suffix in (("free", ""), ("bound", "_bound"))}
DEV = "cuda:0"


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize("call", [0, 1])
@pytest.mark.parametrize("model", ["free", "bound"])
def test_replay_of_the_sugar_call_site(model, call):
    GOLD = GOLDS[model]
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer  # the drop-in package name
    pre = f"c{call}_"
    t = lambda k: torch.tensor(GOLD[pre + k], device=DEV)
    settings = GaussianRasterizationSettings(
        image_height=int(GOLD["H"]), image_width=int(GOLD["W"]), tanfovx=float(GOLD[pre + "tanfov"][0]),
        tanfovy=float(GOLD[pre + "tanfov"][1]), bg=t("bg"), scale_modifier=1.0, viewmatrix=t("viewmatrix"),
        projmatrix=t("projmatrix"), sh_degree=int(GOLD[pre + "sh_degree"]), campos=t("campos"), prefiltered=False, debug=False)
    names = [k[len(pre) + 3:] for k in GOLD.files if k.startswith(pre + "in_")]
    inputs = {}
    for n in ("means3D", "means2D", "shs", "colors_precomp", "opacities", "scales", "rotations", "cov3D_precomp"):
        inputs[n] = t("in_" + n).requires_grad_(True) if n in names else None
    assert (inputs["shs"] is None) != (inputs["colors_precomp"] is None)
    image, radii = GaussianRasterizer(raster_settings=settings)(**inputs)
    img_hw3 = image.transpose(0, 1).transpose(1, 2)  # as the caller returns it (:2283)
    (img_hw3 * torch.tensor(GOLD["dL_dimage_hw3"], device=DEV)).sum().backward()
    assert np.array_equal(radii.cpu().numpy(), GOLD[pre + "radii"])
    assert _rel(img_hw3.detach().cpu().numpy(), GOLD[pre + "image_hw3"]) < 1e-5
    checked = 0
    for n, v in inputs.items():
        key = pre + "grad_" + n
        if v is None or key not in GOLD.files:
            continue
        assert _rel(v.grad.cpu().numpy(), GOLD[key]) < 1e-4, n
        checked += 1
    assert checked >= 5
