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


def test_camera_matrices_with_transposed_strides_give_the_same_gradients():
    """SuGaR hands over `torch.Tensor(...).transpose(0, 1).cuda()` (sugar_model.py:2143-2150): a view matrix whose strides are
    those of its transpose.  Found in round 4 by running the reference class itself on the GPU: the backward took the pointer of
    a temporary `.contiguous()` copy whose block had already gone back to the caching allocator -- the covariance gradients
    (scales, rotations, part of the positions) were wrong by up to 5x while image, colour and opacity gradients were right."""
    GOLD = GOLDS["free"]
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    pre = "c0_"
    t = lambda k: torch.tensor(GOLD[pre + k], device=DEV)
    noncontig = lambda k: torch.tensor(GOLD[pre + k].T.copy()).transpose(0, 1).to(DEV)
    grads = []
    for vm, pm in ((t("viewmatrix"), t("projmatrix")), (noncontig("viewmatrix"), noncontig("projmatrix")),
                   (t("viewmatrix"), noncontig("projmatrix")), (noncontig("viewmatrix"), t("projmatrix"))):
        settings = GaussianRasterizationSettings(
            image_height=int(GOLD["H"]), image_width=int(GOLD["W"]), tanfovx=float(GOLD[pre + "tanfov"][0]),
            tanfovy=float(GOLD[pre + "tanfov"][1]), bg=t("bg"), scale_modifier=1.0, viewmatrix=vm, projmatrix=pm,
            sh_degree=int(GOLD[pre + "sh_degree"]), campos=t("campos"), prefiltered=False, debug=False)
        inputs = {n: (t("in_" + n).requires_grad_(True) if pre + "in_" + n in GOLD.files else None)
                  for n in ("means3D", "means2D", "shs", "colors_precomp", "opacities", "scales", "rotations", "cov3D_precomp")}
        image, _ = GaussianRasterizer(raster_settings=settings)(**inputs)
        (image.transpose(0, 1).transpose(1, 2) * torch.tensor(GOLD["dL_dimage_hw3"], device=DEV)).sum().backward()
        # more work on the same stream right away, as a training loop would do: reuses whatever the call released
        junk = [torch.randn(16, device=DEV) for _ in range(64)]
        grads.append({n: v.grad.cpu().numpy() for n, v in inputs.items() if v is not None})
        del junk
    for g in grads[1:]:
        for n in grads[0]:
            assert _rel(g[n], grads[0][n]) < 2e-5, n
            assert _rel(g[n], GOLD[pre + "grad_" + n]) < 1e-4, n
