import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
GOLD = np.load("/root/repo/tests/golden/sugar_callsite.npz")
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
DEV = "cuda:0"
def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64); return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
def run(tag, vm_noncontig, pm_noncontig=False):
    pre = "c0_"
    t = lambda k: torch.tensor(GOLD[pre + k], device=DEV)
    vm = t("viewmatrix"); pm = t("projmatrix")
    if vm_noncontig:
        vm = torch.tensor(GOLD[pre + "viewmatrix"].T.copy()).transpose(0, 1).to(DEV)
        assert not vm.is_contiguous() and torch.equal(vm.cpu(), torch.tensor(GOLD[pre + "viewmatrix"]))
    if pm_noncontig:
        pm = torch.tensor(GOLD[pre + "projmatrix"].T.copy()).transpose(0, 1).to(DEV)
    st = GaussianRasterizationSettings(image_height=int(GOLD["H"]), image_width=int(GOLD["W"]), tanfovx=float(GOLD[pre + "tanfov"][0]),
        tanfovy=float(GOLD[pre + "tanfov"][1]), bg=t("bg"), scale_modifier=1.0, viewmatrix=vm, projmatrix=pm, sh_degree=3, campos=t("campos"), prefiltered=False, debug=False)
    inputs = {}
    names = [k[len(pre) + 3:] for k in GOLD.files if k.startswith(pre + "in_")]
    for n in ("means3D", "means2D", "shs", "colors_precomp", "opacities", "scales", "rotations", "cov3D_precomp"):
        inputs[n] = t("in_" + n).requires_grad_(True) if n in names else None
    image, radii = GaussianRasterizer(raster_settings=st)(**inputs)
    img = image.transpose(0, 1).transpose(1, 2)
    (img * torch.tensor(GOLD["dL_dimage_hw3"], device=DEV)).sum().backward()
    print(tag, "image", rel(img.detach().cpu().numpy(), GOLD[pre + "image_hw3"]), {n: round(rel(v.grad.cpu().numpy(), GOLD[pre + "grad_" + n]), 6) for n, v in inputs.items() if v is not None})
run("contig      ", False)
run("vm noncontig", True)
run("pm noncontig", False, True)
run("contig again", False)
