import os, sys, types, math
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from tests import ref_env
from tests.test_gpu_reference_sugar import _free_model, _rel, STATE, DEV, GOLD
sm = ref_env.import_sugar_model()
fx = np.load(os.path.join(GOLD, "sugar_callsite.npz"))
model, _ = _free_model(sm, fx, int(fx["W"]), int(fx["H"]))
wimg = torch.as_tensor(fx["dL_dimage_hw3"]).to(DEV)
import sugar_scene.sugar_model as M
calls = []
orig = M.GaussianRasterizer
class Rec:
    def __init__(self, raster_settings): self.inner = orig(raster_settings); self.s = raster_settings
    def __call__(self, **kw):
        for k, v in list(kw.items()):
            if torch.is_tensor(v) and v.requires_grad:
                kw[k] = v.view_as(v); kw[k].retain_grad()
        out = self.inner(**kw); calls.append((self.s, kw)); return out
M.GaussianRasterizer = Rec
for ci, (cam_idx, in_rast, bg) in enumerate(((1, False, None), (5, True, torch.tensor([1., 1., 1.], device=DEV)))):
    pre = f"c{ci}_"
    model.zero_grad(set_to_none=True)
    res = model.render_image_gaussian_rasterizer(camera_indices=cam_idx, bg_color=bg, sh_deg=3, compute_color_in_rasterizer=in_rast, return_2d_radii=True)
    (res["image"] * wimg).sum().backward()
    s, kw = calls[-1]
    print(pre, "image", _rel(res["image"], fx[pre + "image_hw3"]))
    for k in ("viewmatrix", "projmatrix", "campos"):
        print("   ", k, _rel(getattr(s, k), fx[pre + k]), getattr(s, k).flatten()[:4].tolist(), fx[pre+k].flatten()[:4].tolist())
    print("    tanfov", s.tanfovx, s.tanfovy, fx[pre + "tanfov"])
    for k, v in kw.items():
        if v is None: continue
        if pre + "in_" + k in fx.files: print("    in", k, _rel(v, fx[pre + "in_" + k]))
        if v.grad is not None and pre + "grad_" + k in fx.files: print("    grad", k, _rel(v.grad, fx[pre + "grad_" + k]))
    for name in STATE:
        print("    param", name, _rel(getattr(model, name).grad, fx[pre + "param_grad" + name]))
