"""Generate golden vectors by IMPORTING the reference's own Python helpers (run in the build container,
where /root/reference exists; the resulting .npz files are committed and travel to the GPU box).

    python tests/golden/make_golden.py

What is pinned (SURVEY.md section 8c "in-repo PyTorch restatements usable as cross-checks"):
  * SH -> RGB:       sugar_utils/spherical_harmonics.py:117-172 (eval_sh) + clamp_min(.+0.5, 0) as in
                     gaussian_splatting/gaussian_renderer/__init__.py:73-78   == DGR forward.cu:20-71
  * cov3D:           gaussian_splatting/utils/general_utils.py:78-110 (build_rotation / build_scaling_rotation,
                     strip_symmetric) as used by scene/gaussian_model.py:27-31 == DGR forward.cu:118-152
  * projection:      sugar_utils/graphics_utils.py:65-85 (getProjectionMatrix), :51-63 (getWorld2View2)
The reference hard-codes device="cuda" in general_utils; torch.zeros is wrapped to force CPU while its
functions run (the reference files themselves are not modified).
"""
import importlib.util
import math
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    sh_mod = _load("ref_sh", f"{REF}/sugar_utils/spherical_harmonics.py")
    gu_mod = _load("ref_gu", f"{REF}/gaussian_splatting/utils/general_utils.py")
    gr_mod = _load("ref_gr", f"{REF}/sugar_utils/graphics_utils.py")

    g = torch.Generator().manual_seed(1234)
    P = 96
    means = torch.rand(P, 3, generator=g) * 2 - 1
    campos = torch.tensor([2.5, -1.0, 0.7])
    shs = torch.randn(P, 16, 3, generator=g) * 0.4  # [P, M, 3] rasterizer layout
    dirs = means - campos[None]
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    out = {"means": means.numpy(), "campos": campos.numpy(), "shs": shs.numpy()}
    for deg in range(4):
        shs_view = shs.transpose(1, 2)  # [P, 3, M] as eval_sh expects
        rgb = torch.clamp_min(sh_mod.eval_sh(deg, shs_view, dirs) + 0.5, 0.0)
        out[f"rgb_deg{deg}"] = rgb.numpy()

    scales = torch.exp(torch.rand(P, 3, generator=g) * 3 - 4)
    rots = torch.randn(P, 4, generator=g)
    rots = rots / rots.norm(dim=1, keepdim=True)
    orig_zeros = torch.zeros

    def cpu_zeros(*a, **k):
        k["device"] = "cpu"
        return orig_zeros(*a, **k)

    torch.zeros = cpu_zeros
    try:
        for mod_name, mod in (("1", 1.0), ("1p7", 1.7)):
            L = gu_mod.build_scaling_rotation(mod * scales, rots)
            cov = L @ L.transpose(1, 2)
            out[f"cov3D_mod{mod_name}"] = gu_mod.strip_symmetric(cov).numpy()
    finally:
        torch.zeros = orig_zeros
    out["scales"] = scales.numpy()
    out["rots"] = rots.numpy()

    fovx, fovy = math.radians(60.0), math.radians(41.0)
    out["proj_znear0p01_zfar100_fovx60_fovy41"] = gr_mod.getProjectionMatrix(0.01, 100.0, fovx, fovy).numpy()
    R = np.array([[0.36, 0.48, -0.8], [-0.8, 0.6, 0.0], [0.48, 0.64, 0.6]])
    t = np.array([0.1, -0.2, 3.0])
    out["w2v_R"] = R
    out["w2v_t"] = t
    out["w2v"] = gr_mod.getWorld2View2(R, t)
    # photometric loss: sugar_utils/loss_utils.py (l1_loss, ssim) combined as gaussian_splatting/train.py:88-90
    lu = _load("ref_lu", f"{REF}/sugar_utils/loss_utils.py")
    img = torch.rand(3, 37, 45, generator=g, requires_grad=True)
    gt = (img.detach() + 0.25 * torch.randn(3, 37, 45, generator=g)).clamp(0, 1)
    gt[:, :5, :7] = img.detach()[:, :5, :7]  # an exactly-equal patch: sign(0) = 0 in the L1 gradient
    lam = 0.2
    loss = (1.0 - lam) * lu.l1_loss(img, gt) + lam * (1.0 - lu.ssim(img, gt))
    loss.backward()
    out["loss_img"] = img.detach().numpy(); out["loss_gt"] = gt.numpy()
    out["loss_value"] = np.float32(loss.item()); out["loss_grad"] = img.grad.numpy()
    out["loss_l1"] = np.float32(lu.l1_loss(img, gt).item()); out["loss_ssim"] = np.float32(lu.ssim(img, gt).item())
    # SuGaR.get_points_rgb (sugar_scene/sugar_model.py:862-881, restated: the class cannot be imported) on the reference's own
    # eval_sh, forward and autograd gradients, for sh_levels 1..4 (own generator: the vectors above stay unchanged)
    g2 = torch.Generator().manual_seed(4321)
    Pc = 160
    pos = (torch.rand(Pc, 3, generator=g2) * 2 - 1)
    cam = torch.tensor([[1.7, 0.4, -2.2]])
    shc = torch.randn(Pc, 16, 3, generator=g2) * 0.5
    w = torch.randn(Pc, 3, generator=g2)
    out["prgb_positions"] = pos.numpy(); out["prgb_camera_center"] = cam.numpy()
    out["prgb_sh_coordinates"] = shc.numpy(); out["prgb_weights"] = w.numpy()
    for sh_levels in (1, 2, 3, 4):
        p_ = pos.clone().requires_grad_(True); s_ = shc.clone().requires_grad_(True)
        render_directions = torch.nn.functional.normalize(p_ - cam, dim=-1)
        sh_coordinates = s_[:, :sh_levels ** 2]
        shs_view = sh_coordinates.transpose(-1, -2).view(-1, 3, sh_levels ** 2)
        sh2rgb = sh_mod.eval_sh(sh_levels - 1, shs_view, render_directions)
        colors = torch.clamp_min(sh2rgb + 0.5, 0.0).view(-1, 3)
        (colors * w).sum().backward()
        out[f"prgb_colors_l{sh_levels}"] = colors.detach().numpy()
        out[f"prgb_dsh_l{sh_levels}"] = s_.grad.numpy(); out[f"prgb_dpos_l{sh_levels}"] = (p_.grad if p_.grad is not None else torch.zeros_like(pos)).numpy()
    np.savez(os.path.join(HERE, "reference_helpers.npz"), **out)
    print("wrote", os.path.join(HERE, "reference_helpers.npz"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    sys.exit(main())
