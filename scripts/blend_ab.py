"""Developer A/B of blend-kernel variants (sgr_set_blend_variant) on the metric workload: HIP-event time of the blend
stages and the difference of the outputs against variant 0.
    python scripts/blend_ab.py [variants, e.g. 0 1 3] [--bwd] [--workload metric]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sugar_amd import _lib, synthetic as syn  # noqa: E402
from sugar_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _C  # noqa: E402

STAGES = ["preprocess", "bin_count_scan", "bin_scatter", "depth_sort", "blend_fwd", "blend_bwd", "preprocess_bwd"]


def main():
    args = [a for a in sys.argv[1:] if a.lstrip("-").isdigit()]
    variants = [int(a) for a in args] or [0]
    bwd = "--bwd" in sys.argv
    workload = "metric"
    if "--workload" in sys.argv:
        workload = sys.argv[sys.argv.index("--workload") + 1]
    lib = _lib.load()
    dev = torch.device("cuda:0")
    scene, cams, bg = syn.make_config(workload)
    W, H = cams[0].image_width, cams[0].image_height
    inputs = dict(means3D=scene.means3D.to(dev), opacities=scene.opacities.to(dev), shs=scene.shs.to(dev),
                  scales=scene.scales.to(dev), rotations=scene.rotations.to(dev))
    g = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
    ref = {}
    for v in variants:
        lib.sgr_set_blend_variant(v)
        outs = []
        for rep in range(3):
            if rep == 1:
                lib.sgr_profile_enable((1 << len(STAGES)) - 1)
            for ci, cam in enumerate(cams):
                st = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, bg.to(dev), 1.0, cam.viewmatrix.to(dev),
                                                   cam.projmatrix.to(dev), 3, cam.campos.to(dev), False, False)
                leaves = {k: t.clone().requires_grad_(bwd) for k, t in inputs.items()}
                m2 = torch.zeros_like(leaves["means3D"], requires_grad=bwd)
                with torch.set_grad_enabled(bwd):
                    color, radii = GaussianRasterizer(st)(means2D=m2, **leaves)
                if bwd:
                    color.backward(g)
                if rep == 0 and ci < 2:
                    lf = _C.last_forward
                    o = lib.sgr_img_final_T_offset(W, H)
                    fT = lf["img"][o:o + 4 * W * H].view(torch.float32).clone()
                    o = lib.sgr_img_n_contrib_offset(W, H)
                    nc = lf["img"][o:o + 4 * W * H].view(torch.int32).clone()
                    T = ((W + 15) // 16) * ((H + 15) // 16)
                    o = lib.sgr_img_tile_maxc_offset(W, H)
                    mc = lf["img"][o:o + 4 * T].view(torch.int32).clone()
                    o = lib.sgr_img_tile_walked_offset(W, H)
                    wk = lf["img"][o:o + 4 * T].view(torch.int32).clone()
                    rec = dict(color=color.detach().clone(), final_T=fT, n_contrib=nc, tile_maxc=mc, tile_walked=wk)
                    if bwd:
                        rec.update({"d_" + k: t.grad.clone() for k, t in leaves.items()})
                    outs.append(rec)
        torch.cuda.synchronize()
        lib.sgr_profile_enable(0)
        ms = (C.c_double * len(STAGES))()
        cnt = (C.c_int64 * len(STAGES))()
        lib.sgr_profile_read(ms, cnt, len(STAGES))
        line = f"variant {v}: " + "  ".join(f"{n}={1e3 * ms[i] / max(cnt[i], 1):.1f}us" for i, n in enumerate(STAGES) if cnt[i] and "blend" in n)
        if not ref:
            ref = outs
        else:
            for ci, (a, b) in enumerate(zip(ref, outs)):
                parts = []
                for k in a:
                    if a[k].dtype in (torch.int32,):
                        parts.append(f"{k}:neq={(a[k] != b[k]).sum().item()}")
                    else:
                        d = (a[k] - b[k]).abs().max().item()
                        nrm = ((a[k] - b[k]).norm() / a[k].norm().clamp_min(1e-30)).item()
                        parts.append(f"{k}:max={d:.2e},rel={nrm:.1e}")
                line += f"\n    cam{ci} vs variant {variants[0]}: " + " ".join(parts)
        print(line, flush=True)
    lib.sgr_set_blend_variant(0)


if __name__ == "__main__":
    main()
