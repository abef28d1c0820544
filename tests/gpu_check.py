"""Developer check on the GPU box: prints parity statistics HIP vs CPU oracle and rough timings.
    python tests/gpu_check.py [--big]
"""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sugar_amd import synthetic as syn, build
from oracle import cpu_oracle as orc
from tests import parity_utils as pu

build.build()

def check(name, scene, cam, bg, **opts):
    st = pu.run_oracle(scene, cam, bg, **opts)
    H, W = cam.image_height, cam.image_width
    g = np.random.default_rng(0).standard_normal((3, H, W)).astype(np.float32)
    hp = pu.run_hip(scene, cam, bg, grad_out=g, **opts)
    gr = orc.backward(st, g)
    print(f"== {name}: P={scene.means3D.shape[0]} {W}x{H} R oracle={st['num_rendered']} hip={hp['num_rendered']}")
    print("  radii equal:", np.array_equal(st["radii"], hp["radii"]), " mismatches:", int((st["radii"] != hp["radii"]).sum()))
    vis = st["radii"] > 0
    rec = hp["rec"]
    print("  means2D bits equal:", np.array_equal(rec[vis, 0:2].view(np.uint32), st["means2D"][vis].view(np.uint32)),
          " conic bits:", np.array_equal(rec[vis][:, [2, 3, 4]].view(np.uint32), st["conic_opacity"][vis][:, :3].view(np.uint32)),
          " depth bits:", np.array_equal(rec[vis, pu.REC_DEPTH].view(np.uint32), st["depths"][vis].view(np.uint32)))
    if "shs" in pu.scene_kwargs(scene, cam, bg, **opts):
        print("  rgb:", pu.rel_stats(rec[vis, pu.REC_RGB], st["rgb"][vis]))
    lens_o = st["ranges"][:, 1] - st["ranges"][:, 0]
    lens_h = np.diff(hp["tile_start"])
    print("  tile lens equal:", np.array_equal(lens_o, lens_h), " point_list equal:", np.array_equal(st["point_list"], hp["point_list"]))
    print("  n_contrib mismatches:", int((st["n_contrib"] != hp["n_contrib"]).sum()), "of", W * H)
    print("  final_T:", pu.rel_stats(hp["final_T"], st["final_T"]))
    print("  color  :", pu.rel_stats(hp["color"], st["color"]))
    names = dict(means3D="dL_dmeans3D", means2D="dL_dmeans2D", opacities="dL_dopacity", shs="dL_dsh", colors_precomp="dL_dcolors",
                 scales="dL_dscales", rotations="dL_drotations", cov3D_precomp="dL_dcov3D")
    for k, v in hp["grads"].items():
        ref = gr[names[k]]
        print(f"  grad {k:14s}", pu.rel_stats(v.reshape(ref.shape), ref))

torch.manual_seed(0)
scene, cams, bg = syn.make_config("config1")
check("config1/sh", scene, cams[0], bg)
check("config1/colors_precomp+white", scene, cams[3], torch.ones(3), use_sh=False)
check("config1/cov_precomp deg1 mod1.3", scene, cams[5], torch.tensor([0.2, 0.5, 0.7]), use_cov=True, sh_degree=1, scale_modifier=1.3)
small = syn.make_scene(3000, 11, 0.01, 0.3)
check("partial tiles 250x190", small, syn.orbit_cameras(250, 190)[2], torch.tensor([0.1, 0.2, 0.3]))
inside = syn.make_scene(2000, 12, 0.05, 0.5)
check("camera inside cloud (near culls)", inside, syn.look_at_camera((0.1, 0.0, 0.0), (1.0, 0.2, 0.0), 200, 120), torch.zeros(3))

if "--big" in sys.argv:
    from sugar_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device("cuda:0")
    for name in ("config2", "metric"):
        scene, cams, bg = syn.make_config(name)
        cam = cams[0]
        settings = GaussianRasterizationSettings(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy, bg.to(dev), 1.0,
                                                 cam.viewmatrix.to(dev), cam.projmatrix.to(dev), 3, cam.campos.to(dev), False, False)
        rast = GaussianRasterizer(settings)
        m = scene.means3D.to(dev).requires_grad_(True); m2 = torch.zeros_like(m, requires_grad=True)
        op = scene.opacities.to(dev).requires_grad_(True); sh = scene.shs.to(dev).requires_grad_(True)
        sc = scene.scales.to(dev).requires_grad_(True); ro = scene.rotations.to(dev).requires_grad_(True)
        g = torch.randn(3, cam.image_height, cam.image_width, device=dev)
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.time()
            color, radii = rast(m, m2, op, shs=sh, scales=sc, rotations=ro)
            torch.cuda.synchronize(); t1 = time.time()
            color.backward(g)
            torch.cuda.synchronize(); t2 = time.time()
        R = color.grad_fn.num_rendered if color.grad_fn is not None else -1
        print(f"== timing {name}: fwd {1e3*(t1-t0):.2f} ms  bwd {1e3*(t2-t1):.2f} ms  visible {(radii>0).sum().item()}")
