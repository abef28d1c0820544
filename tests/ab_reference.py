"""A/B on the GPU box: the REFERENCE rasterizer (oracle/_ref, hipcc build of the unmodified CUDA sources) versus this
repository's HIP rasterizer, same inputs, same MI355X.  Prints parity statistics (reference vs CPU oracle, reference vs
product) and fwd / bwd timings.   python tests/ab_reference.py [config ...]
"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sugar_amd import synthetic as syn, build
from oracle import ref_gpu, cpu_oracle as orc
from tests import parity_utils as pu

build.build()
dev = torch.device("cuda:0")


def dev_inputs(scene, cam, bg):
    return dict(viewmatrix=cam.viewmatrix.to(dev), projmatrix=cam.projmatrix.to(dev), campos=cam.campos.to(dev), bg=bg.to(dev),
                W=cam.image_width, H=cam.image_height, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy)


def parity(name, scene, cam, bg):
    H, W = cam.image_height, cam.image_width
    g = np.random.default_rng(0).standard_normal((3, H, W)).astype(np.float32)
    st = ref_gpu.forward(scene.means3D.to(dev), scene.opacities.to(dev), shs=scene.shs.to(dev), scales=scene.scales.to(dev),
                         rotations=scene.rotations.to(dev), **dev_inputs(scene, cam, bg))
    rg = {k: v.cpu().numpy() for k, v in ref_gpu.backward(st, torch.as_tensor(g).to(dev)).items()}
    rd = ref_gpu.decode(st)
    co = pu.run_oracle(scene, cam, bg)
    cg = orc.backward(co, g)
    hp = pu.run_hip(scene, cam, bg, grad_out=g)
    P = scene.means3D.shape[0]
    vis = (co["radii"] > 0) & (rd["radii"] > 0)
    print(f"== {name}: P={P} {W}x{H} R ref={rd['num_rendered']} oracle={co['num_rendered']} hip={hp['num_rendered']}")
    print("  [ref vs oracle] radii mismatches:", int((rd["radii"] != co["radii"]).sum()),
          " tiles_touched mismatches:", int((rd["tiles_touched"] != co["tiles_touched"]).sum()),
          " depth-bit mismatches:", int((rd["depths"][vis].view(np.uint32) != co["depths"][vis].view(np.uint32)).sum()),
          " means2D-bit mismatches:", int((rd["means2D"][vis].view(np.uint32) != co["means2D"][vis].view(np.uint32)).any(axis=1).sum()),
          " conic-bit mismatches:", int((rd["conic_opacity"][vis][:, :3].view(np.uint32) != co["conic_opacity"][vis][:, :3].view(np.uint32)).any(axis=1).sum()))
    same_len = np.array_equal(rd["ranges"][:, 1] - rd["ranges"][:, 0], co["ranges"][:, 1] - co["ranges"][:, 0])
    pl_eq = rd["num_rendered"] == co["num_rendered"] and np.array_equal(rd["point_list"], co["point_list"])
    pl_mis = int((rd["point_list"] != co["point_list"]).sum()) if rd["num_rendered"] == co["num_rendered"] else -1
    print("  [ref vs oracle] tile lens equal:", same_len, " point_list equal:", pl_eq, " differing slots:", pl_mis)
    print("  [ref vs oracle] conic rel:", pu.rel_stats(rd["conic_opacity"][vis][:, :3], co["conic_opacity"][vis][:, :3]))
    print("  [ref vs oracle] n_contrib mismatches:", int((rd["n_contrib"] != co["n_contrib"]).sum()), " color:", pu.rel_stats(rd["color"], co["color"]))
    print("  [hip vs ref   ] n_contrib mismatches:", int((rd["n_contrib"] != hp["n_contrib"]).sum()), " color:", pu.rel_stats(hp["color"], rd["color"]))
    names = dict(means3D="dL_dmeans3D", means2D="dL_dmeans2D", opacities="dL_dopacity", shs="dL_dsh", scales="dL_dscales", rotations="dL_drotations")
    for k, n in names.items():
        print(f"  grad {k:10s} ref-vs-oracle norm_rel {pu.rel_stats(rg[n], cg[n])['norm_rel']:.2e}   hip-vs-ref norm_rel {pu.rel_stats(hp['grads'][k].reshape(rg[n].shape), rg[n])['norm_rel']:.2e}")


def timing(name, reps=5):
    from sugar_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    scene, cams, bg = syn.make_config(name)
    cam = cams[0]
    H, W = cam.image_height, cam.image_width
    d = dev_inputs(scene, cam, bg)
    m, op, sh, sc, ro = (t.to(dev) for t in (scene.means3D, scene.opacities, scene.shs, scene.scales, scene.rotations))
    g = torch.randn(3, H, W, device=dev)
    tf, tb = [], []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        st = ref_gpu.forward(m, op, shs=sh, scales=sc, rotations=ro, **d)
        t1 = time.perf_counter()
        ref_gpu.backward(st, g)
        t2 = time.perf_counter()
        tf.append(t1 - t0); tb.append(t2 - t1)
    ref_f, ref_b = 1e3 * min(tf), 1e3 * min(tb)
    settings = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, d["bg"], 1.0, d["viewmatrix"], d["projmatrix"], 3, d["campos"], False, False)
    rast = GaussianRasterizer(settings)
    leaves = [t.clone().requires_grad_(True) for t in (m, op, sh, sc, ro)]
    m2 = torch.zeros_like(m, requires_grad=True)
    tf, tb = [], []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        color, radii = rast(leaves[0], m2, leaves[1], shs=leaves[2], scales=leaves[3], rotations=leaves[4])
        torch.cuda.synchronize(); t1 = time.perf_counter()
        color.backward(g)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        tf.append(t1 - t0); tb.append(t2 - t1)
    my_f, my_b = 1e3 * min(tf), 1e3 * min(tb)
    res = dict(config=name, P=m.shape[0], W=W, H=H, R=st["R"], reference_hipcc_fwd_ms=ref_f, reference_hipcc_bwd_ms=ref_b,
               sugar_amd_fwd_ms=my_f, sugar_amd_bwd_ms=my_b, speedup_fwd_bwd=(ref_f + ref_b) / (my_f + my_b))
    print("== A/B", json.dumps(res))
    return res


if __name__ == "__main__":
    scene, cams, bg = syn.make_config("config1")
    parity("config1", scene, cams[0], bg)
    parity("config1 cam5 white", scene, cams[5], torch.ones(3))
    s2 = syn.make_scene(60000, 41, 0.004, 0.05)
    parity("60k 640x480", s2, syn.orbit_cameras(640, 480)[2], torch.zeros(3))
    out = [timing(n) for n in (sys.argv[1:] or ["config2", "metric"])]
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/ab_reference.json", "w"), indent=1)
