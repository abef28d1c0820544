"""Forward(+backward) at the larger BASELINE configs (config3: 2M @1080p, config5: 6M @ 4K forward only): timings and sanity."""
import os, sys, time, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sugar_amd import synthetic as syn
from sugar_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _C
dev = torch.device("cuda:0")
out = []
for name, do_bwd in (("config3", True), ("config5", False)):
    scene, cams, bg = syn.make_config(name)
    cam = cams[0]
    H, W = cam.image_height, cam.image_width
    st = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, bg.to(dev), 1.0, cam.viewmatrix.to(dev), cam.projmatrix.to(dev), 3, cam.campos.to(dev), False, False)
    rast = GaussianRasterizer(st)
    m, op, sh, sc, ro = (t.to(dev).requires_grad_(do_bwd) for t in (scene.means3D, scene.opacities, scene.shs, scene.scales, scene.rotations))
    m2 = torch.zeros_like(m, requires_grad=do_bwd)
    g = torch.randn(3, H, W, device=dev)
    tf, tb = [], []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        color, radii = rast(m, m2, op, shs=sh, scales=sc, rotations=ro)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        if do_bwd:
            color.backward(g); torch.cuda.synchronize()
        t2 = time.perf_counter()
        tf.append(t1 - t0); tb.append(t2 - t1)
    R = _C.last_forward["num_rendered"]
    r = dict(config=name, P=m.shape[0], W=W, H=H, R=R, visible=int((radii > 0).sum()), fwd_ms=1e3 * min(tf), bwd_ms=(1e3 * min(tb) if do_bwd else None),
             finite=bool(torch.isfinite(color).all()), mean=float(color.mean()))
    print(json.dumps(r), flush=True)
    out.append(r)
    del m, op, sh, sc, ro, m2, color, radii
    torch.cuda.empty_cache()
json.dump(out, open("gpurun_out/scale_check.json", "w"), indent=1)
