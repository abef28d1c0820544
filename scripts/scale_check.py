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


# ---- BASELINE config 3 in the shape of SuGaR's coarse-SDF step: colours from get_points_rgb as
# `colors_precomp`, an RGB render and a depth render (depth as colour, bg = max depth, coarse_sdf.py:579-590) with their
# backwards, the density-field regulariser on 1M samples x 16 neighbours, and the k-NN rebuild timed apart.
def sugar_coarse_step():
    from sugar_amd.shcolor import sh_to_rgb
    from sugar_amd.field import density_field, scaled_rotation
    from sugar_amd.knn import knn_points
    scene, cams, bg = syn.make_config("config3")
    cam = cams[0]
    H, W = cam.image_height, cam.image_width
    view, proj, campos = cam.viewmatrix.to(dev), cam.projmatrix.to(dev), cam.campos.to(dev)
    m, op, sh, sc, ro = (t.to(dev).requires_grad_(True) for t in (scene.means3D, scene.opacities, scene.shs, scene.scales, scene.rotations))
    P = m.shape[0]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    knn_idx = knn_points(m.detach()[None], m.detach()[None], K=16).idx[0]
    torch.cuda.synchronize(); knn_ms = 1e3 * (time.perf_counter() - t0)
    gen = torch.Generator().manual_seed(0)
    gi = torch.randint(0, P, (1_000_000,), generator=gen).to(dev)
    noise = torch.randn(1_000_000, 3, generator=gen).to(dev)
    g_img = torch.randn(3, H, W, device=dev)

    def step():
        colors = sh_to_rgb(sh, 4, positions=m, camera_centers=campos[None])
        st = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, bg.to(dev), 1.0, view, proj, 3, campos, False, False)
        rgb, _ = GaussianRasterizer(st)(m, torch.zeros_like(m, requires_grad=True), op, colors_precomp=colors, scales=sc, rotations=ro)
        depth = (torch.cat([m, torch.ones_like(m[:, :1])], dim=1) @ view)[:, 2:3]
        max_depth = depth.detach().max()
        st_d = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, max_depth.expand(3).contiguous(), 1.0, view, proj, 3, campos, False, False)
        dimg, _ = GaussianRasterizer(st_d)(m, torch.zeros_like(m, requires_grad=True), op, colors_precomp=depth.expand(-1, 3).contiguous(), scales=sc, rotations=ro)
        x = (m[gi] + sc[gi] * noise).detach()
        B = scaled_rotation(torch.nn.functional.normalize(ro, dim=-1), sc, inverse_scales=True)  # get_covariance(sqrt, inverse)
        _, dens = density_field(x, knn_idx[gi], m, B, op)
        loss = (rgb * g_img).mean() + dimg.mean() + dens.mean()
        loss.backward()
        for t in (m, op, sh, sc, ro):
            t.grad = None
        return loss

    step(); torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    r = dict(config="config3 coarse-SDF-shaped step", P=P, W=W, H=H, step_ms=1e3 * min(ts), knn_rebuild_ms=knn_ms,
             parts="sh_to_rgb + RGB render + depth render (fwd+bwd each) + density field 1M x 16 (fwd+bwd), stock torch glue")
    print(json.dumps(r), flush=True)
    return r


out.append(sugar_coarse_step())
json.dump(out, open("gpurun_out/scale_check.json", "w"), indent=1)
