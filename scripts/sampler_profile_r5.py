"""Where one level-set sampling pass of bench.py's config-4 line spends its time (sugar_amd.sampler.sample_level_sets on the flat,
mesh-bound scene): stage by stage with a device synchronise behind each.    python scripts/sampler_profile_r5.py [config] [P]"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sugar_amd import sampler, synthetic as syn
from sugar_amd.field import level_set_points, scaled_rotation
from sugar_amd.knn import knn_points
from sugar_amd.sugar_patch import random_prefix_of_permutation

cfg = sys.argv[1] if len(sys.argv) > 1 else "config4"
dev = torch.device("cuda:0")
scene, cams, bg = syn.make_config(cfg, P=int(sys.argv[2]) if len(sys.argv) > 2 else None)
m, sc, ro, op = (t.to(dev) for t in (scene.means3D, scene.scales, scene.rotations, scene.opacities))
out = {"config": cfg, "P": int(m.shape[0])}


def timed(name, fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    out[name + "_ms"] = 1e3 * (time.perf_counter() - t0) / n
    return r


for ci in (0, 3):
    c = cams[ci]
    cam = c._replace(viewmatrix=c.viewmatrix.to(dev), projmatrix=c.projmatrix.to(dev), campos=c.campos.to(dev))
    depth = timed(f"cam{ci}_depth_render", lambda: sampler.render_depth(m, sc, ro, op, cam))
    flat = depth.reshape(-1)
    valid = timed(f"cam{ci}_valid_pixels", lambda: torch.logical_not(flat < 0.).nonzero(as_tuple=True)[0])
    out[f"cam{ci}_n_valid"] = int(valid.shape[0])
    picked = timed(f"cam{ci}_pick", lambda: valid[random_prefix_of_permutation(valid.shape[0], min(124_000, valid.shape[0]), dev)])
    world = timed(f"cam{ci}_unproject", lambda: sampler.unproject_pixels(picked, flat, cam))
    nbr = timed(f"cam{ci}_knn16", lambda: knn_points(world[None], m[None], K=16).idx[0])
    d0 = (world - m[nbr[:, 0]]).norm(dim=-1)
    out[f"cam{ci}_dist_to_nearest_quantiles"] = [float(torch.quantile(d0, q)) for q in (0.5, 0.9, 0.99, 1.0)]
    B = timed(f"cam{ci}_scaled_rotation", lambda: scaled_rotation(ro, sc, inverse_scales=True))
    stds = timed(f"cam{ci}_view_std", lambda: sampler.view_std(m, ro, sc, cam.campos))
    res = timed(f"cam{ci}_level_sets", lambda: level_set_points(world, nbr, cam.campos.reshape(1, 3), m, B, op.reshape(-1, 1), stds))
    out[f"cam{ci}_points"] = {str(k): int(v["intersection_points"].shape[0]) for k, v in res.items()}
    timed(f"cam{ci}_whole_pass", lambda: sampler.sample_level_sets(m, sc, ro, op, cam))
# the k-NN on its own: self query over the cloud (the coarse trainers' rebuild)
timed("knn16_self_query", lambda: knn_points(m[None], m[None], K=16).idx, n=3)
print(json.dumps(out, indent=1))
