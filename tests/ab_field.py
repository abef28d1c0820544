"""A/B on the GPU box for SURVEY.md section 8 rows a18/a20/a21 at BASELINE sizes: this repository's HIP density field
(forward + backward), level-set sampler and k-NN versus the reference's own tensor code (oracle/sugar_field_torch.py, a
restatement of sugar_scene/sugar_model.py:1266-1276 and :1971-2079) run in float32 ON THE SAME MI355X through stock torch.
    python tests/ab_field.py            -> one JSON line (also written to gpurun_out/ab_field.json)
"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sugar_amd import synthetic as syn, build
from sugar_amd.field import density_field, level_set_points
from sugar_amd.knn import knn_points
from oracle import sugar_field_torch as ref

build.build()
dev = torch.device("cuda:0")


def say(*a):
    if os.environ.get("AB_FIELD_VERBOSE"):
        torch.cuda.synchronize(); print("[ab_field]", *a, file=sys.stderr, flush=True)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


def quat_to_rotmat(q):
    r, x, y, z = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)


def main():
    P = int(os.environ.get("AB_FIELD_P", 1_000_000))
    N = int(os.environ.get("AB_FIELD_N", 1_000_000))     # coarse_sdf.py:150 samples per step
    NPIX = int(os.environ.get("AB_FIELD_PIX", 124_000))  # pixels per view
    K = 16
    sc = syn.make_scene(P, 11, 0.004, 0.03)
    pts = sc.means3D.to(dev)
    Rm = quat_to_rotmat(sc.rotations.to(dev))
    B = (Rm * (1.0 / sc.scales.to(dev).clamp(min=1e-8))[:, None]).contiguous()
    strengths = sc.opacities.to(dev)
    out = {"P": P, "samples": N, "pixels": NPIX, "K": K}
    say("scene")
    out["knn_ms"] = timed(lambda: knn_points(pts[None], pts[None], K=K), 3)
    knn_idx = knn_points(pts[None], pts[None], K=K).idx[0]
    g = torch.Generator(device="cpu").manual_seed(0)
    gi = torch.randint(0, P, (N,), generator=g).to(dev)
    x = (pts[gi] + sc.scales.to(dev)[gi] * torch.randn(N, 3, generator=g).to(dev)).contiguous()
    nb = knn_idx[gi].contiguous()
    go = torch.randn(N, K, generator=g).to(dev); gd = torch.randn(N, generator=g).to(dev)

    def run(fn):
        xr = x.clone().requires_grad_(True); cr = pts.clone().requires_grad_(True)
        Br = B.clone().requires_grad_(True); sr = strengths.clone().requires_grad_(True)
        o, d = fn(xr, nb, cr, Br, sr, 1.0)
        ((o * go).sum() + (d * gd).sum()).backward()
        return o, d, xr.grad, cr.grad, Br.grad, sr.grad

    def fwd_only(fn):
        with torch.no_grad():
            return fn(x, nb, pts, B, strengths, 1.0)

    say("inputs")
    def atomics_variant(fn):
        from sugar_amd import field as F
        F._DensityField.use_gather = False
        try:
            return fn()
        finally:
            F._DensityField.use_gather = True

    a = run(density_field); say("hip density"); b = run(ref.density_field); say("torch density")
    rel = lambda u, v: float((u - v).norm() / v.norm())
    out["density_field"] = {
        "hip_fwd_ms": timed(lambda: fwd_only(density_field)), "torch_fwd_ms": timed(lambda: fwd_only(ref.density_field)),
        "hip_fwd_bwd_ms": timed(lambda: run(density_field)), "hip_fwd_bwd_atomics_ms": atomics_variant(lambda: timed(lambda: run(density_field))),
        "torch_fwd_bwd_ms": timed(lambda: run(ref.density_field)),
        "rel_err": {n: rel(u, v) for n, u, v in zip(["opac", "dens", "dx", "dcenters", "dB", "dstrengths"], a, b)},
        "algorithmic_bytes_fwd": N * (12 + K * 8 + K * 52 + K * 4 + 4),
    }
    d = out["density_field"]
    d["hip_fwd_GBs"] = d["algorithmic_bytes_fwd"] / (d["hip_fwd_ms"] * 1e-3) / 1e9
    del a, b
    say("density timed")
    torch.cuda.empty_cache()
    # level-set sampler
    gp = torch.randint(0, P, (NPIX,), generator=g).to(dev)
    world = (pts[gp] + 0.3 * sc.scales.to(dev)[gp] * torch.randn(NPIX, 3, generator=g).to(dev)).contiguous()
    nbp = knn_idx[gp].contiguous()
    cam_center = torch.tensor([2.5, -1.0, 0.8], device=dev)
    to_cam = torch.nn.functional.normalize(cam_center - pts, dim=-1)
    gstd = (sc.scales.to(dev) * (Rm.transpose(1, 2) @ to_cam[..., None])[..., 0]).norm(dim=-1)
    levels = (0.1, 0.3, 0.5)
    with torch.no_grad():
        say("level-set inputs")
        h = level_set_points(world, nbp, cam_center, pts, B, strengths, gstd, levels)
        say("hip level set")
        # the reference walks the samples in passes (sugar_model.py:1988-2011); stock torch's batched 3x3 product faults on
        # this stack beyond 2^24 batch entries, so the passes here are 40k pixels (13.4M (sample, neighbour) pairs) each
        def ref_level_sets():
            parts = [ref.level_set_points(world[i:i + 40000], nbp[i:i + 40000], cam_center, pts, B, strengths, gstd, levels)
                     for i in range(0, NPIX, 40000)]
            return {lv: {k: torch.cat([p_[lv][k] for p_ in parts]) for k in ("valid", "intersection_points", "normals")}
                    for lv in levels}
        r = ref_level_sets()
        say("torch level set")
        mism = max(float((h[lv]["valid"] != r[lv]["valid"]).float().mean()) for lv in levels)
        out["level_set"] = {
            "hip_ms": timed(lambda: level_set_points(world, nbp, cam_center, pts, B, strengths, gstd, levels)),
            "torch_ms": timed(ref_level_sets),
            "valid_mismatch_frac": mism, "valid_frac": float(h[levels[1]]["valid"].float().mean()),
            "pair_evals": NPIX * 21 * K,
        }
    for k in ("density_field", "level_set"):
        s = out[k]
        a_, b_ = ("hip_fwd_bwd_ms", "torch_fwd_bwd_ms") if k == "density_field" else ("hip_ms", "torch_ms")
        s["speedup"] = s[b_] / s[a_]
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/ab_field.json", "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
