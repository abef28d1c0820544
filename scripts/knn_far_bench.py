"""124k sampler-like queries against 1M reference points, 10 % of them outside the cloud (the verdict's item: <= 2.5 ms)."""
import sys, os, time, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sugar_amd.knn import knn_points
from sugar_amd import synthetic as syn
dev = torch.device("cuda:0")
pts = syn.make_scene(1_000_000, 7, 0.002, 0.03).means3D.to(dev)
g = torch.Generator().manual_seed(0)
N = 124_000
near = pts[torch.randint(0, pts.shape[0], (N,), generator=g).to(dev)] + 0.01 * torch.randn(N, 3, generator=g).to(dev)
out = {}
for frac in (0.0, 0.1, 0.3):
    q = near.clone()
    n_far = int(frac * N)
    if n_far:
        d = torch.randn(n_far, 3, generator=g); d = d / d.norm(dim=1, keepdim=True)
        q[:n_far] = (d * (1.8 + 1.5 * torch.rand(n_far, 1, generator=g))).to(dev)  # outside the [-1, 1]^3 cloud, towards a camera
    fn = lambda: knn_points(q[None], pts[None], K=16, method="grid")
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): r = fn()
    torch.cuda.synchronize()
    out[f"ms_far_{int(100 * frac)}pct"] = 1e3 * (time.perf_counter() - t0) / 10
    if frac == 0.1:
        b = knn_points(q[None, :20000], pts[None], K=16, method="brute")
        out["identical_to_exhaustive_first_20000"] = bool(torch.equal(b.idx, r.idx[:, :20000]) and torch.equal(b.dists, r.dists[:, :20000]))
print(json.dumps(out))
