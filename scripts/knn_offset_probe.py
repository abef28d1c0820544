"""k-NN(16) time against a surface cloud (config 4's centres) for 124k queries at a controlled distance from the surface."""
import json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from sugar_amd import synthetic as syn
from sugar_amd.knn import knn_points
dev = torch.device("cuda:0")
b = syn.make_bound_scene(1_000_000, 4)
p = b.scene.means3D.to(dev)
g = torch.Generator().manual_seed(0)
base = b.scene.means3D[torch.randint(0, p.shape[0], (124_000,), generator=g)]
n = base / base.norm(dim=1, keepdim=True)
out = {}
for off in (0.0, 0.005, 0.02, 0.05, 0.1, 0.2, 0.4, 1.0, -0.05, -0.2):
    q = (base + off * n + 0.002 * torch.randn(base.shape, generator=g)).to(dev)
    knn_points(q[None], p[None], K=16); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): r = knn_points(q[None], p[None], K=16)
    torch.cuda.synchronize()
    out[str(off)] = round(1e3 * (time.perf_counter() - t0) / 3, 3)
print(json.dumps(out))
