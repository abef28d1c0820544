"""Development probe: per-call counters of the ball kernel (library built with -DSGR_KNN_STATS) against config 4's surface cloud.
   SGR_LIB_PATH=sugar_amd/variants/lib_knnstats.so python scripts/knn_stats_probe.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from sugar_amd import synthetic as syn
from sugar_amd.knn import knn_points
dev = torch.device("cuda:0")
b = syn.make_bound_scene(1_000_000, 4)
p = b.scene.means3D.to(dev)
g = torch.Generator().manual_seed(0)
base = b.scene.means3D[torch.randint(0, p.shape[0], (124_000,), generator=g)]
n = base / base.norm(dim=1, keepdim=True)
for off in (0.0, 0.02, 0.05, 0.2, 1.0):
    q = (base + off * n + 0.002 * torch.randn(base.shape, generator=g)).to(dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    knn_points(q[None], p[None], K=16); torch.cuda.synchronize()
    print(f"offset {off}: {1e3 * (time.perf_counter() - t0):.2f} ms", file=sys.stderr, flush=True)
