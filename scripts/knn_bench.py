import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sugar_amd.knn import knn_points, distCUDA2
from sugar_amd import synthetic as syn
dev = torch.device("cuda:0")
for P in (100_000, 1_000_000):
    pts = syn.make_scene(P, 3, 0.01, 0.02).means3D.to(dev)
    for name, fn in (("distCUDA2 grid", lambda: distCUDA2(pts, method="grid")), ("knn K=16 grid", lambda: knn_points(pts[None], pts[None], K=16, method="grid")),
                     ("knn K=16 exhaustive", lambda: knn_points(pts[None], pts[None], K=16, method="brute"))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        print(f"P={P} {name}: {1e3*(time.perf_counter()-t0):.1f} ms", flush=True)
