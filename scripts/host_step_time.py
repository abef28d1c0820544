"""Host-side cost of one TRAIN step (tiny scene: the kernels are negligible, so the step time is what the Python loop and the
launches cost).  bench.py's step is GPU-bound only while this stays below the GPU time of the step."""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sugar_amd import synthetic as syn
from sugar_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from sugar_amd.train_step import GaussianParams, ViewShardedTrainer
dev = torch.device("cuda:0")
scene = syn.make_scene(2000, 3, 0.01, 0.05)
cams = [c._replace(viewmatrix=c.viewmatrix.to(dev), projmatrix=c.projmatrix.to(dev), campos=c.campos.to(dev)) for c in syn.orbit_cameras(64, 64)]
gts = [torch.rand(3, 64, 64, device=dev) for _ in cams]
tr = ViewShardedTrainer(GaussianParams(scene, dev), GaussianRasterizer, GaussianRasterizationSettings, torch.zeros(3, device=dev))
for i in range(30): tr.step(cams[i % 8], gts[i % 8])
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(200): tr.step(cams[i % 8], gts[i % 8])
    torch.cuda.synchronize()
    print("train step wall on a tiny scene: %.1f us" % (1e6 * (time.perf_counter() - t0) / 200))
if "--profile" in sys.argv:
    pr = cProfile.Profile(); pr.enable()
    for i in range(200): tr.step(cams[i % 8], gts[i % 8])
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
