"""Are the intermittent slow bench windows the interpreter's garbage collector?  Windows of 20 train steps at the metric workload,
first as the process stands, then after gc.freeze(): prints each phase's median / max window and the collections that ran."""
import gc, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sugar_amd import synthetic as syn
from sugar_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from sugar_amd.train_step import GaussianParams, ViewShardedTrainer
dev = torch.device("cuda:0")
scene, cams, bg = syn.make_config("metric")
cams = [c._replace(viewmatrix=c.viewmatrix.to(dev), projmatrix=c.projmatrix.to(dev), campos=c.campos.to(dev)) for c in cams]
H, W = cams[0].image_height, cams[0].image_width
gts = [torch.rand(3, H, W, device=dev) for _ in cams]
tr = ViewShardedTrainer(GaussianParams(scene, dev), GaussianRasterizer, GaussianRasterizationSettings, bg.to(dev))
for i in range(16): tr.step(cams[i % 8], gts[i % 8])
torch.cuda.synchronize()
stats = []
gc.callbacks.append(lambda phase, info: stats.append((info["generation"], time.perf_counter())) if phase == "start" else None)

def windows(n):
    out = []
    for w in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(20): tr.step(cams[i % 8], gts[i % 8])
        torch.cuda.synchronize(); out.append(1e3 * (time.perf_counter() - t0) / 20)
    return out

for name in ("as is", "after gc.freeze()"):
    stats.clear()
    ws = sorted(windows(int(sys.argv[1]) if len(sys.argv) > 1 else 40))
    gens = [g for g, _ in stats]
    print(f"{name:18s} median {ws[len(ws)//2]:.3f}  max {ws[-1]:.3f}  second {ws[-2]:.3f} ms/step;  collections gen0/1/2: "
          f"{gens.count(0)}/{gens.count(1)}/{gens.count(2)};  tracked objects {len(gc.get_objects())}")
    gc.collect(); gc.freeze()
