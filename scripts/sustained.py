"""Burst versus sustained: bench.py times 20 steps (25 ms) after a few warm-up steps; this runs the same step for several
seconds and prints ms/step per window of 250 steps, with the GPU clock and power rocm-smi reports in between."""
import os, subprocess, sys, time, torch
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

def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
        keep = [l.strip() for l in out.splitlines() if ("sclk" in l or "mclk" in l or "Power" in l) and "GPU[0]" in l]
        return " | ".join(k.split(":", 1)[-1].strip() for k in keep)
    except Exception as e:
        return f"(rocm-smi: {e})"

for i in range(13): tr.step(cams[i % 8], gts[i % 8])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20): tr.step(cams[i % 8], gts[i % 8])
torch.cuda.synchronize()
print(f"first 20 steps after 13 warm-up steps: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms/step  R = {tr.last_num_rendered / 1e6:.2f} M")
n_win = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for w in range(n_win):
    t0 = time.perf_counter()
    for i in range(250): tr.step(cams[i % 8], gts[i % 8])
    torch.cuda.synchronize()
    dt = 1e3 * (time.perf_counter() - t0) / 250
    print(f"steps {20 + 250 * w:5d}..{20 + 250 * (w + 1):5d}: {dt:.3f} ms/step  R = {tr.last_num_rendered / 1e6:.2f} M" + (f"   [{smi()}]" if w % 4 == 3 else ""), flush=True)
