"""Developer diagnostic (GPU box): where an iteration of the reference's own optimisation loop (oracle/reference_loop.py =
gaussian_splatting/train.py:69-128) spends its time on the drop-ins, at the metric workload.  torch profiler, kernels by GPU time."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sugar_amd import synthetic as syn, shims
from oracle import reference_loop as rl

dev = torch.device("cuda:0")
scene, cams, bg = syn.make_config("metric")
patch = "--no-patch-losses" not in sys.argv
all_patches = "--all-patches" in sys.argv   # + the optimiser and the densification statistics (round 5)
ref = rl.import_reference()
if patch:
    shims.install_losses()
    ref = rl.import_reference()
if all_patches:
    shims.install_optimizer()
    shims.install_densifier()
opt = rl.optimization_params()
gaussians = rl.make_gaussians(ref, scene, dev, opt)
gts = [torch.rand(3, c.image_height, c.image_width) for c in cams]
loop = rl.Loop(ref, gaussians, [rl.make_viewpoint(c, gt, dev) for c, gt in zip(cams, gts)], bg.to(dev), opt=opt)
for _ in range(10):
    loop.loop_body()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
N = 8
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(N):
        loop.loop_body()
    torch.cuda.synchronize()
print(f"patch_losses={patch}; all_patches={all_patches}; {N} iterations; optimizer {type(gaussians.optimizer).__name__}")
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=70))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=14, max_name_column_width=70))
