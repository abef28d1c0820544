"""The sync-free sampling pass of bench.py's config-4 lines, N times (for a kernel trace: scripts/r06_call5.sh).
    python scripts/sampler_trace_r6.py [config4|config4_opaque] [passes]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sugar_amd import sampler, synthetic as syn

cfg = sys.argv[1] if len(sys.argv) > 1 else "config4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda:0")
scene, cams, bg = syn.make_config(cfg)
m, sc, ro, op = (t.to(dev) for t in (scene.means3D, scene.scales, scene.rotations, scene.opacities))
cams = [c._replace(viewmatrix=c.viewmatrix.to(dev), projmatrix=c.projmatrix.to(dev), campos=c.campos.to(dev)) for c in cams]
for i in range(n):
    r = sampler.sample_level_sets(m, sc, ro, op, cams[i % len(cams)], sync_free=True)
torch.cuda.synchronize()
print({str(k): int(v["count"]) for k, v in r.items()})
