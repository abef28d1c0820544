#!/usr/bin/env bash
set -uo pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/r06"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for dm in 3072 0; do
  rm -rf /tmp/prof_d
  SGR_DEEP_MIN=$dm timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_d -o kt -- python "$R/bench.py" --workload config4 --plain-3dgs-step --no-cpu-baseline --no-reference-loop --cameras 0 --drift-steps 0 --no-densify-variant --steps 20 --preroll 32 > "$OUT/trace_c4_deep$dm.log" 2>&1
  python "$R/scripts/rocpd_summary.py" /tmp/prof_d/kt_results.db 40 > "$OUT/kernels_c4_deep$dm.txt" 2>&1
  echo "== deep_min $dm"; grep -E "k_blend|k_deep" "$OUT/kernels_c4_deep$dm.txt" | cut -c1-120
done
cd "$R"
python - <<'P'
import torch, sys
sys.path.insert(0, ".")
from sugar_amd import synthetic as syn, _lib
from sugar_amd.train_step import GaussianParams, NativeTrainer
dev = torch.device("cuda:0")
scene, cams, bg = syn.make_config("config4")
W, H = cams[0].image_width, cams[0].image_height
cams = [c._replace(viewmatrix=c.viewmatrix.to(dev), projmatrix=c.projmatrix.to(dev), campos=c.campos.to(dev)) for c in cams]
gts = [torch.rand(3, H, W, device=dev) for _ in cams]
tr = NativeTrainer(GaussianParams(scene, dev), bg.to(dev), W, H)
for s in range(16):
    tr.step(cams[s % 8], gts[s % 8], cam_key=s % 8)
tr.synchronize()
for k in range(8):
    h = tr._hints[k][0]
    print("cam", k, "max hint", int(h.max()), "tiles > 1024:", int((h > 1024).sum()), "> 2048:", int((h > 2048).sum()), "> 3072:", int((h > 3072).sum()), "sum", int(h.sum()))
P
