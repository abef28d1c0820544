#!/usr/bin/env bash
set -uo pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/r06"; mkdir -p "$OUT"; cd "$R"
timeout 300 python -m pytest tests/test_gpu_pick.py -q 2>&1 | tail -2
for dm in 1024 2048 3072 0; do
    SGR_DEEP_MIN=$dm timeout 600 python bench.py --workload config4 --plain-3dgs-step --no-cpu-baseline --no-reference-loop --cameras 0 --drift-steps 0 --no-densify-variant --steps 60 > "$OUT/bench_c4_deep${dm}.json" 2> "$OUT/bench_c4_deep${dm}.err"
    python - "$OUT/bench_c4_deep${dm}.json" "config4 deep_min=$dm" <<'P'
import json,sys
d=json.load(open(sys.argv[1])); s=d["stages_ms"]
print(sys.argv[2], round(d["value"],1), round(d["ms_per_step"],4), "blend_fwd", round(s["blend_fwd"],4), "blend_bwd", round(s["blend_bwd"],4))
P
done
for dm in 2048 3072; do
  SGR_DEEP_MIN=$dm timeout 600 python bench.py --no-cpu-baseline --no-reference-loop --cameras 0 --drift-steps 0 --no-densify-variant --steps 100 > "$OUT/bench_metric_deep${dm}.json" 2> "$OUT/bench_metric_deep${dm}.err"
  python - "$OUT/bench_metric_deep${dm}.json" "metric deep_min=$dm" <<'P'
import json,sys
d=json.load(open(sys.argv[1])); s=d["stages_ms"]
print(sys.argv[2], round(d["value"],1), round(d["ms_per_step"],4), "blend_fwd", round(s["blend_fwd"],4))
P
done
timeout 300 python -m pytest tests/test_gpu_deep.py tests/test_gpu_native_trainer.py -q 2>&1 | tail -2
