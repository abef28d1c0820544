#!/usr/bin/env bash
set -uo pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/r06"; mkdir -p "$OUT"; cd "$R"
timeout 600 python -m pytest tests/test_gpu_deep.py -q -x 2>&1 | tail -15
timeout 1500 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu_call8.log" 2>&1; echo "pytest rc $? $(tail -1 $OUT/pytest_gpu_call8.log)"
grep -E "^(FAILED|ERROR)" "$OUT/pytest_gpu_call8.log" | head
for dm in 1024 0; do
  for rep in 1 2; do
    SGR_DEEP_MIN=$dm timeout 600 python bench.py --workload config4 --plain-3dgs-step --no-cpu-baseline --no-reference-loop --cameras 0 --drift-steps 0 --no-densify-variant --steps 60 > "$OUT/bench_c4_deep${dm}_$rep.json" 2> "$OUT/bench_c4_deep${dm}_$rep.err"
    python - "$OUT/bench_c4_deep${dm}_$rep.json" "config4 deep_min=$dm rep $rep" <<'P'
import json,sys
d=json.load(open(sys.argv[1])); s=d["stages_ms"]
print(sys.argv[2], round(d["value"],1), round(d["ms_per_step"],4), "blend_fwd", round(s["blend_fwd"],4), "blend_bwd", round(s["blend_bwd"],4), "cover", d["stages_cover_frac"] and round(d["stages_cover_frac"],3))
P
  done
done
for dm in 1024 0; do
  SGR_DEEP_MIN=$dm timeout 600 python bench.py --no-cpu-baseline --no-reference-loop --cameras 0 --drift-steps 0 --no-densify-variant --steps 100 > "$OUT/bench_metric_deep${dm}.json" 2> "$OUT/bench_metric_deep${dm}.err"
  python - "$OUT/bench_metric_deep${dm}.json" "metric deep_min=$dm" <<'P'
import json,sys
d=json.load(open(sys.argv[1])); s=d["stages_ms"]
print(sys.argv[2], round(d["value"],1), round(d["ms_per_step"],4), "blend_fwd", round(s["blend_fwd"],4))
P
done
