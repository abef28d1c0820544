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
# packed-power variant of the exact forward body: A/B, two interleaved rounds
for rep in 1 2; do
  for lib in "" "$R/sugar_amd/variants/lib_fwdpk.so"; do
    if [ -n "$lib" ]; then export SGR_LIB_PATH="$lib"; tag=pk; else unset SGR_LIB_PATH; tag=plain; fi
    SGR_DEEP_MIN=0 timeout 600 python bench.py --no-cpu-baseline --no-reference-loop --cameras 0 --drift-steps 0 --no-densify-variant --steps 100 > "$OUT/bench_fwdpk_${tag}_$rep.json" 2> "$OUT/bench_fwdpk_${tag}_$rep.err"
    python - "$OUT/bench_fwdpk_${tag}_$rep.json" "fwd body $tag rep $rep" <<'P'
import json,sys
d=json.load(open(sys.argv[1])); s=d["stages_ms"]
print(sys.argv[2], round(d["value"],1), round(d["ms_per_step"],4), "blend_fwd(timed region)", round(d["roofline"]["launch_ms"],4), "blend_fwd(stage pass)", round(s["blend_fwd"],4))
P
  done
done
unset SGR_LIB_PATH
SGR_LIB_PATH="$R/sugar_amd/variants/lib_fwdpk.so" timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -k "metric-5-True-sh or config4-1-True-sh" 2>&1 | tail -2
