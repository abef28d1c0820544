#!/usr/bin/env bash
set -uo pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/r06"; mkdir -p "$OUT"; cd "$R"
timeout 900 python -m pytest tests/test_gpu_pick.py tests/test_gpu_parity.py tests/test_gpu_field.py tests/test_gpu_sugar_field.py tests/test_gpu_reference_sugar.py tests/test_gpu_reference.py tests/test_gpu_reference_trainer.py -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for W in config4 config4_opaque; do
  rm -rf /tmp/prof_s
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o kt -- python "$R/scripts/sampler_trace_r6.py" $W 16 > "$OUT/sampler_trace_$W.log" 2>&1
  python "$R/scripts/rocpd_summary.py" /tmp/prof_s/kt_results.db 40 > "$OUT/sampler_kernels_$W.txt" 2>&1
  echo "== $W"; tail -1 "$OUT/sampler_trace_$W.log"; head -16 "$OUT/sampler_kernels_$W.txt" | cut -c1-130
done
cd "$R"
for W in config4 config4_opaque; do
  timeout 900 python bench.py --workload $W --no-cpu-baseline --steps 20 --warmup 5 --preroll 64 > "$OUT/bench_${W}_call12.json" 2> "$OUT/bench_${W}_call12.err"
  python - "$OUT/bench_${W}_call12.json" $W <<'P'
import json,sys
d=json.load(open(sys.argv[1])); r=d["refine_step"]
print(sys.argv[2], round(d["value"],1), round(d["ms_per_step"],4), r["sampler_pass_ms_device"], r["level_set_points_last_view"])
P
done
