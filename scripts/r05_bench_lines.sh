#!/usr/bin/env bash
# the bench lines of scripts/r05_on_box.sh alone (no tests, no PMC passes): gpurun_out/r05/lines/
set -uo pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/r05/lines"; mkdir -p "$OUT"; cd "$R"
{ rocm-smi --showclocks --showpower --showtemp --showperflevel 2>&1 | grep -v "^=\|^$" | head -20; } > "$OUT/box_state.txt"
python bench.py --steps 20 --warmup 5 > "$OUT/bench_metric.json" 2> "$OUT/bench_metric.err"
python -c "import json; j=json.load(open('$OUT/bench_metric.json')); print('metric', round(j['value'],1), round(j['ms_per_step'],4))"
for w in config2 config3 config4 config5; do
  python bench.py --workload $w --steps 20 --warmup 5 --preroll 64 --drift-steps 0 --no-densify-variant --no-reference-loop > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"
done
python bench.py --workload config5 --host-sync --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_config5_unmodified_caller.json" 2> /dev/null
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_kt
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python "$R/bench.py" --steps 20 --warmup 5 --preroll 24 --no-cpu-baseline --no-densify-variant --drift-steps 0 --no-reference-loop > "$OUT/bench_under_rocprof.log" 2>&1
python "$R/scripts/rocpd_summary.py" /tmp/prof_kt/kt_results.db 60 > "$OUT/kernel_stats.txt" 2>&1
grep '^{' "$OUT/bench_under_rocprof.log" | tail -1 > "$OUT/bench_under_rocprof.json" || true
