#!/usr/bin/env bash
set -uo pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/r06"; mkdir -p "$OUT"; cd "$R"
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu_call1.log" 2>&1; echo "pytest rc $? $(tail -1 $OUT/pytest_gpu_call1.log)"
bash scripts/r06_grad_switches.sh 0 1 3 4 8 16 32 64 28 60 127
SGR_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline > "$OUT/bench_gloo2_selflaunch.json" 2> "$OUT/bench_gloo2_selflaunch.err"; echo "gloo2 rc $?"; head -c 600 "$OUT/bench_gloo2_selflaunch.json"; echo
timeout 900 python bench.py > "$OUT/bench_metric_call1.json" 2> "$OUT/bench_metric_call1.err"; echo "bench rc $?"
python - <<'P'
import json
d=json.load(open("gpurun_out/r06/bench_metric_call1.json"))
print(d["value"], d["ms_per_step"], d["stages_ms"], d.get("stages_cover_frac"), d["roofline"]["frac"])
P
