#!/usr/bin/env bash
set -uo pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/r06"; mkdir -p "$OUT"; cd "$R"
for i in 1 2; do
  timeout 1500 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu_call13_$i.log" 2>&1; echo "pytest run $i rc $? $(tail -1 $OUT/pytest_gpu_call13_$i.log)"
  grep -E "^(FAILED|ERROR)" "$OUT/pytest_gpu_call13_$i.log" | head
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.err"; echo "bench rc $?"
python -c "
import json; d=json.load(open('$OUT/bench_driver_cmd.json')); print(d['value'], d['ms_per_step'], d['stages_cover_ok'], d['roofline']['frac'], d['roofline_valu']['frac'])"
