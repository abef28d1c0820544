#!/usr/bin/env bash
# same-box A/B of the backward blend's target occupancy (SGR_BLEND_DEFS=-DSGR_BWD_WAVES=..)
set -uo pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/r06/bw_ab"; mkdir -p "$OUT"; cd "$R"
for round in 1 2; do
  for name in default bw4 bw6; do
    if [ $name = default ]; then unset SGR_LIB_PATH; else export SGR_LIB_PATH="$R/sugar_amd/variants/lib_$name.so"; fi
    python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-densify-variant --drift-steps 0 --no-reference-loop --cameras 0 > "$OUT/bench_${name}_$round.json" 2> "$OUT/bench_${name}_$round.err"
    python -c "import json; d=json.load(open('$OUT/bench_${name}_$round.json')); print('$name $round', round(d['ms_per_step'],4), 'blend_bwd', round(d['stages_ms']['blend_bwd'],4))"
  done
done
