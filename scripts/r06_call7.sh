#!/usr/bin/env bash
set -uo pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/r06"; mkdir -p "$OUT"; cd "$R"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x > "$OUT/pytest_gpu_call7.log" 2>&1; echo "pytest rc $? $(tail -1 $OUT/pytest_gpu_call7.log)"
grep -E "^(FAILED|ERROR)" "$OUT/pytest_gpu_call7.log" | head
for v in 0 1; do
  if [ $v = 1 ]; then export SGR_NO_SORT_OVERLAP=1; else unset SGR_NO_SORT_OVERLAP; fi
  for rep in 1 2 3; do
    timeout 600 python bench.py --no-cpu-baseline --no-reference-loop --cameras 0 --drift-steps 0 --no-densify-variant --steps 100 > "$OUT/bench_overlap_${v}_$rep.json" 2> "$OUT/bench_overlap_${v}_$rep.err"
    python - "$OUT/bench_overlap_${v}_$rep.json" "no_overlap=$v rep $rep" <<'P'
import json,sys
d=json.load(open(sys.argv[1])); s=d["stages_ms"]
print(sys.argv[2], round(d["value"],1), round(d["ms_per_step"],4), "pre", round(s["preprocess"],4), "sort", round(s["depth_sort"],4), "cover", round(d["stages_cover_frac"],3))
P
  done
done
