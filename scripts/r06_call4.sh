#!/usr/bin/env bash
set -uo pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/r06"; mkdir -p "$OUT"; cd "$R"
timeout 1500 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu_call4.log" 2>&1; echo "pytest rc $? $(tail -1 $OUT/pytest_gpu_call4.log)"
grep -E "^(FAILED|ERROR)" "$OUT/pytest_gpu_call4.log" | head -20
for W in config4 config4_opaque; do
  timeout 900 python bench.py --workload $W --no-cpu-baseline > "$OUT/bench_${W}_call4.json" 2> "$OUT/bench_${W}_call4.err"; echo "bench $W rc $?"
done
timeout 600 python bench.py --force-collectives --native-collectives --no-cpu-baseline --no-reference-loop --cameras 0 --drift-steps 0 --no-densify-variant > "$OUT/bench_metric_forced_native_call4.json" 2> "$OUT/bench_metric_forced_native_call4.err"; echo "bench forced rc $?"
python - <<'P'
import json
for n in ("bench_config4_call4","bench_config4_opaque_call4","bench_metric_forced_native_call4"):
    try:
        d=json.load(open(f"gpurun_out/r06/{n}.json"))
        print(n, round(d["value"],1), round(d["ms_per_step"],4), d.get("comm_exposed_ms"), json.dumps(d.get("refine_step"))[:900])
        print("   stages", {k: round(v,4) for k,v in d["stages_ms"].items()})
    except Exception as e: print(n, "ERR", e)
P
