#!/usr/bin/env bash
# gpurun --timeout 3000 -- 'bash scripts/r06_final.sh'     everything profiles/r06_* of the final state is made from -> gpurun_out/r06/final/
set -uo pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/r06/final"; mkdir -p "$OUT"; cd "$R"
{ rocm-smi --showclocks --showpower --showtemp --showperflevel 2>&1 | grep -v "^=\|^$" | head -20; } > "$OUT/box_state.txt"
timeout 1500 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc $? $(tail -1 $OUT/pytest_gpu.log)"
cp gpurun_out/fullsize_parity.json "$OUT/fullsize_parity.json" 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc $?"
timeout 1200 python bench.py > "$OUT/bench_metric.json" 2> "$OUT/bench_metric.err"; echo "bench rc $?"
for w in config2 config3 config5 config4 config4_opaque; do
  timeout 900 python bench.py --workload $w --steps 20 --warmup 5 --preroll 64 --drift-steps 0 --no-densify-variant --no-reference-loop > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"; echo "bench $w rc $?"
done
timeout 600 python bench.py --force-collectives --native-collectives --no-cpu-baseline --no-reference-loop --cameras 0 --drift-steps 0 --no-densify-variant > "$OUT/bench_metric_forced_collectives_in_library.json" 2> "$OUT/bench_forced.err"
SGR_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline > "$OUT/bench_two_rank_gloo_selflaunch.json" 2> "$OUT/bench_gloo2.err"; echo "gloo2 rc $?"
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_kt
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python "$R/bench.py" --steps 20 --warmup 5 --preroll 24 --no-cpu-baseline --no-densify-variant --drift-steps 0 --no-reference-loop --cameras 0 > "$OUT/bench_under_rocprof.log" 2>&1
python "$R/scripts/rocpd_summary.py" /tmp/prof_kt/kt_results.db 60 > "$OUT/kernel_stats.txt" 2>&1
grep '^{' "$OUT/bench_under_rocprof.log" | tail -1 > "$OUT/bench_under_rocprof.json" || true
rm -rf /tmp/prof_kt
# PMC passes over the same command (separate passes, no trace domains besides --kernel-trace): traffic, instruction counts, LDS conflicts
for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS; do
  rm -rf /tmp/prof_pmc
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/prof_pmc -o pmc -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --preroll 16 --no-densify-variant --drift-steps 0 --no-reference-loop --cameras 0 > /tmp/pmc.log 2>&1
  python "$R/scripts/rocpd_pmc_summary.py" /tmp/prof_pmc/pmc_results.db k_ > "$OUT/pmc_metric_$C.txt" 2>&1
  grep -E "k_blend" "$OUT/pmc_metric_$C.txt" | cut -c1-130
done
rm -rf /tmp/prof_pmc
cd "$R"
python scripts/pmc_reduce.py "$OUT" metric pmc_metric_ r06 > "$OUT/pmc_blend_fwd_metric.json" 2> "$OUT/pmc_reduce.err"; cat "$OUT/pmc_blend_fwd_metric.json" | head -20
python - <<'P'
import json
d=json.load(open("gpurun_out/r06/final/bench_metric.json"))
print("metric", round(d["value"],1), round(d["ms_per_step"],4), "cover", d.get("stages_cover_frac"), "roofline", round(d["roofline"]["frac"],4))
for w in ("config2","config3","config5","config4","config4_opaque"):
    try:
        j=json.load(open(f"gpurun_out/r06/final/bench_{w}.json")); print(w, round(j["value"],1), round(j["ms_per_step"],4))
    except Exception as e: print(w, "ERR", e)
P
