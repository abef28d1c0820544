#!/usr/bin/env bash
# gpurun -- 'bash scripts/r03_on_box.sh'   round-3 evidence in one call: box state, bench lines of every workload, kernel trace,
# PMC passes (separate runs, --kernel-trace only) -- summaries under gpurun_out/r03/
set -uo pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/r03"
mkdir -p "$OUT"
cd "$R"
{ echo "# partition modes"; rocm-smi --showcomputepartition --showmemorypartition 2>&1 | grep -v "^=\|^$" | head -8; echo "# rocm-smi before"; rocm-smi --showclocks --showpower --showtemp --showmemuse --showperflevel 2>&1 | grep -v "^=\|^$" | head -40; } > "$OUT/box_state.txt"
( for i in $(seq 1 60); do rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|mclk\|fclk\|Power (W)" | tr -s "\t " " " | tr "\n" "|"; echo; sleep 0.25; done ) > "$OUT/box_clocks_during_bench.txt" &
SAMPLER=$!
python bench.py --steps 20 --warmup 5 > "$OUT/bench_metric.json" 2> "$OUT/bench_metric.err"
kill $SAMPLER 2>/dev/null; wait $SAMPLER 2>/dev/null
{ echo "# rocm-smi right after the metric bench"; rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -v "^=\|^$" | head -30; } >> "$OUT/box_state.txt"
for w in config2 config3 config5; do
  python bench.py --workload $w --steps 20 --warmup 5 --preroll 64 --drift-steps 0 --no-densify-variant > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"
done
python scripts/config4_rehearsal.py > "$OUT/config4_rehearsal.json" 2> "$OUT/config4.err"
python bench.py --steps 20 --warmup 5 --force-collectives --no-cpu-baseline --drift-steps 0 --no-densify-variant > "$OUT/bench_metric_forced_collectives.json" 2> "$OUT/bench_fc.err"
python bench.py --steps 20 --warmup 5 --python-step --no-cpu-baseline --drift-steps 0 > "$OUT/bench_metric_python_step.json" 2> "$OUT/bench_py.err"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python "$R/bench.py" --steps 20 --warmup 5 --preroll 24 --no-cpu-baseline --no-densify-variant --drift-steps 0 > "$OUT/bench_under_rocprof.log" 2>&1
python "$R/scripts/rocpd_summary.py" /tmp/prof_kt/kt_results.db 60 > "$OUT/kernel_stats.txt" 2>&1
grep '^{' "$OUT/bench_under_rocprof.log" | tail -1 > "$OUT/bench_under_rocprof.json" || true
rm -rf /tmp/prof_kt
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES"; do
  TAG=$(echo $C | cut -d' ' -f1)
  rm -rf /tmp/prof_pmc
  rocprofv3 --kernel-trace --pmc $C -d /tmp/prof_pmc -o pmc -- python "$R/bench.py" --steps 3 --warmup 1 --preroll 16 --no-cpu-baseline --no-densify-variant --drift-steps 0 > /tmp/pmc.log 2>&1
  python "$R/scripts/rocpd_pmc_summary.py" /tmp/prof_pmc/pmc_results.db k_ > "$OUT/pmc_$TAG.txt" 2>&1
done
rm -rf /tmp/prof_pmc
ls -la "$OUT"
