#!/usr/bin/env bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + separate PMC passes of bench.py, reduced to small text
# summaries under gpurun_out/ (the rocpd databases are too large to merge back and are deleted).
#   gpurun -- 'bash scripts/profile_on_box.sh r01'
set -uo pipefail
TAG="${1:-r01}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/profile_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --preroll 0 > "$OUT/bench_under_rocprof.log" 2>&1
python "$R/scripts/rocpd_summary.py" /tmp/prof_kt/kt_results.db 60 > "$OUT/kernel_stats.txt" 2>&1
grep '^{' "$OUT/bench_under_rocprof.log" | tail -1 > "$OUT/bench_under_rocprof.json" || true
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d /tmp/prof_$C -o pmc -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --preroll 0 > "$OUT/bench_pmc_$C.log" 2>&1
  python "$R/scripts/rocpd_pmc_summary.py" /tmp/prof_$C/pmc_results.db k_ > "$OUT/pmc_$C.txt" 2>&1
done
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" ; do
  rocprofv3 --kernel-trace --pmc $C -d /tmp/prof_lds -o pmc -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --preroll 0 > "$OUT/bench_pmc_lds.log" 2>&1
  python "$R/scripts/rocpd_pmc_summary.py" /tmp/prof_lds/pmc_results.db k_blend > "$OUT/pmc_lds.txt" 2>&1
done
rm -rf /tmp/prof_kt /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE /tmp/prof_lds
rm -f "$OUT"/bench_pmc_*.log
ls -la "$OUT"
