#!/usr/bin/env bash
# same-box A/B of the level-1 slice count of the two-level binning (SGR_EXTRA_DEFS=-DSGR_B2_SLICES=..)
set -uo pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/r06/b2_ab"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
  for name in default b2_4096 b2_1024; do
    if [ $name = default ]; then unset SGR_LIB_PATH; else export SGR_LIB_PATH="$R/sugar_amd/variants/lib_$name.so"; fi
    rm -rf /tmp/prof_ab
    rocprofv3 --kernel-trace --stats -d /tmp/prof_ab -o kt -- python "$R/bench.py" --steps 30 --warmup 5 --preroll 24 --no-cpu-baseline --no-densify-variant --drift-steps 0 --no-reference-loop --cameras 0 > "$OUT/bench_${name}_$round.log" 2>&1
    python "$R/scripts/rocpd_summary.py" /tmp/prof_ab/kt_results.db 40 > "$OUT/kernels_${name}_$round.txt" 2>&1
    echo "== $name $round"; grep -E "k_sup_|k_tile_" "$OUT/kernels_${name}_$round.txt" | cut -c1-100
    grep '^{' "$OUT/bench_${name}_$round.log" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ms_per_step', round(d['ms_per_step'],4), 'bin_count_scan', round(d['stages_ms']['bin_count_scan'],4))"
  done
done
