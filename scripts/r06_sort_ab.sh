#!/usr/bin/env bash
# same-box A/B of the radix sort's chunk size / waves per chunk (variants built with SGR_EXTRA_DEFS="-DRS_ITEMS=.. -DRS_WAVES=..")
set -uo pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/r06/sort_ab"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
  for name in default rs_2048_8 rs_2048_16 rs_4096_8 rs_8192_16; do
    if [ $name = default ]; then unset SGR_LIB_PATH; else export SGR_LIB_PATH="$R/sugar_amd/variants/lib_$name.so"; fi
    rm -rf /tmp/prof_ab
    rocprofv3 --kernel-trace --stats -d /tmp/prof_ab -o kt -- python "$R/bench.py" --steps 30 --warmup 5 --preroll 24 --no-cpu-baseline --no-densify-variant --drift-steps 0 --no-reference-loop --cameras 0 > "$OUT/bench_${name}_$round.log" 2>&1
    python "$R/scripts/rocpd_summary.py" /tmp/prof_ab/kt_results.db 40 > "$OUT/kernels_${name}_$round.txt" 2>&1
    echo "== $name $round"; grep -E "k_rs_" "$OUT/kernels_${name}_$round.txt" | cut -c1-100
    grep '^{' "$OUT/bench_${name}_$round.log" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ms_per_step', round(d['ms_per_step'],4), 'depth_sort', round(d['stages_ms']['depth_sort'],4))"
  done
done
