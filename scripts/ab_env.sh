#!/usr/bin/env bash
# gpurun -- 'FILT="k_blend_fwd" bash scripts/ab_env.sh "" "SGR_EXP_X=1" ...'   same-box A/B of environment settings (one per argument;
# "" = none): bench under rocprofv3 --kernel-trace, two interleaved rounds, per-kernel averages -> gpurun_out/ab/
set -uo pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
FILT="${FILT:-k_blend}"
OUT="$R/gpurun_out/ab"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
  i=0
  for setting in "$@"; do
    i=$((i+1)); tag="env${i}_$round"
    rm -rf /tmp/prof_ab
    env $setting rocprofv3 --kernel-trace --stats -d /tmp/prof_ab -o kt -- python "$R/bench.py" --steps 30 --warmup 5 --preroll 24 --no-cpu-baseline --no-densify-variant --drift-steps 0 ${BENCH_ARGS:-} > "$OUT/bench_$tag.log" 2>&1
    python "$R/scripts/rocpd_summary.py" /tmp/prof_ab/kt_results.db 40 > "$OUT/kernels_$tag.txt" 2>&1
    echo "== $tag [$setting]"; grep "$FILT" "$OUT/kernels_$tag.txt" | cut -c1-96
    grep '^{' "$OUT/bench_$tag.log" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ms_per_step', round(d['ms_per_step'],4))"
  done
done
