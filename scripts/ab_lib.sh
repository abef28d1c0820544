#!/usr/bin/env bash
# gpurun -- 'FILT="k_blend" bash scripts/ab_lib.sh a.so b.so [c.so ...]'   same-box A/B of builds of the library
# (SGR_BUILD_OUT=... python -m sugar_amd.build makes a variant; SGR_LIB_PATH selects it): bench under rocprofv3 --kernel-trace,
# two interleaved rounds, per-kernel averages -> gpurun_out/ab/
set -uo pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
FILT="${FILT:-k_blend}"
OUT="$R/gpurun_out/ab"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
  for name in "$@"; do
    lib="$R/sugar_amd/$name"; tag="$(basename "${name%.so}")_$round"
    rm -rf /tmp/prof_ab
    SGR_LIB_PATH=$lib rocprofv3 --kernel-trace --stats -d /tmp/prof_ab -o kt -- python "$R/bench.py" --steps 30 --warmup 5 --preroll 24 --no-cpu-baseline --no-densify-variant --drift-steps 0 --no-reference-loop --cameras 0 > "$OUT/bench_$tag.log" 2>&1
    python "$R/scripts/rocpd_summary.py" /tmp/prof_ab/kt_results.db 40 > "$OUT/kernels_$tag.txt" 2>&1
    echo "== $tag"; grep "$FILT" "$OUT/kernels_$tag.txt" | cut -c1-96
    grep '^{' "$OUT/bench_$tag.log" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ms_per_step', round(d['ms_per_step'],4))"
  done
done
