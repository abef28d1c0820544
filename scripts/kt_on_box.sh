#!/usr/bin/env bash
# gpurun -- 'bash scripts/kt_on_box.sh TAG [bench.py args...]'   rocprofv3 kernel trace of bench.py, per-kernel summary only
set -uo pipefail
TAG="$1"; shift
R="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python "$R/bench.py" --no-cpu-baseline --no-densify-variant --drift-steps 0 "$@" > "$R/gpurun_out/kt_$TAG.log" 2>&1
python "$R/scripts/rocpd_summary.py" /tmp/prof_kt/kt_results.db 60 > "$R/gpurun_out/kt_$TAG.txt" 2>&1
rm -rf /tmp/prof_kt
head -40 "$R/gpurun_out/kt_$TAG.txt" | cut -c1-140
