#!/usr/bin/env bash
# gpurun -- 'bash scripts/pmc_bench.sh TAG "COUNTERS" kernel-filter [bench args]'   one PMC pass over bench.py, per-kernel avg/min/max
set -uo pipefail
TAG="$1"; CTRS="$2"; FILT="${3:-k_}"; shift 3 || true
R="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p "$R/gpurun_out/r02"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/prof_pmc -o pmc -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --preroll 0 --no-stage-events "$@" > /tmp/pmc.log 2>&1
python "$R/scripts/rocpd_pmc_summary.py" /tmp/prof_pmc/pmc_results.db $FILT > "$R/gpurun_out/r02/pmc_$TAG.txt" 2>&1
rm -rf /tmp/prof_pmc
cut -c1-120 "$R/gpurun_out/r02/pmc_$TAG.txt"
