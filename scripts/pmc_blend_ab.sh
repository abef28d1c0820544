#!/usr/bin/env bash
# gpurun -- 'bash scripts/pmc_blend_ab.sh TAG "VARIANTS ARGS" "COUNTERS" [kernel-filter]'   one PMC pass over scripts/blend_ab.py
set -uo pipefail
TAG="$1"; ARGS="$2"; CTRS="$3"; FILT="${4:-k_blend}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p "$R/gpurun_out/r02"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/prof_pmc -o pmc -- python "$R/scripts/blend_ab.py" $ARGS > /tmp/pmc.log 2>&1
python "$R/scripts/rocpd_pmc_summary.py" /tmp/prof_pmc/pmc_results.db $FILT > "$R/gpurun_out/r02/pmc_$TAG.txt" 2>&1
rm -rf /tmp/prof_pmc
cut -c1-130 "$R/gpurun_out/r02/pmc_$TAG.txt"
