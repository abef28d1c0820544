#!/usr/bin/env bash
# gpurun -- 'bash scripts/pmc_on_box.sh TAG "COUNTER1 COUNTER2 ..." [kernel-filter]'   (one PMC pass, summary only)
set -uo pipefail
TAG="$1"; CTRS="$2"; FILT="${3:-k_}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/prof_pmc -o pmc -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --preroll 0 > /tmp/pmc.log 2>&1
python "$R/scripts/rocpd_pmc_summary.py" /tmp/prof_pmc/pmc_results.db $FILT > "$R/gpurun_out/pmc_$TAG.txt" 2>&1
rm -rf /tmp/prof_pmc
cat "$R/gpurun_out/pmc_$TAG.txt" | cut -c1-150
