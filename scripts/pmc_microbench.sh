#!/usr/bin/env bash
# gpurun -- 'bash scripts/pmc_microbench.sh <binary under scripts/microbench> "CTR1 CTR2;CTR3 ..." [kernel-filter]'
# one rocprofv3 PMC pass per ';'-separated counter group over a microbenchmark binary, summaries to stdout
set -uo pipefail
BIN="$1"; GROUPS_="$2"; FILT="${3:-k_}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra GS <<< "$GROUPS_"
for G in "${GS[@]}"; do
  rm -rf /tmp/prof_pmc
  rocprofv3 --kernel-trace --pmc $G -d /tmp/prof_pmc -o pmc -- "$R/scripts/microbench/$BIN" > /tmp/pmc.log 2>&1
  python "$R/scripts/rocpd_pmc_summary.py" /tmp/prof_pmc/pmc_results.db $FILT 2>&1 | cut -c1-110
done
rm -rf /tmp/prof_pmc
