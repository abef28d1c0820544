#!/usr/bin/env bash
# gpurun -- 'FILT=k_blend bash scripts/ab_pmc.sh COUNTER a.so b.so ...'   one rocprofv3 --pmc pass per build of the library
set -uo pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; FILT="${FILT:-k_blend}"; C="$1"; shift
cd /tmp && export TMPDIR=/tmp
for name in "$@"; do
  rm -rf /tmp/prof_pmc
  SGR_LIB_PATH="$R/sugar_amd/$name" rocprofv3 --kernel-trace --pmc $C -d /tmp/prof_pmc -o pmc -- python "$R/bench.py" --steps 3 --warmup 1 --preroll 16 --no-cpu-baseline --no-densify-variant --drift-steps 0 --no-reference-loop --cameras 0 > /tmp/pmc.log 2>&1
  echo "== $name $C"; python "$R/scripts/rocpd_pmc_summary.py" /tmp/prof_pmc/pmc_results.db k_ 2>&1 | grep "$FILT" | cut -c1-130
done
