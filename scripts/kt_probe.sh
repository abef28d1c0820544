#!/usr/bin/env bash
# gpurun -- 'bash scripts/kt_probe.sh TAG'   rocprofv3 kernel trace of scripts/hint_probe.py, binning kernels only
set -uo pipefail
TAG="$1"
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python "$R/scripts/hint_probe.py" metric > /tmp/kt_probe.log 2>&1
tail -3 /tmp/kt_probe.log
python "$R/scripts/rocpd_summary.py" /tmp/prof_kt/kt_results.db 60 | grep -i "tile_pass\|sup_\|tile_scan\|blend_fwd" | cut -c1-120
rm -rf /tmp/prof_kt
