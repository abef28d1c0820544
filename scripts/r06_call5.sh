#!/usr/bin/env bash
set -uo pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/r06"; mkdir -p "$OUT"; cd "$R"
timeout 600 python -m pytest tests/test_gpu_pick.py tests/test_gpu_field.py tests/test_gpu_train.py -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for W in config4 config4_opaque; do
  rm -rf /tmp/prof_s
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o kt -- python "$R/scripts/sampler_trace_r6.py" $W 16 > "$OUT/sampler_trace_$W.log" 2>&1
  python "$R/scripts/rocpd_summary.py" /tmp/prof_s/kt_results.db 40 > "$OUT/sampler_kernels_$W.txt" 2>&1
  echo "== $W"; tail -1 "$OUT/sampler_trace_$W.log"; head -30 "$OUT/sampler_kernels_$W.txt" | cut -c1-140
done
