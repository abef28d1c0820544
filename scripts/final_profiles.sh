#!/usr/bin/env bash
# gpurun --timeout 1500 -- 'bash scripts/final_profiles.sh r02'
# Everything profiles/<TAG>_* is made from: the default bench line, the rocprofv3 kernel trace and PMC passes of bench.py,
# the VALU / LDS counters of the blend kernels, the A/B against the reference's own sources (oracle/_ref) and the
# timings at the larger BASELINE configs.  Outputs land under gpurun_out/final_<TAG>/ (copy the ones to keep into profiles/).
set -uo pipefail
TAG="${1:-r02}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/final_$TAG"
mkdir -p "$OUT"
cd "$R"
python bench.py > "$OUT/bench_n1.log" 2>&1; grep '^{' "$OUT/bench_n1.log" | tail -1 > "$OUT/bench_n1.json"
bash scripts/profile_on_box.sh "$TAG" > /dev/null 2>&1
cp gpurun_out/profile_$TAG/kernel_stats.txt gpurun_out/profile_$TAG/pmc_*.txt gpurun_out/profile_$TAG/bench_under_rocprof.json "$OUT/" 2>/dev/null
bash scripts/pmc_on_box.sh v1 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" k_ > /dev/null 2>&1
bash scripts/pmc_on_box.sh v2 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" k_ > /dev/null 2>&1
cat gpurun_out/pmc_v1.txt gpurun_out/pmc_v2.txt > "$OUT/pmc_valu.txt"
python tests/ab_reference.py config2 metric > "$OUT/ab_reference.log" 2>&1; cp gpurun_out/ab_reference.json "$OUT/" 2>/dev/null
python scripts/scale_check.py > "$OUT/scale_check.log" 2>&1; cp gpurun_out/scale_check.json "$OUT/" 2>/dev/null
ls -la "$OUT"; cat "$OUT/bench_n1.json" | cut -c1-400
