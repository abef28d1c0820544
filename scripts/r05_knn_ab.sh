#!/usr/bin/env bash
# gpurun -- 'bash scripts/r05_knn_ab.sh'   same-box A/B of the exact k-NN: the library with knn.hip as it was at the start of round 5
# (commit 00402e9, built into sugar_amd/variants/lib_knn_round_start.so by SGR_SRC_OVERRIDE) against the final one -- the offset probe
# (124k queries at a fixed distance from config 4's surface), the round-4 far-query bench over a volume, the level-set sampling pass.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/r05
{
for which in round_start final; do
  if [ $which = round_start ]; then export SGR_LIB_PATH=$R/sugar_amd/variants/lib_knn_round_start.so SGR_TORCH_EXT=0; else unset SGR_LIB_PATH SGR_TORCH_EXT; fi
  echo "== knn.hip: $which"
  echo -n "offset probe (ms per call by distance): "; python scripts/knn_offset_probe.py 2>/dev/null | tail -1
  echo -n "offset probe again:                     "; python scripts/knn_offset_probe.py 2>/dev/null | tail -1
  echo -n "far-query bench over a volume:          "; python scripts/knn_far_bench.py 2>/dev/null | tail -1
  python scripts/sampler_profile_r5.py config4 2>/dev/null | grep "cam3_knn16_ms\|cam3_whole_pass_ms\|knn16_self_query_ms"
done
} | tee gpurun_out/r05/knn_ab_same_box.txt
