#!/usr/bin/env bash
# gpurun --timeout 2400 -- 'bash scripts/r04_on_box.sh'   round-4 evidence in one call (summaries under gpurun_out/r04/):
#   box state, the default bench line (+ config 2 / 3 / 5 lines, config 5 also as an unmodified caller makes the call), kernel
#   trace of the bench, PMC passes of the bench (separate runs, --kernel-trace only) and their reduction to
#   profiles/pmc_blend_fwd.json, the mesh z-buffer (kernel trace + FETCH / WRITE / VALU counters of scripts/mesh_bench.py), the
#   k-NN with far queries, the config-4 rehearsal with the reference class, per-kernel register / scratch use.
set -uo pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/r04"
mkdir -p "$OUT"
cd "$R"
{ echo "# partition modes"; rocm-smi --showcomputepartition --showmemorypartition 2>&1 | grep -v "^=\|^$" | head -8; echo "# rocm-smi before"; rocm-smi --showclocks --showpower --showtemp --showmemuse --showperflevel 2>&1 | grep -v "^=\|^$" | head -40; } > "$OUT/box_state.txt"
python bench.py --steps 20 --warmup 5 > "$OUT/bench_metric.json" 2> "$OUT/bench_metric.err"
for w in config2 config3 config5; do
  python bench.py --workload $w --steps 20 --warmup 5 --preroll 64 --drift-steps 0 --no-densify-variant --no-reference-loop > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"
done
python bench.py --workload config5 --host-sync --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_config5_unmodified_caller.json" 2> "$OUT/bench_c5u.err"
python scripts/config4_rehearsal_r4.py > "$OUT/config4_rehearsal.json" 2> "$OUT/config4.err"
python scripts/knn_far_bench.py > "$OUT/knn_far_bench.json" 2> "$OUT/knn_far.err"
python scripts/mesh_bench.py > "$OUT/mesh_bench.json" 2> "$OUT/mesh_bench.err"
python scripts/overlap_probe.py > "$OUT/overlap_probe.json" 2> "$OUT/overlap.err"
python scripts/kernel_meta.py > "$OUT/kernel_meta.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python "$R/bench.py" --steps 20 --warmup 5 --preroll 24 --no-cpu-baseline --no-densify-variant --drift-steps 0 --no-reference-loop > "$OUT/bench_under_rocprof.log" 2>&1
python "$R/scripts/rocpd_summary.py" /tmp/prof_kt/kt_results.db 60 > "$OUT/kernel_stats.txt" 2>&1
grep '^{' "$OUT/bench_under_rocprof.log" | tail -1 > "$OUT/bench_under_rocprof.json" || true
rm -rf /tmp/prof_kt
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES"; do
  TAG=$(echo $C | cut -d' ' -f1)
  rm -rf /tmp/prof_pmc
  rocprofv3 --kernel-trace --pmc $C -d /tmp/prof_pmc -o pmc -- python "$R/bench.py" --steps 3 --warmup 1 --preroll 16 --no-cpu-baseline --no-densify-variant --drift-steps 0 --no-reference-loop > /tmp/pmc.log 2>&1
  python "$R/scripts/rocpd_pmc_summary.py" /tmp/prof_pmc/pmc_results.db k_ > "$OUT/pmc_$TAG.txt" 2>&1
done
# the mesh z-buffer: kernel trace and counters of scripts/mesh_bench.py (2M faces @ 1080p, faces_per_pixel 10 / 1)
rm -rf /tmp/prof_kt
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python "$R/scripts/mesh_bench.py" > /tmp/mesh.log 2>&1
python "$R/scripts/rocpd_summary.py" /tmp/prof_kt/kt_results.db 30 > "$OUT/mesh_kernel_stats.txt" 2>&1
rm -rf /tmp/prof_kt
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  TAG=$(echo $C | cut -d' ' -f1)
  rm -rf /tmp/prof_pmc
  rocprofv3 --kernel-trace --pmc $C -d /tmp/prof_pmc -o pmc -- python "$R/scripts/mesh_bench.py" > /tmp/pmc.log 2>&1
  python "$R/scripts/rocpd_pmc_summary.py" /tmp/prof_pmc/pmc_results.db k_mesh k_splat > "$OUT/mesh_pmc_$TAG.txt" 2>&1
done
rm -rf /tmp/prof_pmc
python "$R/scripts/pmc_reduce.py" "$OUT" > "$OUT/pmc_blend_fwd.json" 2> "$OUT/pmc_reduce.err"
ls -la "$OUT"
