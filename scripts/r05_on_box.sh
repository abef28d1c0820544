#!/usr/bin/env bash
# gpurun --timeout 3000 -- 'bash scripts/r05_on_box.sh'   round-5 evidence in one call (summaries under gpurun_out/r05/final/):
#   the GPU test suite, box state, one bench line per BASELINE config AS DEFINED (metric, 2, 3 = coarse-SDF-shaped step, 4 = flat
#   scene + sampler pass, 5, 5 as an unmodified caller), the gradient exchange forced on a one-rank RCCL group (both ways) and a
#   two-rank gloo rehearsal of the bench on the one GPU, the kernel trace of the metric bench, FETCH_SIZE / WRITE_SIZE passes for
#   EVERY workload (separate runs, --kernel-trace only) + instruction / LDS-conflict passes for the metric, their reductions to
#   profiles/pmc_blend_fwd_<workload>.json, the sampler profile on the flat scene with its kernel trace, the torch-profiler table of
#   the reference's loop with every opt-in binding, per-kernel register / scratch use.
set -uo pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/r05/final"
mkdir -p "$OUT"
cd "$R"
{ echo "# partition modes"; rocm-smi --showcomputepartition --showmemorypartition 2>&1 | grep -v "^=\|^$" | head -8; echo "# rocm-smi before"; rocm-smi --showclocks --showpower --showtemp --showmemuse --showperflevel 2>&1 | grep -v "^=\|^$" | head -40; } > "$OUT/box_state.txt"
python -m pytest tests -m gpu -q 2>&1 | tail -12 > "$OUT/gpu_tests.log"
cp gpurun_out/fullsize_parity.json "$OUT/fullsize_parity.json" 2>/dev/null
( time python bench.py --steps 20 --warmup 5 > "$OUT/bench_metric.json" 2> "$OUT/bench_metric.err" ) 2> "$OUT/bench_metric_wall_time.txt"
for w in config2 config3 config4 config5; do
  python bench.py --workload $w --steps 20 --warmup 5 --preroll 64 --drift-steps 0 --no-densify-variant --no-reference-loop > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"
done
python bench.py --workload config5 --host-sync --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_config5_unmodified_caller.json" 2> "$OUT/bench_c5u.err"
python bench.py --workload config3 --plain-3dgs-step --steps 20 --warmup 5 --preroll 64 --drift-steps 0 --no-densify-variant --no-reference-loop --no-cpu-baseline > "$OUT/bench_config3_plain_3dgs_step.json" 2> /dev/null
python bench.py --workload config4 --plain-3dgs-step --steps 20 --warmup 5 --preroll 64 --drift-steps 0 --no-densify-variant --no-reference-loop --no-cpu-baseline > "$OUT/bench_config4_plain_3dgs_step.json" 2> /dev/null
python bench.py --steps 20 --warmup 5 --force-collectives --no-cpu-baseline --no-densify-variant --drift-steps 0 --no-reference-loop > "$OUT/bench_metric_forced_collectives.json" 2> /dev/null
python bench.py --steps 20 --warmup 5 --force-collectives --native-collectives --no-cpu-baseline --no-densify-variant --drift-steps 0 --no-reference-loop > "$OUT/bench_metric_forced_collectives_in_library.json" 2> /dev/null
# two ranks sharing the one GPU over gloo: a FUNCTIONAL rehearsal of the launch line the driver uses for N > 1 (timings mean nothing)
SGR_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 6 --warmup 2 --preroll 16 > "$OUT/bench_two_rank_gloo_rehearsal.json" 2> "$OUT/bench_two_rank.err"
SGR_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29572 bench.py --gpus 2 --workload config4 --gaussians 200000 --steps 4 --warmup 2 --preroll 16 > "$OUT/bench_two_rank_gloo_rehearsal_config4.json" 2>> "$OUT/bench_two_rank.err"
python scripts/kernel_meta.py > "$OUT/kernel_meta.txt" 2>&1
python scripts/sampler_profile_r5.py config4 > "$OUT/sampler_profile_config4.json" 2> /dev/null
python scripts/sampler_profile_r5.py metric > "$OUT/sampler_profile_metric_scene.json" 2> /dev/null
python scripts/knn_far_bench.py > "$OUT/knn_far_bench.json" 2> /dev/null
python scripts/knn_offset_probe.py > "$OUT/knn_offset_probe.json" 2> /dev/null
SGR_KNN_DIRECT_MAX=0 python scripts/knn_offset_probe.py > "$OUT/knn_offset_probe_ring_walk_first.json" 2> /dev/null
python scripts/reference_loop_profile.py --all-patches > "$OUT/reference_loop_torch_profile_all_patches.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python "$R/bench.py" --steps 20 --warmup 5 --preroll 24 --no-cpu-baseline --no-densify-variant --drift-steps 0 --no-reference-loop > "$OUT/bench_under_rocprof.log" 2>&1
python "$R/scripts/rocpd_summary.py" /tmp/prof_kt/kt_results.db 60 > "$OUT/kernel_stats.txt" 2>&1
grep '^{' "$OUT/bench_under_rocprof.log" | tail -1 > "$OUT/bench_under_rocprof.json" || true
rm -rf /tmp/prof_kt
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python "$R/scripts/sampler_profile_r5.py" config4 > /tmp/sp.log 2>&1
python "$R/scripts/rocpd_summary.py" /tmp/prof_kt/kt_results.db 16 > "$OUT/sampler_kernel_stats.txt" 2>&1
rm -rf /tmp/prof_kt
for W in metric config2 config3 config4 config5; do
  CS=("FETCH_SIZE" "WRITE_SIZE")
  if [ "$W" = metric ]; then CS+=("SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES"); fi
  for C in "${CS[@]}"; do
    TAG=$(echo $C | cut -d' ' -f1)
    rm -rf /tmp/prof_pmc
    rocprofv3 --kernel-trace --pmc $C -d /tmp/prof_pmc -o pmc -- python "$R/bench.py" --workload $W --steps 3 --warmup 1 --preroll 16 --no-cpu-baseline --no-densify-variant --drift-steps 0 --no-reference-loop --cameras 0 > /tmp/pmc.log 2>&1
    python "$R/scripts/rocpd_pmc_summary.py" /tmp/prof_pmc/pmc_results.db k_ > "$OUT/pmc_${W}_$TAG.txt" 2>&1
  done
  python "$R/scripts/pmc_reduce.py" "$OUT" $W "pmc_${W}_" r05 > "$OUT/pmc_blend_fwd_$W.json" 2>> "$OUT/pmc_reduce.err"
done
rm -rf /tmp/prof_pmc
ls -la "$OUT"
