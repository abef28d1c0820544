#!/usr/bin/env bash
# round-5 GPU call: kernel trace of the level-set sampler on the flat scene; same-box A/B of the in-place walk-hint repair; the
# unmodified coarse trainer at 2M Gaussians @ 1080p (BASELINE config 3's size) with every opt-in binding
set -uo pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/r05"; mkdir -p "$OUT"; cd "$R"
python -m pytest tests/test_gpu_native_trainer.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python "$R/scripts/sampler_profile_r5.py" > "$OUT/sampler_profile_under_rocprof.json" 2> /tmp/sp.err
python "$R/scripts/rocpd_summary.py" /tmp/prof_kt/kt_results.db 25 > "$OUT/sampler_kernel_stats.txt" 2>&1
head -20 "$OUT/sampler_kernel_stats.txt" | cut -c1-150
cd "$R"
for round in 1 2; do
  for mode in repair norepair; do
    if [ $mode = norepair ]; then export SGR_NO_HINT_REPAIR=1; else unset SGR_NO_HINT_REPAIR; fi
    python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-densify-variant --no-reference-loop > "$OUT/ab_${mode}_$round.json" 2> /tmp/ab.err
    python - "$OUT/ab_${mode}_$round.json" <<'P'
import json,sys
j=json.load(open(sys.argv[1])); m=j.get("many_cameras_shuffled",{}); a=j.get("after_training",{})
print(sys.argv[1].split("/")[-1], "ms/step", round(j["ms_per_step"],4), "fwd+bwd", round(j["ms_fwd_bwd"],4), "scatter", round(j["stages_ms"]["bin_scatter"],4), "blend_fwd", round(j["stages_ms"]["blend_fwd"],4),
      "| drift", round(a.get("ms_per_step",0),4), "| many", round(m.get("ms_per_step",0),4), m.get("forwards_repeated"), m.get("hint_off_windows"), m.get("tiles_repaired_in_place"))
P
  done
done
unset SGR_NO_HINT_REPAIR
timeout 900 python scripts/run_reference_trainer.py --gaussians 2000000 --cameras 64 --width 1920 --height 1080 --stop-at 9200 --patch-losses --patch-optimizer --patch-gathers --patch-densifier --out "$OUT/trainer_2M" > "$OUT/trainer_2M.log" 2>&1
tail -3 "$OUT/trainer_2M.log"; ls "$OUT/trainer_2M" 2>/dev/null
