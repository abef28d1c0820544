#!/usr/bin/env bash
# same-box A/B of two builds of the library on the metric workload and on config 4's flat scene (plain train step):
#   gpurun -- 'bash scripts/r05_ab_blend.sh lib_oldblend.so lib_newblend.so'
set -uo pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/r05/ab"; mkdir -p "$OUT"; cd "$R"
for round in 1 2; do
  for W in metric config4; do
    for name in "$@"; do
      tag="${W}_$(basename "${name%.so}")_$round"
      extra=""; [ "$W" = config4 ] && extra="--plain-3dgs-step"
      SGR_LIB_PATH="$R/sugar_amd/variants/$name" python bench.py --workload $W $extra --steps 40 --warmup 5 --preroll 64 --no-cpu-baseline --no-densify-variant --drift-steps 0 --no-reference-loop --cameras 0 > "$OUT/$tag.json" 2> "$OUT/$tag.err" || tail -3 "$OUT/$tag.err"
      python - "$OUT/$tag.json" "$tag" <<'P'
import json,sys
j=json.load(open(sys.argv[1])); s=j["stages_ms"]
print(f"{sys.argv[2]:40s} ms/step {j['ms_per_step']:.4f} fwd+bwd {j['ms_fwd_bwd']:.4f} blend_fwd {s['blend_fwd']:.4f} blend_bwd {s['blend_bwd']:.4f} scatter {s['bin_scatter']:.4f}")
P
    done
  done
done
