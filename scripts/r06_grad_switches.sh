#!/usr/bin/env bash
# Which departure from the reference's arithmetic carries the gradient error?  (VERDICT r05 "weak" 1)
#   here:  for m in 0 1 3 4 8 16 32 64 127 ...; do SGR_BUILD_OUT=sugar_amd/variants/lib_x$m.so SGR_BLEND_DEFS="-DSGR_BLEND_CXX=$m" python -m sugar_amd.build; done
#   gpurun -- 'bash scripts/r06_grad_switches.sh 0 1 3 ...'      -> gpurun_out/r06/grad_switch_*.json + a table
set -uo pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/r06"; mkdir -p "$OUT"; cd "$R"
K='config3-2-True-sh or metric-5-True-sh or config4-1-True-sh or config3-4-True-precomp'
run() {  # tag, lib path ("" = the product build)
  rm -f gpurun_out/fullsize_parity.json
  if [ -n "$2" ]; then export SGR_LIB_PATH="$2"; else unset SGR_LIB_PATH; fi
  timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k "$K" > "$OUT/grad_switch_$1.log" 2>&1
  echo "== $1: $(tail -1 "$OUT/grad_switch_$1.log")"
  cp gpurun_out/fullsize_parity.json "$OUT/grad_switch_$1.json" 2>/dev/null
}
run asm ""
for m in "$@"; do run "x$m" "$R/sugar_amd/variants/lib_x$m.so"; done
unset SGR_LIB_PATH
python - "$OUT" asm $(for m in "$@"; do echo x$m; done) <<'P'
import json, sys, os
out = sys.argv[1]
rows = []
for tag in sys.argv[2:]:
    p = os.path.join(out, f"grad_switch_{tag}.json")
    if not os.path.exists(p):
        continue
    d = json.load(open(p))
    for case, v in d.items():
        g = v.get("grads", {})
        worst = max(((e["product_vs_reference"]["norm_rel"], n) for n, e in g.items()), default=(0, "-"))
        rows.append((tag, case, v.get("image", {}).get("norm_rel", 0), v.get("n_contrib_mismatch_frac", 0), worst[0], worst[1],
                     {n: e["product_vs_reference"]["norm_rel"] for n, e in g.items()}))
with open(os.path.join(out, "grad_switch_table.txt"), "w") as f:
    for r in rows:
        line = f"{r[0]:6s} {r[1]:24s} image {r[2]:.2e} flips {r[3]:.1e} worst {r[4]:.2e} ({r[5]})  " + " ".join(f"{k}={v:.1e}" for k, v in r[6].items())
        print(line); f.write(line + "\n")
P
