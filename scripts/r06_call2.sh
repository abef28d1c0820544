#!/usr/bin/env bash
set -uo pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/r06"; mkdir -p "$OUT"; cd "$R"
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_speculative.py tests/test_gpu_parity.py -x -q > "$OUT/pytest_gpu_call2.log" 2>&1; echo "pytest rc $? $(tail -1 $OUT/pytest_gpu_call2.log)"
cp gpurun_out/fullsize_parity.json "$OUT/fullsize_parity_call2.json"
timeout 900 python bench.py --no-cpu-baseline --no-reference-loop --cameras 0 --drift-steps 0 > "$OUT/bench_metric_call2.json" 2> "$OUT/bench_metric_call2.err"; echo "bench rc $?"
python - <<'P'
import json
d=json.load(open("gpurun_out/r06/bench_metric_call2.json"))
print(d["value"], d["ms_per_step"], d.get("exact_alpha_variant"))
d=json.load(open("gpurun_out/r06/fullsize_parity_call2.json"))
for k,v in d.items():
    g=v.get("grads",{})
    w=max(((e.get("product_vs_reference",e)["norm_rel"],n) for n,e in g.items()), default=(0,"-"))
    print(f"{k:40s} image {v.get('image',{}).get('norm_rel',0):.2e} flips {v.get('n_contrib_mismatch_frac',0):.1e} worst {w[0]:.2e} {w[1]}")
P
