#!/usr/bin/env bash
set -uo pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/r06"; mkdir -p "$OUT"; cd "$R"
timeout 1500 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu_call3.log" 2>&1; echo "pytest rc $? $(tail -1 $OUT/pytest_gpu_call3.log)"
grep -E "^(FAILED|ERROR)" "$OUT/pytest_gpu_call3.log" | head -20
cp gpurun_out/fullsize_parity.json "$OUT/fullsize_parity_call3.json"
timeout 900 python bench.py --no-cpu-baseline --no-reference-loop --cameras 0 --drift-steps 0 > "$OUT/bench_metric_call3.json" 2> "$OUT/bench_metric_call3.err"; echo "bench rc $?"
timeout 600 python bench.py --force-collectives --native-collectives --no-cpu-baseline --no-reference-loop --cameras 0 --drift-steps 0 --no-densify-variant > "$OUT/bench_metric_forced_native_call3.json" 2> "$OUT/bench_metric_forced_native_call3.err"; echo "bench forced rc $?"
timeout 600 python bench.py --force-collectives --no-cpu-baseline --no-reference-loop --cameras 0 --drift-steps 0 --no-densify-variant > "$OUT/bench_metric_forced_torch_call3.json" 2> "$OUT/bench_metric_forced_torch_call3.err"; echo "bench forced torch rc $?"
python - <<'P'
import json
for n in ("bench_metric_call3","bench_metric_forced_native_call3","bench_metric_forced_torch_call3"):
    try:
        d=json.load(open(f"gpurun_out/r06/{n}.json"))
        print(n, round(d["value"],1), round(d["ms_per_step"],4), d.get("fast_alpha_variant",{}).get("ms_per_step"), d.get("comm_exposed_ms"), d.get("stages_cover_frac"))
    except Exception as e: print(n, "ERR", e)
d=json.load(open("gpurun_out/r06/fullsize_parity_call3.json"))
for k,v in d.items():
    g=v.get("grads",{})
    w=max(((e.get("product_vs_reference",e)["norm_rel"],n) for n,e in g.items()), default=(0,"-"))
    print(f"{k:40s} image {v.get('image',{}).get('norm_rel',0):.2e} flips {v.get('n_contrib_mismatch_frac',0):.1e} worst {w[0]:.2e} {w[1]}")
P
